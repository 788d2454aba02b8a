cd $GRAFT_REPO_ROOT
TAG=${1:-r06g}
timeout 900 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/${TAG}_bench_head.json 2> gpurun_out/${TAG}_bench_head.err
tail -3 gpurun_out/${TAG}_bench_head.err
python - <<PY
import json
r=json.loads(open("gpurun_out/${TAG}_bench_head.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","steady_state_ms_per_step","verified") if k in r}, (r.get("batch_latency_ms") or {}).get("median"))
PY
timeout 2400 python -m pytest tests/test_async_gpu.py tests/test_pipeline_gpu.py tests/test_cobatch_gpu.py -x -q -m gpu --durations=8 2>&1 | tail -16 > gpurun_out/${TAG}_pipeline_tests.txt
tail -14 gpurun_out/${TAG}_pipeline_tests.txt
