# round 6: the tails in every loop: tests that walk the loops, then the default bench (with its `also` lines)
cd $GRAFT_REPO_ROOT
TAG=${1:-r06U}
timeout 1500 python -m pytest tests/test_cobatch_gpu.py tests/test_sampler_gpu.py tests/test_denoiser_gpu.py tests/test_async_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/${TAG}_tests.txt
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python - <<PY
import json
r=json.loads(open("gpurun_out/${TAG}_bench_default.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","steady_state_ms_per_step","verified") if k in r}, (r.get("batch_latency_ms") or {}).get("median"))
for k,v in (r.get("also") or {}).items():
    print(k, {q:v[q] for q in ("ms_per_step","value","verified") if isinstance(v,dict) and q in v})
PY
