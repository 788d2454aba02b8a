cd $GRAFT_REPO_ROOT
TAG=${1:-r06i}
python profiles/dbg/dyn_loop_check.py 16 48 8 2>&1 | grep "MISMATCHES"
timeout 900 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/${TAG}_bench_head.json 2> gpurun_out/${TAG}_bench_head.err
tail -2 gpurun_out/${TAG}_bench_head.err
python - <<PY
import json
r=json.loads(open("gpurun_out/${TAG}_bench_head.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","steady_state_ms_per_step","verified") if k in r}, (r.get("batch_latency_ms") or {}).get("median"), r.get("verification"))
PY
timeout 1200 python profiles/race_stress.py --reps 12 --batches 10 --B 16 --tag ${TAG}_dyn 2>&1 | tail -3
