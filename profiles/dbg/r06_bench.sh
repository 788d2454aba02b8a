# round 6: headline bench (driver's settings, no extras) + kernel / CU-time tables.  bash profiles/dbg/r06_bench.sh TAG [more bench args]
cd $GRAFT_REPO_ROOT
TAG=$1; shift
timeout 900 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline "$@" > gpurun_out/${TAG}_bench_head.json 2> gpurun_out/${TAG}_bench_head.err
python - <<PY
import json
r=json.loads(open("gpurun_out/${TAG}_bench_head.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","steady_state_ms_per_step","verified") if k in r}, (r.get("batch_latency_ms") or {}).get("median"))
rf=r["roofline"]; print({k:v for k,v in rf.items() if k not in ("kernel","note")})
PY
bash profiles/dbg/rocprof_bench.sh $TAG 2>&1 | tail -34
