cd $GRAFT_REPO_ROOT
TAG=${1:-r06l}
timeout 900 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/${TAG}_bench_head.json 2> gpurun_out/${TAG}_bench_head.err
tail -2 gpurun_out/${TAG}_bench_head.err
python - <<PY
import json
r=json.loads(open("gpurun_out/${TAG}_bench_head.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","steady_state_ms_per_step","verified") if k in r}, (r.get("batch_latency_ms") or {}).get("median"), r.get("verification"))
PY
COBATCH=1 PIPELINED=1 TIMED_REGION=20 CALIBRATE=0 python profiles/lane_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_timed_region.txt; python profiles/dbg/timeline_rows.py gpurun_out/${TAG}_timed_region.txt | tail -32
timeout 600 python profiles/race_stress.py --reps 8 --batches 10 --B 16 --tag ${TAG} 2>&1 | tail -1 | cut -c1-200
