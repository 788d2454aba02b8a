RG_BENCH_MODEL_KWARGS='{"batch_lanes": 4}' timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/l4_full.json 2> gpurun_out/l4_full.err
python - <<PY
import json
r = json.loads(open("gpurun_out/l4_full.json").read().strip().splitlines()[-1])
print("lanes 4 full:", r["ms_per_step"], r.get("steady_state_ms_per_step"), r["value"], r.get("verified"))
for k, v in r.get("also", {}).items():
    print("   ", k, {kk: v[kk] for kk in v if kk in ("ms_per_step", "value", "verified", "ms_per_pass", "ms")} if isinstance(v, dict) else v)
PY
LANES="4 8" KS="20" TAG=rep1 bash profiles/dbg/sweep_lanes_auto.sh
LANES="4 8" KS="20" TAG=rep2 bash profiles/dbg/sweep_lanes_auto.sh
