#!/bin/bash
# batch_lanes sweep with the pipeline's own choice of launch forms; LANES / KS from the environment
mkdir -p gpurun_out
for L in ${LANES:-4 5 8}; do
  for K in ${KS:-20 40}; do
    f=gpurun_out/${TAG:-auto}_l${L}_k${K}
    RG_BENCH_MODEL_KWARGS="{\"batch_lanes\": $L}" timeout 600 python bench.py --steps $K --warmup 5 --no-also --no-cpu-baseline > $f.json 2> $f.err
    python - <<PY
import json
try:
    r = json.loads(open("$f.json").read().strip().splitlines()[-1])
    print("lanes $L K $K:", r["ms_per_step"], r.get("steady_state_ms_per_step"), r["value"], r.get("verified"), r["roofline"].get("launch_us"), r["roofline"].get("launch_form"), r["config"].get("engines"))
except Exception as e:
    print("lanes $L K $K: failed", e)
PY
  done
done
