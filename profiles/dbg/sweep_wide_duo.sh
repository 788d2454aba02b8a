#!/bin/bash
# lanes x (pairs=False, duo=True): the 64-workgroup two-sequence form (classifier-free pairs in workgroups of their own) in the pipeline
mkdir -p gpurun_out
for L in 4 5 6; do
  for K in 20 40; do
    RG_BENCH_MODEL_KWARGS="{\"batch_lanes\": $L, \"session_options\": {\"seq_pairs\": false, \"seq_duo\": true}}" \
      timeout 600 python bench.py --steps $K --warmup 5 --no-also --no-cpu-baseline > gpurun_out/wide_duo_l${L}_k${K}.json 2> gpurun_out/wide_duo_l${L}_k${K}.err
    python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/wide_duo_l${L}_k${K}.json").read().strip().splitlines()[-1])
    print("lanes $L K $K:", r["ms_per_step"], r.get("steady_state_ms_per_step"), r["value"], r.get("verified"), r["roofline"].get("launch_us"), r["roofline"].get("launch_form"))
except Exception as e:
    print("lanes $L K $K: failed", e)
PY
  done
done
