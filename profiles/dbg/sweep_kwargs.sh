#!/bin/bash
# bench.py --no-also --no-cpu-baseline with model kwargs: KW='{"decode_stream": true}' KS="20 40" TAG=dec bash profiles/dbg/sweep_kwargs.sh
mkdir -p gpurun_out
for K in ${KS:-20 40}; do
  for R in ${REPS:-1}; do
    f=gpurun_out/${TAG:-kw}_k${K}_r${R}
    RG_BENCH_MODEL_KWARGS="$KW" timeout 600 python bench.py --steps $K --warmup 5 --no-also --no-cpu-baseline > $f.json 2> $f.err
    python - <<PY
import json
try:
    r = json.loads(open("$f.json").read().strip().splitlines()[-1])
    print("${TAG:-kw} K $K:", r["ms_per_step"], r.get("steady_state_ms_per_step"), r["value"], r.get("verified"), r["roofline"].get("avg_launch_us"), r["roofline"].get("launch_form"), (r.get("batch_latency_ms") or {}).get("median"))
except Exception as e:
    print("${TAG:-kw} K $K: failed", e); print(open("$f.err").read()[-1500:])
PY
  done
done
