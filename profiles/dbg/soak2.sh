cd $GRAFT_REPO_ROOT
RG_BENCH_VERIFY_BATCHES=16 timeout 900 python bench.py --steps 100 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/soak_k100.json 2> gpurun_out/soak_k100.err
python - <<PY
import json
r=json.loads(open('gpurun_out/soak_k100.json').read().strip().splitlines()[-1])
print("K=100:", r['value'], r['ms_per_step'], r.get('steady_state_ms_per_step'), r['verified'], r.get('verification'), (r.get('batch_latency_ms') or {}).get('median'))
PY
timeout 900 python profiles/race_stress.py --reps 15 --batches 18 --B 16 --batch-lanes 8 --tag r05z_8lanes_B16 2>&1 | tail -2
