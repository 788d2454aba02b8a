cd $GRAFT_REPO_ROOT
timeout 600 python profiles/race_stress.py --reps 20 --batches 10 --B 16 --tag r05B_default_4lanes_B16 2>&1 | tail -1
RG_BENCH_VERIFY_BATCHES=16 timeout 600 python bench.py --steps 100 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/soak_r05B_k100.json 2> gpurun_out/soak_r05B_k100.err
python - <<PY
import json
r=json.loads(open('gpurun_out/soak_r05B_k100.json').read().strip().splitlines()[-1])
print("K=100:", r['value'], r['ms_per_step'], r.get('steady_state_ms_per_step'), r['verified'], r.get('verification'), (r.get('batch_latency_ms') or {}).get('median'))
PY
