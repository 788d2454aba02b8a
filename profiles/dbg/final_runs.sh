cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05y_bench_default.json 2> gpurun_out/r05y_bench_default.err
python - <<PY
import json
r=json.loads(open('gpurun_out/r05y_bench_default.json').read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','ms_per_step','steady_state_ms_per_step','verified','batch_latency_ms') if k in r})
rf=r['roofline']; print({k:v for k,v in rf.items() if k not in ('kernel','note')})
for k,v in r['also'].items(): print(k, v.get('ms_per_step'), v.get('value'), v['verified']['verified'])
PY
timeout 900 python profiles/race_stress.py --reps 20 --batches 10 --B 16 --tag r05y_default_4lanes_B16 2>&1 | tail -3
bash profiles/dbg/rocprof_bench.sh r05y 2>&1 | tail -45
