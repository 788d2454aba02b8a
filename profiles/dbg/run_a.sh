timeout 300 python bench.py --workload base --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r.get('steady_state_ms_per_step'), r['batch_latency_ms']['median'], r['roofline']['frac'])"
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r.get('steady_state_ms_per_step'), r['batch_latency_ms']['median'], r['roofline']['frac'])"; done
