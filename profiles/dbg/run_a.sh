run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r.get('steady_state_ms_per_step'), r['batch_latency_ms']['median'])"; }
export GPU_MAX_HW_QUEUES=8
for cfg in "0 2" "0 4" "1 2" "1 3" "1 4" "1 6"; do set -- $cfg
echo "=== TAIL_ASIDE=$1 MAX_INFLIGHT=$2"; RG_TAIL_ASIDE=$1 RG_MAX_INFLIGHT=$2 run
done
