run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['batch_latency_ms']['median'])"; }
for rep in 1 2; do
echo "=== all off, 4 queues"; GPU_MAX_HW_QUEUES=4 RG_FRONT_STREAM=0 RG_COND_ASIDE=0 RG_CLIP_ENCODE_ASIDE=0 run
echo "=== clip aside only, 4 queues"; GPU_MAX_HW_QUEUES=4 RG_FRONT_STREAM=0 RG_COND_ASIDE=0 RG_CLIP_ENCODE_ASIDE=1 run
echo "=== all on, 8 queues"; run
echo "=== front only, 8 queues"; RG_COND_ASIDE=0 RG_CLIP_ENCODE_ASIDE=0 run
done
