# A/B of pipeline constructor options on the headline bench, alternating, in one process sequence on one box:
#   bash profiles/dbg/r06_ab.sh TAG REPS 'json1' 'json2' ...
cd $GRAFT_REPO_ROOT
TAG=$1; REPS=$2; shift; shift
OUT=gpurun_out/${TAG}_ab.txt; : > $OUT
for r in $(seq 1 $REPS); do
  for kw in "$@"; do
    RG_BENCH_MODEL_KWARGS="$kw" timeout 600 python bench.py --steps ${STEPS:-20} --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-70s ms_per_step %.2f steady %.2f verified %s latency %s' % ('$kw', r['ms_per_step'], r.get('steady_state_ms_per_step') or 0, r.get('verified'), (r.get('batch_latency_ms') or {}).get('median')))" >> $OUT
  done
done
cat $OUT
