# A/B of library builds (RG_LIB_TAG) on the headline bench with verification detail: bash profiles/dbg/r06_ab2.sh TAG REPS "tag1 tag2" [model kwargs json]
cd $GRAFT_REPO_ROOT
TAG=$1; REPS=$2; TAGS=$3; KW=${4:-"{}"}
OUT=gpurun_out/${TAG}_ab.txt; : > $OUT
for r in $(seq 1 $REPS); do
  for t in $TAGS; do
    if [ "$t" = product ]; then unset RG_LIB_TAG; else export RG_LIB_TAG=$t; fi
    RG_BENCH_MODEL_KWARGS="$KW" timeout 600 python bench.py --steps ${STEPS:-20} --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); v=r.get('verification') or {}
print('%-10s ms_per_step %.2f steady %.2f verified %s %s' % ('$t', r['ms_per_step'], r.get('steady_state_ms_per_step') or 0, r.get('verified'), v.get('first_mismatch') or ''))" >> $OUT
  done
done
cat $OUT
