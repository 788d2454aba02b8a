# round 6: launch times (and bit identity) of tagged experiment builds of the library.  bash profiles/dbg/r06_variants.sh OUT "tag1 tag2 ..." [check tags]
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1.txt; : > $OUT
for t in $2; do
  if [ "$t" = product ]; then unset RG_LIB_TAG; else export RG_LIB_TAG=$t; fi
  timeout 300 python profiles/dbg/seq2_time.py ${FORMS:-one,duo,duo_pairs} 2>&1 | grep "us per forward" | sed "s/^/[$t] /" >> $OUT
done
for t in $3; do
  if [ "$t" = product ]; then unset RG_LIB_TAG; else export RG_LIB_TAG=$t; fi
  timeout 600 python profiles/dbg/seq2_check.py 8 2>&1 | grep -v "^B=.*compared\|amdgpu.ids" | head -12 | sed "s/^/[$t] /" >> $OUT
done
cat $OUT
