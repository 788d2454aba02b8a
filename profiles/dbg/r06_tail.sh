# round 6: the denoiser forwards' tails (csrc/rg_tail.h): equality tests, launch times of the builds with / without it, bench A/B
cd $GRAFT_REPO_ROOT
TAG=${1:-r06T}
timeout 900 python -m pytest tests/test_cobatch_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/${TAG}_tests.txt
bash profiles/dbg/r06_variants.sh ${TAG}_variants "product hb nt" "product" > /dev/null 2>&1
cat gpurun_out/${TAG}_variants.txt
bash profiles/dbg/r06_ab.sh ${TAG} ${REPS:-2} '{"tail_glue": false}' '{"tail_glue": true}'
