# round 6: the tail with its loads batched: equality tests, launch times, three A/B pairs of the headline bench
cd $GRAFT_REPO_ROOT
TAG=${1:-r06Y}
timeout 900 python -m pytest tests/test_cobatch_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee gpurun_out/${TAG}_tests.txt
timeout 600 python profiles/dbg/seq2_check.py 8 2>&1 | grep "MISMATCH\|us per" | tee gpurun_out/${TAG}_seq2_check.txt
bash profiles/dbg/r06_ab.sh ${TAG} ${REPS:-3} '{"tail_glue": false}' '{}'
