# round 6: tail_glue as a session option of the pipeline: tests, then one A/B pair of the headline bench
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_cobatch_gpu.py tests/test_async_gpu.py tests/test_sampler_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee gpurun_out/r06W_tests.txt
bash profiles/dbg/r06_ab.sh r06W 1 '{"tail_glue": false}' '{}'
