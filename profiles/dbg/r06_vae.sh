cd $GRAFT_REPO_ROOT
TAG=${1:-r06f}
timeout 600 python profiles/dbg/venc_time.py 2>&1 | grep "ms per call" > gpurun_out/${TAG}_vae_times.txt
cat gpurun_out/${TAG}_vae_times.txt
timeout 1500 python -m pytest tests/test_vae_oracle_gpu.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/${TAG}_vae_tests.txt
cat gpurun_out/${TAG}_vae_tests.txt
