timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -150 > gpurun_out/full_gpu_tests.txt
tail -3 gpurun_out/full_gpu_tests.txt
