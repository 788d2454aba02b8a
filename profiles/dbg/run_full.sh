timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -150 > gpurun_out/full_gpu_tests.txt
tail -3 gpurun_out/full_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 > gpurun_out/smoke.txt; tail -2 gpurun_out/smoke.txt
timeout 1500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo bench rc=$?; tail -c 300 gpurun_out/bench_default.err
