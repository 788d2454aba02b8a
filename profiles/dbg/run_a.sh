timeout 1500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo rc=$?; tail -c 600 gpurun_out/bench_default.err; wc -c gpurun_out/bench_default.json
