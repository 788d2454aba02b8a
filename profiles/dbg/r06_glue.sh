cd $GRAFT_REPO_ROOT
python profiles/dbg/dyn_loop_check.py 16 48 8 2>&1 | grep -v amdgpu | tail -7
timeout 900 python -m pytest tests/test_cobatch_gpu.py tests/test_async_gpu.py tests/test_sampler_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python profiles/race_stress.py --reps 4 --batches 12 --B 16 --layers 8 --db 4096 --tag r06B 2>&1 | python profiles/dbg/stress_fmt.py | tail -1
bash profiles/dbg/r06_ab.sh r06B 3 '{}'
