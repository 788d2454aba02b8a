cd $GRAFT_REPO_ROOT
timeout 600 python profiles/dbg/venc_time.py 2>&1 | grep "ms per call"
timeout 600 python -m pytest tests/test_vae_oracle_gpu.py tests/test_denoiser_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python profiles/race_stress.py --reps 4 --batches 12 --B 16 --layers 8 --db 4096 --tag r06x 2>&1 | python profiles/dbg/stress_fmt.py | tail -1
bash profiles/dbg/r06_ab.sh r06x 2 '{}'
