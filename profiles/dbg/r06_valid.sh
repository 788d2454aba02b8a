cd $GRAFT_REPO_ROOT
TAG=${1:-r06s}
timeout 900 python -m pytest tests/test_async_gpu.py -x -q -m gpu -k "full_depth" 2>&1 | tail -4
timeout 600 python profiles/dbg/seq2_check.py 8 2>&1 | grep "MISMATCH\|us per"
for kw in '{"dynamic_forms": false}' '{"dynamic_forms": true}'; do
  timeout 700 python profiles/race_stress.py --reps 6 --batches 12 --B 16 --layers 8 --db 4096 --tag ${TAG} --model-kwargs "$kw" 2>&1 | python profiles/dbg/stress_fmt.py | tail -1
done
bash profiles/dbg/r06_ab.sh ${TAG} 2 '{"dynamic_forms": false}' '{"dynamic_forms": true}'
