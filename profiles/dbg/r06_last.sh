# round 6, last tree: stamps + PMC of the final rg_seq2.hip, copied where bench.py reads them, then the default bench, the
# rocprofv3 kernel tables, the new tests and the race stress
cd $GRAFT_REPO_ROOT
bash profiles/dbg/r06_final.sh stamps pmc
cp gpurun_out/r06_seq2_stamps.json gpurun_out/r06_seq2_stamps.txt gpurun_out/r06_pmc_seq2_wide.json gpurun_out/r06_pmc_seq2_wide.txt gpurun_out/r06_pmc_seq2_pairs.json gpurun_out/r06_pmc_seq2_pairs.txt profiles/
timeout 600 python -m pytest tests/test_cobatch_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee gpurun_out/r06V_tests.txt
bash profiles/dbg/r06_final.sh bench prof stress
