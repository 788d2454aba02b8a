cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SEQ_PAIRS=1 SEQ_DUO=1
cd $R
rm -rf gpurun_out/pmc5 && mkdir -p gpurun_out/pmc5
rocprofv3 --kernel-trace --stats -d gpurun_out/pmc5/seq_stats --output-format csv -- python3 profiles/pmc_seq.py > gpurun_out/pmc5/info.txt 2> gpurun_out/pmc5/e1.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc5/seq_fetch --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> gpurun_out/pmc5/e2.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc5/seq_write --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> gpurun_out/pmc5/e3.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc5/seq_mfma --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> gpurun_out/pmc5/e4.txt
python3 profiles/pmc_seq_summarize.py gpurun_out/pmc5 "$(grep '^{' gpurun_out/pmc5/info.txt | tail -1)" gpurun_out/r05k_pmc_seq2_pairs.json > gpurun_out/r05k_pmc_seq2_pairs.txt 2>&1
cat gpurun_out/r05k_pmc_seq2_pairs.txt | tail -25
find gpurun_out/pmc5 -name "*.csv" -size +3M -delete
