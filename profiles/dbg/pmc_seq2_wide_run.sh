cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SEQ_PAIRS=0 SEQ_DUO=1
cd $R
rm -rf gpurun_out/pmc6 && mkdir -p gpurun_out/pmc6
rocprofv3 --kernel-trace --stats -d gpurun_out/pmc6/seq_stats --output-format csv -- python3 profiles/pmc_seq.py > gpurun_out/pmc6/info.txt 2> gpurun_out/pmc6/e1.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc6/seq_fetch --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> gpurun_out/pmc6/e2.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc6/seq_write --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> gpurun_out/pmc6/e3.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc6/seq_mfma --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> gpurun_out/pmc6/e4.txt
python3 profiles/pmc_seq_summarize.py gpurun_out/pmc6 "$(grep '^{' gpurun_out/pmc6/info.txt | tail -1)" gpurun_out/r05w_pmc_seq2_wide.json > gpurun_out/r05w_pmc_seq2_wide.txt 2>&1
cat gpurun_out/r05w_pmc_seq2_wide.txt | tail -25
find gpurun_out/pmc6 -name "*.csv" -size +3M -delete
