# FETCH / WRITE passes + timing for a tagged build: bash profiles/dbg/pmc_seq2_traffic.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export SEQ_PAIRS=1 SEQ_DUO=1 RG_LIB_TAG=$1
D=gpurun_out/pmc5_$1
rm -rf $D && mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D/seq_stats --output-format csv -- python3 profiles/pmc_seq.py > $D/info.txt 2> $D/e1.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/seq_fetch --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> $D/e2.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/seq_write --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> $D/e3.txt
python3 profiles/pmc_seq_summarize.py $D "$(grep '^{' $D/info.txt | tail -1)" | tail -8
find $D -name "*.csv" -size +3M -delete
