# kernel trace of the default bench + the gap analysis: bash profiles/dbg/rocprof_trace_gaps.sh <name>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/prof_$1
rm -rf $D && mkdir -p $D
rocprofv3 --kernel-trace -d $D --output-format csv -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-also > $D/bench.json 2> $D/err.txt
f=$(find $D -name "*kernel_trace.csv" | head -1)
head -1 $f
python3 profiles/dbg/chain_gaps.py $f | tee gpurun_out/$1_chain_gaps.txt
find $D -name "*.csv" -size +2M -delete
