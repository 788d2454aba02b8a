# round 6, run 1: bit identity of the two denoiser kernels, launch times, in-kernel stamps, SQ / instruction-cache counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06a}
cd $R
mkdir -p gpurun_out
timeout 600 python profiles/dbg/seq2_check.py 8 > gpurun_out/${TAG}_seq2_check.txt 2>&1
tail -12 gpurun_out/${TAG}_seq2_check.txt
timeout 300 python profiles/dbg/seq2_time.py one,duo,duo_pairs > gpurun_out/${TAG}_seq2_time.txt 2>&1
cat gpurun_out/${TAG}_seq2_time.txt
RG_DIAG=1 timeout 300 python profiles/dbg/seq2_stamps.py 64 0 > gpurun_out/${TAG}_seq2_stamps.txt 2>&1
head -4 gpurun_out/${TAG}_seq2_stamps.txt | cut -c1-400
export SEQ_PAIRS=0 SEQ_DUO=1 SEQ_REPS=4
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${TAG}_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_${TAG}_$i --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> gpurun_out/pmc_${TAG}_$i.err
done
python3 - $TAG <<'PY' > gpurun_out/${TAG}_pmc_sq.txt 2>&1
import csv, glob, collections, sys
tag = sys.argv[1]
for i in range(1, 5):
    for f in glob.glob("gpurun_out/pmc_%s_%d/**/*counter_collection.csv" % (tag, i), recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "rg_seq2_kernel" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print("set%d %-28s launches %d mean %.5g" % (i, k, len(v), sum(v) / len(v)))
PY
cat gpurun_out/${TAG}_pmc_sq.txt
tail -3 gpurun_out/pmc_${TAG}_2.err
find gpurun_out -name "*.csv" -size +3M -delete
