cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SEQ_PAIRS=0 SEQ_DUO=1 SEQ_REPS=4
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_valu_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_valu_$i --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> gpurun_out/pmc_valu_$i.err
done
python3 - <<'PY' > gpurun_out/r06_pmc_sq.txt 2>&1
import csv, glob, collections
for i in (1, 2):
    for f in glob.glob("gpurun_out/pmc_valu_%d/**/*counter_collection.csv" % i, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "rg_seq2_kernel" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print("set%d %-28s launches %d mean %.5g" % (i, k, len(v), sum(v) / len(v)))
PY
cat gpurun_out/r06_pmc_sq.txt
find gpurun_out -name "*.csv" -size +3M -delete
