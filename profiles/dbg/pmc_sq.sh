cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/sq_a --output-format csv -- python3 $R/profiles/pmc_seq.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/sq_b --output-format csv -- python3 $R/profiles/pmc_seq.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $R/gpurun_out/sq_c --output-format csv -- python3 $R/profiles/pmc_seq.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("sq_a", "sq_b", "sq_c"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "rg_seq_kernel" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print(d, k, "launches", len(v), "mean %.4g" % (sum(v) / len(v)))
PY
