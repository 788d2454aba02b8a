cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_enc && rocprofv3 --kernel-trace -d gpurun_out/prof_enc --output-format csv -- python3 profiles/dbg/enc_kernels.py ${1:-16} > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_enc/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last 5 replays: everything after the last spin kernel
idx = max(i for i, r in enumerate(rows) if "spin" in r["Kernel_Name"] or "sleep" in r["Kernel_Name"].lower())
rows = rows[idx + 1:]
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"][:70]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernels per replay: %.1f, kernel time per replay %.1f us, span per replay %.1f us" % (len(rows) / 5, sum(a[1] for a in agg.values()) / 5, (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 5e3))
for n, a in agg.items():
    print("%-72s per replay %5.1f x %7.1f us" % (n, a[0] / 5, a[1] / a[0]))
PY
find gpurun_out/prof_enc -name "*.csv" -size +2M -delete
