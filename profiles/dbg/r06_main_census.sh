# kernels on the caller's queue during steady-state submissions (between two marker kernels), per submission
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/prof_main; rm -rf $D
TRACE=1 rocprofv3 --kernel-trace -d $D --output-format csv -- python3 profiles/dbg/op_census.py > /dev/null 2> $D.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("$D/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "spin_kernel" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
q0 = rows[a]["Queue_Id"]
seg = [r for r in rows[a + 1:b] if r["Queue_Id"] == q0]
N = 6
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    x = agg[r["Kernel_Name"][:64]]; x[0] += 1; x[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e6
print("caller's queue %s: %.1f kernels per submission, %.2f ms of kernel time per submission, %.1f ms between the markers per submission" % (q0, len(seg) / N, sum(x[1] for x in agg.values()) / N / 1e3, span / N))
for n, x in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print("   %-66s per submission %6.1f x %7.1f us" % (n, x[0] / N, x[1] / x[0]))
PY
find $D -name "*.csv" -size +2M -delete
