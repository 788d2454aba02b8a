# Which small kernels sit on the lanes' queues (and on the caller's) per batch: rocprofv3 kernel trace of a short bench run,
# aggregated per hardware queue.  bash profiles/dbg/r06_queue_census.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/prof_census; rm -rf $D
rocprofv3 --kernel-trace -d $D --output-format csv -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-also > /dev/null 2> $D.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("$D/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys())[:14])
qk = "Queue_Id" if "Queue_Id" in rows[0] else None
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in rows:
    q = r[qk]
    n = r["Kernel_Name"][:60]
    a = per[q][n]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = open("gpurun_out/r06_queue_census.txt", "w")
for q, ks in sorted(per.items(), key=lambda kv: -sum(a[0] for a in kv[1].values())):
    n2 = ks.get("rg_seq2_kernel(rg_seq_args)", [0, 0])[0]
    tot = sum(a[0] for a in ks.values())
    print("queue %s: %d kernels, %d rg_seq2 launches (%.1f chains)" % (q, tot, n2, n2 / 50.0), file=out)
    for n, a in sorted(ks.items(), key=lambda kv: -kv[1][0])[:18]:
        print("    %-62s calls=%6d  per chain %.1f  avg_us %.1f" % (n, a[0], a[0] / max(1.0, n2 / 50.0), a[1] / a[0]), file=out)
out.close()
print(open("gpurun_out/r06_queue_census.txt").read()[:9000])
PY
find $D -name "*.csv" -size +2M -delete
