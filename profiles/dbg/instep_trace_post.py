import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# sampling-graph kernels at B = 16: stylize_kernel with 344 workgroups is unique to M = 1376
def wgs(r): return int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
marks = [i for i, r in enumerate(rows) if "FillFunctor" in r["Kernel_Name"] and int(r["Grid_Size_X"]) <= 256]
split = marks[-1] if marks else len(rows)
for name, part in (("in-step", rows[:split]), ("standalone", rows[split:])):
    agg = collections.defaultdict(list)
    for r in part[-60000:]:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        agg[(k, wgs(r))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(name)
    for (k, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print("   %-40s wgs=%4d n=%5d avg %6.2f us" % (k, w, len(v), sum(v) / len(v)))
