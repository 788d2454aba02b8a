import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def wgs(r): return int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
marks = [i for i, r in enumerate(rows) if "FillFunctor" in r["Kernel_Name"] and int(r["Grid_Size_X"]) <= 256]
split = marks[-1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 344
for name, part in (("in-step", rows[:split]), ("standalone", rows[split:])):
    idx = [i for i, r in enumerate(part) if "cfg_ddim" in r["Kernel_Name"] and wgs(r) == W]
    spans, busy, others = [], [], []
    for a, b in zip(idx[:-1], idx[1:]):
        t0, t1 = int(part[a]["End_Timestamp"]), int(part[b]["End_Timestamp"])
        if t1 - t0 > 3e6:
            continue
        q = part[b]["Queue_Id"]
        mine = [r for r in part[a + 1:b + 1] if r["Queue_Id"] == q]
        oth = [r for r in part[a + 1:b + 1] if r["Queue_Id"] != q]
        spans.append((t1 - t0) / 1e3)
        busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mine) / 1e3)
        others.append(len(oth))
    if spans:
        print("%-10s steps %4d: span per step median %.1f us, own-queue kernel time %.1f us (%d kernels), kernels of other queues inside the span: median %d"
              % (name, len(spans), statistics.median(spans), statistics.median(busy), len(mine), statistics.median(others)))
