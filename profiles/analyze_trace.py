"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the kernel sequence of one DDIM
step of the inversion graph (largest grids) and of the sampling graph.  Usage: analyze_trace.py <kernel_trace.csv>"""
import collections
import csv
import statistics
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:46]


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "cfg_ddim" in r["Kernel_Name"]]
    wins = collections.defaultdict(list)
    for a, b in zip(idx[:-1], idx[1:]):
        if b - a < 200:
            wins[(b - a, rows[a + 3]["Grid_Size_X"])].append((a, b))
    for sig, v in sorted(wins.items(), key=lambda kv: -len(kv[1]))[:4]:
        durs = [sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(a + 1, b + 1)) / 1e3
                for a, b in v]
        span = [(int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 for a, b in v]
        print("step signature (kernels, grid of 3rd kernel)=%s: %d steps, median kernel time %.1f us, median span %.1f us"
              % (sig, len(v), statistics.median(durs), statistics.median(span)))
        a, b = v[len(v) // 2]
        agg = collections.OrderedDict()
        for i in range(a + 1, b + 1):
            r = rows[i]
            key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            agg.setdefault(key, []).append(d)
        for (name, wgs), ds in agg.items():
            print("    %-46s wgs=%5d  n=%2d  avg %.2f us  total %.1f us" % (name, wgs, len(ds), sum(ds) / len(ds), sum(ds)))


if __name__ == "__main__":
    main(sys.argv[1])
