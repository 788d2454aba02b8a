"""Per-launch view of a graph-replayed forward chain: from a rocprofv3 --kernel-trace CSV of
`step_micro.py --B=<B> <engine>` print, per position inside one forward step, the kernel, its workgroups, its mean duration
and the mean gap to the previous kernel's end (replays only: the last 5 x 10 forwards).
Usage: launch_trace.py <kernel_trace.csv> <launches per forward>"""
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:40]


def main(path, per):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-per * 50:]
    dur = [[] for _ in range(per)]
    gap = [[] for _ in range(per)]
    for i, r in enumerate(rows):
        k = i % per
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if i:
            gap[k].append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
    tot_d = tot_g = 0.0
    for k in range(per):
        r = rows[k]
        d, g = sum(dur[k]) / len(dur[k]), sum(gap[k]) / max(1, len(gap[k]))
        tot_d, tot_g = tot_d + d, tot_g + g
        print("%3d %-40s wgs=%5d  dur %6.2f us  gap %5.2f us" % (k, short(r["Kernel_Name"]),
              int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), d, g))
    print("per forward: kernels %.1f us + gaps %.1f us = %.1f us" % (tot_d, tot_g, tot_d + tot_g))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
