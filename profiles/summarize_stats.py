"""rocprofv3 --kernel-trace --stats: *_kernel_stats.csv -> the text summary kept under profiles/ (top kernels by total time)."""
import csv
import sys


def main(path, header=""):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(header)
    print("total kernel time %.1f ms; top kernels by total time:" % (tot / 1e6))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
        print("%-100s calls=%6d total_ms=%8.2f avg_us=%8.2f pct=%s" % (r["Name"][:100], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                                      float(r["AverageNs"]) / 1e3, r["Percentage"]))


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
