"""Compact view of profiles/lane_timeline.py's TIMED_REGION output: one row per chain / decode, device start -> end."""
import re, sys
L = [l for l in open(sys.argv[1])]
i = [k for k, l in enumerate(L) if "timed region" in l][0]
print(L[i].strip())
for l in L[i + 1:]:
    m = re.search(r"^\s+\('(\w+)',\s*(.{0,30}).*host call\s+([\d.]+) ->\s+[\d.]+ ms \| device\s+([\d.]+) ->\s+([\d.]+) ms \(([\d.]+)\)", l)
    if m and m.group(1) in ("invert", "cobatch", "guided", "sample", "dec"):
        print("%-8s %-30s host %7s  device %7s -> %7s (%s)" % (m.group(1), m.group(2), m.group(3), m.group(4), m.group(5), m.group(6)))
