"""From a rocprofv3 kernel trace of bench.py: what sits between two consecutive denoiser launches of one lane.
usage: python3 profiles/dbg/chain_gaps.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = "Stream_Id" if "Stream_Id" in rows[0] and len({r["Stream_Id"] for r in rows}) > 2 else "Queue_Id"
by = collections.defaultdict(list)
for r in rows:
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    nm = r["Kernel_Name"]
    if "rg_seq2_kernel" in nm or "rg_seq_kernel" in nm:
        nm = nm.split("(")[0] + " x%d workgroups" % wg
    by[r[key]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm))
gaps, inside, trans, names = [], [], [], collections.Counter()
durs = collections.defaultdict(list)
for q, ks in by.items():
    ks.sort()
    last = None
    for i, (s, e, n) in enumerate(ks):
        if "rg_seq2_kernel" in n or "rg_seq_kernel" in n:
            if last is not None and i - last <= 8:
                mid = ks[last + 1:i]
                g = (s - ks[last][1]) / 1e3
                if g < 2000:
                    gaps.append(g)
                    inside.append(sum(b - a for a, b, _ in mid) / 1e3)
                    trans.append(len(mid) + 1)
                    for a, b, m in mid:
                        names[m[:60]] += 1
                        durs[m[:60]].append((b - a) / 1e3)
            last = i
            durs["<denoiser launch>"].append((e - s) / 1e3)
            durs[n].append((e - s) / 1e3)
            if last is not None and i == last and False:
                pass
import statistics as st
print("lanes keyed by", key, "; pairs of consecutive denoiser launches:", len(gaps))
print("gap between them: mean %.1f us, median %.1f; kernels inside run %.1f us (mean), %d-%d kernels; idle in the gap %.1f us = %.1f us per transition"
      % (st.mean(gaps), st.median(gaps), st.mean(inside), min(trans) - 1, max(trans) - 1, st.mean(gaps) - st.mean(inside),
         (st.mean(gaps) - st.mean(inside)) / st.mean(trans)))
print("denoiser launch: mean %.1f us over %d" % (st.mean(durs["<denoiser launch>"]), len(durs["<denoiser launch>"])))
for n in sorted(durs):
    if "workgroups" in n:
        print("   %-62s x%6d  mean %7.1f us  median %7.1f" % (n, len(durs[n]), st.mean(durs[n]), st.median(durs[n])))
# start-to-start period of consecutive launches of the widest rg_seq2 form on one lane
per = []
for q, ks in by.items():
    seq = [(s, e) for s, e, n in ks if n.startswith("rg_seq2_kernel x64")]
    per += [(b[0] - a[0]) / 1e3 for a, b in zip(seq, seq[1:]) if (b[0] - a[0]) < 5e6]
if per:
    print("rg_seq2 x64: start-to-start period on a lane: mean %.1f us, median %.1f (n=%d)" % (st.mean(per), st.median(per), len(per)))
for n, c in names.most_common(12):
    print("   %-62s x%6d  mean %6.1f us" % (n, c, st.mean(durs[n])))
