"""Phase boundaries of the guided steps out of a kernel trace: inversion = stylize_kernel launches with 516 workgroups
(M = 2064 per lane), sampling = stylize_kernel with 344 (one lane, B = 16) or 172 (two lanes, B = 8) workgroups."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def wgs(r): return int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
marks = [i for i, r in enumerate(rows) if "FillFunctor" in r["Kernel_Name"] and int(r["Grid_Size_X"]) <= 256]
rows = rows[:marks[-1]]
ev = []
for r in rows:
    if r["Kernel_Name"].startswith("(anonymous namespace)::stylize_kernel"):
        w = wgs(r)
        ph = "inv" if w == 516 else ("smp" if w in (344, 172) else None)
        if ph:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), ph, r["Queue_Id"]))
# split into steps: a gap of > 5 ms between consecutive inversion kernels starts a new step
steps, cur, last = [], [], None
for e in ev:
    if e[2] == "inv" and last is not None and e[0] - last > 3e6 and any(x[2] == "smp" for x in cur):
        steps.append(cur); cur = []
    cur.append(e); last = e[1]
steps.append(cur)
t_prev_end = None
for st in steps[-4:]:
    t0 = min(e[0] for e in st)
    out = []
    for ph in ("inv", "smp"):
        for q in sorted(set(e[3] for e in st if e[2] == ph)):
            xs = [e for e in st if e[2] == ph and e[3] == q]
            out.append("%s q%s %.1f->%.1f" % (ph, q, (xs[0][0] - t0) / 1e6, (xs[-1][1] - t0) / 1e6))
    end = max(e[1] for e in st)
    print("step: " + " | ".join(out) + " | since previous step's last sampling kernel: %s ms" % ("%.1f" % ((t0 - t_prev_end) / 1e6) if t_prev_end else "-"))
    t_prev_end = end
