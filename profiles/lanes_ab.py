"""A/B of the schedule knobs on the guided B = 16 workload (bench.py's Workload, graph-replayed steps): clip lanes of the
exemplar inversion, clip lanes of the sampling loops, stylization inside the GEMMs.  ms per step, median of 5."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
bench.Workload.database_index = lambda self: self.model.model.database.index
wl = bench.Workload(rg, "guided", 16, dev, 0, 32768)
db = wl.database
CASES = [dict(lanes=2, pipelined=False), dict(lanes=2, pipelined=True), dict(lanes=2, pipelined=True, cobatch=True, steps=20),
         dict(lanes=2, pipelined=True, cobatch=True, steps=20, inflight=3),
         dict(lanes=3, pipelined=True, cobatch=True, steps=20)]
if len(sys.argv) > 1:
    CASES = [eval("dict(%s)" % a) for a in sys.argv[1:]]
for case in CASES:
    m = wl.model
    m.lanes, m.sample_lanes = case["lanes"], case.get("sample_lanes")
    m.session_options = dict(m.session_options or {})      # (round 2's styl_in_gemm case: the knob was removed in round 5)
    m.async_results = bool(case.get("pipelined"))
    wl.cobatch = bool(case.get("cobatch")) and m.async_results
    m.max_inflight = int(case.get("inflight", 2))
    m._sessions.clear(); m._graphs.clear()
    for _ in range(4):
        wl.step()
    wl.drain()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):       # like bench.py: 5 steps back to back, one synchronisation at the end
        t0 = time.perf_counter()
        n = int(case.get("steps", 5))
        for _ in range(n):
            wl.step()
        wl.drain()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3 / n)
    print(case, "%.1f ms per step (min %.1f)" % (statistics.median(ts), min(ts)), flush=True)
