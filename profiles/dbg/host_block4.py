"""While the lanes' graphs of an asynchronously submitted batch run: which streams still execute a tiny kernel at once?
Every candidate gets its kernel + event first; the events are polled afterwards (no probe delays another)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
wl = bench.Workload(rg, "guided", 16, dev, 0, 32768, pipelined=True)
m = wl.model
if os.environ.get("NO_PART_STREAMS"):
    m.model.gesture_rep_encoder.part_streams = None
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
pool = [torch.cuda.Stream() for _ in range(12)]
z = [torch.zeros(4, device=dev) for _ in range(20)]
torch.cuda.synchronize()
for rep in range(3):
    wl.step()
    named = [("caller", torch.cuda.current_stream()), ("search", m._search_stream), ("lane0", m._lane_streams[0]),
             ("lane1", m._lane_streams[1])] + [("p%d" % i, p) for i, p in enumerate(pool)]
    evs = []
    t0 = time.perf_counter()
    for i, (name, s) in enumerate(named):
        with torch.cuda.stream(s):
            z[i].add_(1)
            e = torch.cuda.Event(); e.record(); evs.append(e)
    done_at = {}
    while len(done_at) < len(named) and time.perf_counter() - t0 < 0.3:
        for (name, _), e in zip(named, evs):
            if name not in done_at and e.query():
                done_at[name] = (time.perf_counter() - t0) * 1e3
        time.sleep(0.0005)
    print("probe kernels done after (ms): " + "  ".join("%s %.0f" % (n, done_at.get(n, -1)) for n, _ in named), flush=True)
    torch.cuda.synchronize()
