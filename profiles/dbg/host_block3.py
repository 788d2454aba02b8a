"""Right after an asynchronous forward() has been queued: which streams can still complete a tiny copy at once?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
wl = bench.Workload(rg, "guided", 16, dev, 0, 32768, pipelined=True)
m = wl.model
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
z = torch.zeros(4, device=dev)
fresh = torch.cuda.Stream()
pool = [torch.cuda.Stream() for _ in range(6)]
torch.cuda.synchronize()
ORDER = [["fresh", "search", "caller"], ["caller", "fresh", "search"], ["p0"], ["p1"], ["p2"], ["p3"], ["p4"], ["p5"]]
for rep in range(len(ORDER)):
    wl.step()
    print("idle right after the submit:", {n: st.query() for n, st in [("search", m._search_stream), ("fresh", fresh),
          ("caller", torch.cuda.current_stream()), ("lane0", m._lane_streams[0]), ("lane1", m._lane_streams[1])] if st is not None},
          "ids", [st.stream_id for st in m._lane_streams], m._search_stream.stream_id if m._search_stream else None)
    named = dict(search=m._search_stream, fresh=fresh, caller=torch.cuda.current_stream(), **{"p%d" % i: p for i, p in enumerate(pool)})
    for name, s in [(n, named[n]) for n in ORDER[rep]]:
        if s is None:
            print(name, "is None"); continue
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            z.cpu()
        print("%-16s tiny D2H copy returned after %.1f ms" % (name, (time.perf_counter() - t0) * 1e3), flush=True)
    torch.cuda.synchronize()
