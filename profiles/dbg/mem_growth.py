"""Reserved / allocated device memory over 300 co-batched steps (record_stream defers frees: does anything pile up?)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
wl = bench.Workload(rg, "guided", 16, dev, 0, 32768)
wl.prime()
t0 = time.perf_counter()
for i in range(300):
    wl.step()
    if i % 50 == 0:
        print("step %3d: allocated %.2f GB, reserved %.2f GB" % (i, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30), flush=True)
wl.drain()
torch.cuda.synchronize()
print("300 steps in %.1f s = %.1f ms per step; allocated %.2f GB, reserved %.2f GB, sessions %d, graphs %d"
      % (time.perf_counter() - t0, (time.perf_counter() - t0) / 0.3, torch.cuda.memory_allocated() / 2**30,
         torch.cuda.memory_reserved() / 2**30, len(wl.model._sessions), len(wl.model._graphs)))
