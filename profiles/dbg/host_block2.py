"""Which line of forward() is the host sitting on while the previous batch runs?  faulthandler dumps the stack 60 ms into the
second of two back-to-back steps."""
import faulthandler, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
wl = bench.Workload(rg, "guided", 16, dev, 0, 32768, pipelined=bool(int(os.environ.get("PIPELINED", "0"))))
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
wl.step()
import time
for delay in (0.02, 0.04, 0.07, 0.1):
    wl.step()
    faulthandler.dump_traceback_later(delay, repeat=False, file=sys.stdout)
    t0 = time.perf_counter()
    wl.step()
    print("forward + packing returned after %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    faulthandler.cancel_dump_traceback_later()
    torch.cuda.synchronize()
torch.cuda.synchronize()
