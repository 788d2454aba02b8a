"""rocprofv3 --kernel-trace target: 3 pipeline steps with one sampling lane (B = 16), a marker, then 3 standalone replays of
the same sampling graph.  Post-processing (instep_trace_post.py) compares the kernels of the sampling graph in both."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
wl = bench.Workload(rg, "guided", 16, dev, 0, 32768)
m = wl.model
m.sample_lanes = int(os.environ["SAMPLE_LANES"]) if "SAMPLE_LANES" in os.environ else None
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
torch.zeros(7, device=dev).fill_(1.0)          # marker kernel (FillFunctor on 7 elements)
torch.cuda.synchronize()
keys = [k for k in m._graphs if k[0] in os.environ.get("GRAPHS", "guided").split(",")]
for key in keys[:1]:                       # one lane's graph alone
    for _ in range(3):
        with torch.cuda.stream(m._lane_streams[0]):
            m._graphs[key][0].replay()
        torch.cuda.synchronize()
