"""Why did the fp32-equivalent workload go from 338 to 577 ms per step?  bench.Workload(precision="fp32") under a few switches."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
bench.Workload.database_index = lambda self: self.model.model.database.index
db = rg.synth.SyntheticDataset(32768, seed=2025, device=dev, feat_device=dev)
for tag, setup in (("default", lambda w: None), ("phases", lambda w: None)):
    w = bench.Workload(rg, "guided", 16, dev, 0, 32768, precision="fp32", database=db)
    setup(w)
    d = w.timed(4, 1, torch.cuda.synchronize)
    print("%s: %.1f ms per step" % (tag, d / 4 * 1e3), flush=True)
    if tag == "phases":
        m = w.model
        w.drain()
        m.async_results = False; m.lanes = 1
        w.step(); torch.cuda.synchronize()
        m.profile_phases, m.phase_ms = True, {}
        m.model.database.phase_ms = m.phase_ms
        w.step(); torch.cuda.synchronize()
        print("   phases (one synchronised single-lane step): " + ", ".join("%s %.1f" % kv for kv in m.phase_ms.items()))
