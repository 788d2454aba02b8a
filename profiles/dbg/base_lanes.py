"""Experiment: base workload (config 2, 32 clips per batch) with whole batches alternating between N lanes via submit()."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
for base_lanes, inflight, slots in ((3, 2, 2), (3, 3, 3), (4, 4, 3), (4, 4, 2), (3, 2, 2), (4, 4, 3)):
    wl = bench.Workload(rg, "base", 32, dev, 0, 32768)
    wl.model.base_lanes = base_lanes
    wl.model.max_inflight = inflight
    wl.model.slots = slots
    dt = wl.timed(16, 4, torch.cuda.synchronize)
    print("base_lanes %d max_inflight %d slots %d: %.2f ms per batch of 32, %.0f frames/s, lane streams %d" % (
        base_lanes, inflight, slots, dt / 16 * 1e3, 32 * 150 * 16 / dt, len(wl.model._lane_streams)), flush=True)
    del wl
    torch.cuda.empty_cache()
