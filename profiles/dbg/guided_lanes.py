"""Experiment: guided workload (config 3) with whole batches rotating over N lanes (N x 128 sequences in flight)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
db = None
for lanes, inflight, slots in ((2, 2, 2), (3, 3, 2), (3, 4, 3), (4, 4, 2)):
    wl = bench.Workload(rg, "guided", 16, dev, 0, 32768, database=db)
    db = wl.database
    wl.model.lanes = lanes
    wl.model.max_inflight = inflight
    wl.model.slots = slots
    dt = wl.timed(24, 6, torch.cuda.synchronize)
    lat = wl.latency_ms()
    print("lanes %d max_inflight %d slots %d: %.2f ms per batch of 16, %.0f frames/s, latency %s, lane streams %d" % (
        lanes, inflight, slots, dt / 24 * 1e3, 16 * 150 * 24 / dt, lat, len(wl.model._lane_streams)), flush=True)
    del wl
    torch.cuda.empty_cache()
