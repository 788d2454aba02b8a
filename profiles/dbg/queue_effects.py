"""Does a graph replay slow down (a) when it is launched onto a stream that is still blocked by an event, (b) while another
hardware queue holds a (tiny, 1-workgroup) running kernel?  Sampling graph of the pipeline, B = 16."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
wl = bench.Workload(rg, "guided", 16, dev, 0, 32768)
m = wl.model
m.sample_lanes = 1
for _ in range(2):
    wl.step()
torch.cuda.synchronize()
key = [k for k in m._graphs if k[0] == "guided"][0]
graph = m._graphs[key][0]
s0, s1 = m._lane_streams[0], m._lane_streams[1]
spin = lambda ms: torch.cuda._sleep(int(ms * 2.4e6))     # cycles at ~2.4 GHz: a single-workgroup busy kernel


def run(name, before=None, beside=None):
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if before is not None:
            with torch.cuda.stream(s1):
                spin(before)
                ev = s1.record_event()
            s0.wait_event(ev)
        if beside is not None:
            with torch.cuda.stream(s1):
                spin(beside)
        with torch.cuda.stream(s0):
            e0.record(); graph.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("%-70s %.2f ms" % (name, best), flush=True)


run("alone")
run("launched while its stream waits 30 ms for an event of another stream", before=30)
run("beside a 1-workgroup spin kernel on another stream (80 ms)", beside=80)
run("alone")
