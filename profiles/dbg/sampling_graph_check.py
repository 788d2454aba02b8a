"""Why does a guided sampling step of the pipeline take longer than step_micro's forward?  Replays the pipeline's own
sampling graph alone, then a graph of forwards only on the same session, then the pieces of a step."""
import importlib, os, sys, time
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
graph, static, outs = m._graphs[key]
st = torch.cuda.Stream()


def timed(fn, n=3):
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print("pipeline sampling graph %s alone: %.2f ms = %.1f us per step" % (str(key)[:30], timed(graph.replay), timed(graph.replay) * 20))
sess = m._session(16, "sample", 0)
x = torch.randn(16, 43, 512, device=dev)
sampler = rg.sampler


def capture(fn):
    with torch.cuda.stream(st):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            fn()
    return g


g_fwd = capture(lambda: [sess.forward(x, 49 - i) for i in range(50)])
print("50 forwards on the pipeline's session: %.1f us per step" % (timed(g_fwd.replay) * 20))
g_loop = capture(lambda: sampler.ddim_sample_loop(sess, x))
print("ddim_sample_loop (forward + cfg_ddim): %.1f us per step" % (timed(g_loop.replay) * 20))
inv = torch.randn(50, 16, 43, 512, device=dev) * (torch.rand(50, 16, 43, 1, device=dev) > 0.5)
noise = torch.randn(50, 16, 43, 512, device=dev)
GI = [2] * 25 + [0] * 25
g_gl = capture(lambda: sampler.ddim_guided_sample_loop(sess, x, inv, GI, 0.1, noise))
print("ddim_guided_sample_loop: %.1f us per step" % (timed(g_gl.replay) * 20))
# fresh session, same weights, step_micro's conditions
sess2 = rg.denoiser.DenoiserSession(m.model.weights, 16)
d = rg.synth.synth_batch(16, seed=1)
mask = torch.ones(16, 43); mask[:, [10, 21, 32]] = 0
sess2.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, {c: torch.ones(16, 43) for c in rg.denoiser.CONDS})
g2 = capture(lambda: [sess2.forward(x, 49 - i) for i in range(50)])
print("50 forwards on a fresh session (pipeline weights): %.1f us per step" % (timed(g2.replay) * 20))
# the same graph on the pipeline's own lane streams and on the caller's stream
for name, s in [("lane stream 0", m._lane_streams[0]), ("lane stream 1", m._lane_streams[1]), ("current stream", torch.cuda.current_stream())]:
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record(); graph.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("pipeline sampling graph on %s: %.2f ms" % (name, best))
# and inside a pipeline step: events around the replay
orig = m._graph_run
log = []
def traced(key, inputs, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(key, inputs, fn); e1.record()
    log.append((key, e0, e1))
    return out
m._graph_run = traced
wl.step(); torch.cuda.synchronize()
for key, e0, e1 in log:
    print("   in-step %-40s %.2f ms" % (str(key)[:40], e0.elapsed_time(e1)))
