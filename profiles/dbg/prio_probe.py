"""Two lanes (B = 32 each) replaying forward graphs while a third stream replays a small-kernel graph (B = 2 forwards) beside
them: lane time per step with the third stream at normal and at low priority."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
rg = importlib.import_module("rag-gesture_amd")
print("stream priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
cfg = rg.synth.default_model_cfg(num_layers=8)
sch = rg.schedule.Schedule()
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, sch, "cuda")


def graph_for(B, st, steps):
    sess = rg.denoiser.DenoiserSession(W, B, ln_mode="folded")
    d = rg.synth.synth_batch(B, seed=1)
    mask = torch.ones(B, 43); mask[:, [10, 21, 32]] = 0
    sess.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, {c: torch.ones(B, 43) for c in rg.denoiser.CONDS})
    x = torch.randn(B, 43, 512, device="cuda")
    with torch.cuda.stream(st):
        sess.forward(x, 40)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for s in range(steps):
                sess.forward(x, 40 - s)
    return g, sess, x


lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
lane_graphs = [graph_for(32, st, 10) for st in lanes]
for prio, name in ((None, "no third stream"), (0, "third stream, normal priority"), (1, "third stream, priority +1"), (-1, "third stream, priority -1")):
    side = torch.cuda.Stream(priority=prio) if prio is not None else None
    sg = graph_for(2, side, 12) if side is not None else None
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in lanes:
            st.wait_event(e0)
        if side is not None:
            side.wait_event(e0)
            with torch.cuda.stream(side):
                sg[0].replay()
        for (g, _, _), st in zip(lane_graphs, lanes):
            with torch.cuda.stream(st):
                g.replay()
        for st in lanes:
            torch.cuda.current_stream().wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 100)
    print("%-34s lanes: %.1f us per forward step of the pair" % (name, best), flush=True)
