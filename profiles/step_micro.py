"""Micro-benchmark: one denoiser forward step (8 layers, CFG rows) as a graph-replayed chain, per row count of the
guided workload (B clips -> M = 2 B 43 rows): microseconds per step and per kernel launch."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rg = importlib.import_module("rag-gesture_amd")
cfg = rg.synth.default_model_cfg(num_layers=8)
sch = rg.schedule.Schedule()
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, sch, "cuda")
STEPS = int(os.environ.get("STEPS", "10"))       # forward steps per captured graph
LN_MODE = os.environ.get("LN_MODE", "folded")
BS = (8, 16, 24, 48)
argv = sys.argv[1:]
if argv and argv[0].startswith("--B="):          # e.g. --B=16,48
    BS, argv = tuple(int(v) for v in argv[0][4:].split(",")), argv[1:]
ENGINES = argv or ["chain", "seq"]
for B, engine in [(b, e) for b in BS for e in ENGINES]:
    kw = {}
    # (round 2's chain+tile64 / chain+stylgemm variants: those session knobs were removed in round 5)
    kw = {}
    sess = rg.denoiser.DenoiserSession(W, B, engine=engine.split("+")[0], ln_mode=LN_MODE, **kw)
    d = rg.synth.synth_batch(B, seed=1)
    mask = torch.ones(B, 43)
    mask[:, [10, 21, 32]] = 0
    from_mask = {c: torch.ones(B, 43) for c in rg.denoiser.CONDS}
    sess.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, from_mask)
    x = torch.randn(B, 43, 512, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for s in range(STEPS):
            sess.forward(x, 40 - s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for s in range(STEPS):
                sess.forward(x, 40 - s)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / STEPS)
    print("B=%2d (M=%4d) %-10s: %.1f us per forward step  (graph of %d steps, ln_mode %s -> %s, ratio %s)"
          % (B, 2 * B * 43, engine, best, STEPS, LN_MODE, sess.ln_mode, sess.ln_ratio), flush=True)
