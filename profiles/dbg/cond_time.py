"""Device time of DenoiserSession.set_conditions (64 clips: 16 of a batch + 48 exemplars, as the pipeline's shared sessions) alone
on the chip, graph replay: fused projection + reduction (rg_cond_kv) against the grouped GEMMs + rg_kv_reduce."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
cfg = rg.synth.default_model_cfg(num_layers=8)
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, rg.schedule.Schedule(), "cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in rg.synth.synth_batch(B, seed=1).items()}
mask = torch.ones(B, 43, device="cuda")
outs = {}
for name, kw in (("fused", dict(kv_fused=True)), ("grouped + kv_reduce", dict(kv_fused=False))):
    sess = rg.denoiser.DenoiserSession(W, B, engine="seq", **kw)
    run = lambda: sess.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, None)
    run(); torch.cuda.synchronize()
    outs[name] = sess.a_pre.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
        with rg.capi.capture(g):
            run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): g.replay()
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("set_conditions B=%d %-22s %.3f ms per call (graph replay, alone on the chip)" % (B, name, e0.elapsed_time(e1) / 10), flush=True)
a, b = outs["fused"], outs["grouped + kv_reduce"]
print("fused vs grouped: rel diff %.3e, max abs %.3e" % (((a - b).norm() / b.norm()).item(), (a - b).abs().max().item()))
