"""Debug: persistent forward under graph replay / concurrent lanes vs eager."""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
cfg = rg.synth.default_model_cfg(num_layers=2)
sch = rg.schedule.Schedule()
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, sch, "cuda")

def rel(a, b): return ((a - b).norm() / b.norm()).item()

B = 2
sessP = rg.denoiser.DenoiserSession(W, B, persistent=True)
sessC = rg.denoiser.DenoiserSession(W, B, persistent=False)
for trial in range(3):
    d = rg.synth.synth_batch(B, seed=10 + trial)
    mask = torch.ones(B, 43); mask[:, [10, 21, 32]] = 0
    qm = {c: torch.ones(B, 43) for c in rg.denoiser.CONDS}
    for s in (sessP, sessC):
        s.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, qm)
    x0 = torch.randn(B, 43, 512, device="cuda")
    outs = {}
    for name, s in (("P", sessP), ("C", sessC)):
        x = x0.clone()
        rg.sampler.ddim_sample_loop(s, x)
        torch.cuda.synchronize()
        outs[name] = x.clone()
    print("trial", trial, "eager loop P vs C", rel(outs["P"], outs["C"]), "abort", sessP.pf.aborted(), flush=True)

# graph: capture the 50-step loop once, replay 3 times with new inputs
xs = torch.empty(B, 43, 512, device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    rg.sampler.ddim_sample_loop(sessP, xs)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    with torch.cuda.graph(g, stream=st):
        rg.sampler.ddim_sample_loop(sessP, xs)
for trial in range(3):
    x0 = torch.randn(B, 43, 512, device="cuda")
    xs.copy_(x0)
    with torch.cuda.stream(st):
        g.replay()
    torch.cuda.synchronize()
    xr = xs.clone()
    x = x0.clone()
    rg.sampler.ddim_sample_loop(sessC, x)
    torch.cuda.synchronize()
    print("replay", trial, "graph P vs eager C", rel(xr, x), "abort", sessP.pf.aborted(), flush=True)

# two concurrent persistent loops on two streams (lanes)
sess2 = rg.denoiser.DenoiserSession(W, B, persistent=True)
sess2.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, qm)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for trial in range(3):
    xa_, xb_ = torch.randn(B, 43, 512, device="cuda"), torch.randn(B, 43, 512, device="cuda")
    ra, rb = xa_.clone(), xb_.clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        rg.sampler.ddim_sample_loop(sessP, ra)
    with torch.cuda.stream(s2):
        rg.sampler.ddim_sample_loop(sess2, rb)
    torch.cuda.synchronize()
    ca, cb = xa_.clone(), xb_.clone()
    rg.sampler.ddim_sample_loop(sessC, ca)
    rg.sampler.ddim_sample_loop(sessC, cb)
    torch.cuda.synchronize()
    print("concurrent", trial, rel(ra, ca), rel(rb, cb), "abort", sessP.pf.aborted(), sess2.pf.aborted(), flush=True)
