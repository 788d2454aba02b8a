"""Dynamic launch forms inside a co-batched loop: the loop on a lane-arbitrated session (narrow for the first steps, wide when the
other lane goes idle, decided on the device) against the same loop on a fixed one-sequence-per-workgroup session: bit for bit.
python profiles/dbg/dyn_loop_check.py [n_a n_b L]"""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
n_a, n_b, L = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 48, 8)
cfg = rg.synth.default_model_cfg(num_layers=L)
sch = rg.schedule.Schedule()
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, sch, "cuda", precision="bf16")
S, T, D = sch.num_timesteps, 43, 512
g = np.random.Generator(np.random.PCG64(9))
rnd = lambda *s: torch.from_numpy(g.standard_normal(s).astype(np.float32)).cuda()
da, db = rg.synth.synth_batch(n_a, seed=3), rg.synth.synth_batch(n_b, seed=4)
ma, mb = torch.ones(n_a, T), torch.ones(n_b, T)
ma[:, [10, 21, 32]] = 0
mb[:, [10, 21, 32]] = 0
qa = {c: (torch.arange(T)[None, :].expand(n_a, T) % 10 != 0).float() for c in rg.denoiser.CONDS}
qb = {c: (torch.arange(T)[None, :].expand(n_b, T) % 10 != 0).float() for c in rg.denoiser.CONDS}
xa0, xb0 = rnd(n_a, T, D), rnd(n_b, T, D)
inverted = rnd(S, n_a, T, D) * (torch.rand(S, n_a, T, 1, device="cuda") > 0.6)
noise = rnd(S, n_a, T, D)
GI = [2] * 25 + [0] * 25
n, lane = 8, 2
state = torch.zeros(n, rg.seqfwd.LANE_STRIDE, device="cuda", dtype=torch.int32)


def run(kw, flip=None, graph=False):
    sc = rg.denoiser.DenoiserSession(W, n_a + n_b, engine="seq", **kw)
    sc.set_conditions(da["word"], da["audio"], da["speaker_ids"], ma, qa, offset=0, finalize=False)
    sc.set_conditions(db["word"], db["audio"], db["speaker_ids"], mb, qb, offset=n_a)
    x_all = torch.cat([xa0, xb0]).contiguous()
    out_b = torch.empty(S, n_b, T, D, device="cuda")
    forms = []
    if flip is not None:           # eager loop, the other lane's load changes at step `flip`
        orig = sc.forward
        cnt = [0]

        def fwd(x, step, step_b=None, split=None):
            state[0, 0] = 10 ** 6 if cnt[0] < flip else 0
            r = orig(x, step, step_b, split)
            forms.append(int(state[lane, 1]))
            cnt[0] += 1
            return r
        sc.forward = fwd
    fn = lambda: rg.sampler.cobatched_loop(sc, x_all, n_a, out_b, inverted_a=inverted, guidance_iters=GI, guidance_lr=0.1, inseq_noise_a=noise)
    if graph:
        x_keep = x_all.clone()
        loop = rg.sampler.GraphedLoop(fn)
        for others in (10 ** 6, 0):
            x_all.copy_(x_keep)
            state.zero_(); state[0, 0] = others
            loop.replay()
            torch.cuda.synchronize()
            forms.append(others)
            yield x_all[:n_a].clone(), out_b.clone(), "graph others=%d" % others
        return
    fn()
    torch.cuda.synchronize()
    yield x_all[:n_a].clone(), out_b.clone(), "forms %s" % "".join(str(f) for f in forms)


(ref_a, ref_b, _), = run(dict(seq_pairs=False, seq_duo=False))
bad = 0
for tag, kw, flip, graph in (("duo fixed", dict(seq_duo=True), None, False),
                             ("dyn all wide", dict(seq_duo=True, lane_dyn=(state, lane, n, 256)), 0, False),
                             ("dyn all narrow", dict(seq_duo=True, lane_dyn=(state, lane, n, 256)), 10 ** 9, False),
                             ("dyn flip at 20", dict(seq_duo=True, lane_dyn=(state, lane, n, 256)), 20, False),
                             ("dyn graph", dict(seq_duo=True, lane_dyn=(state, lane, n, 256)), None, True)):
    state.zero_()
    for a, b, info in run(kw, flip, graph):
        ok = torch.equal(a, ref_a) and torch.equal(b, ref_b)
        bad += not ok
        print("%-16s %-60s sampling equal %s  inversion levels equal %s  (max diff %.3e / %.3e)"
              % (tag, info[:60], torch.equal(a, ref_a), torch.equal(b, ref_b), (a - ref_a).abs().max().item(), (b - ref_b).abs().max().item()), flush=True)
print("MISMATCHES: %d" % bad)
