"""sampler.cobatched_loop: the guided sampling loop of one batch and the DDIM inversion of another batch's exemplars advanced
by the SAME denoiser launches (two step groups per forward) must give what the two separate loops give."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rg():
    return importlib.import_module("rag-gesture_amd")


def relerr(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("n_a,n_b", [(2, 3), (8, 24)])
def test_cobatched_loop_equals_separate_loops(rg, n_a, n_b):
    L = 2
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
    mb[0, 5:9] = 0
    qa = {c: (torch.arange(T)[None, :].expand(n_a, T) % 10 != 0).float() for c in rg.denoiser.CONDS}
    qb = {c: (torch.arange(T)[None, :].expand(n_b, T) % 10 != 0).float() for c in rg.denoiser.CONDS}
    xa0, xb0 = rnd(n_a, T, D), rnd(n_b, T, D)
    inverted = rnd(S, n_a, T, D) * (torch.rand(S, n_a, T, 1, device="cuda") > 0.6)
    noise = rnd(S, n_a, T, D)
    GI = [2] * 25 + [0] * 25
    # --- separate loops (the existing path)
    sa = rg.denoiser.DenoiserSession(W, n_a, ln_mode="folded", engine="chain")
    sa.set_conditions(da["word"], da["audio"], da["speaker_ids"], ma, qa)
    ref_a = rg.sampler.ddim_guided_sample_loop(sa, xa0.clone(), inverted, GI, 0.1, noise)
    sb = rg.denoiser.DenoiserSession(W, n_b, ln_mode="folded", engine="chain")
    sb.set_conditions(db["word"], db["audio"], db["speaker_ids"], mb, qb)
    ref_b = rg.sampler.ddim_reverse_sample_loop(sb, xb0.clone(), torch.empty(S, n_b, T, D, device="cuda"))
    # --- one session holding both, filled in two calls
    sc = rg.denoiser.DenoiserSession(W, n_a + n_b, ln_mode="folded", engine="chain")
    sc.set_conditions(da["word"], da["audio"], da["speaker_ids"], ma, qa, offset=0, finalize=False)
    sc.set_conditions(db["word"], db["audio"], db["speaker_ids"], mb, qb, offset=n_a)
    x_all = torch.cat([xa0, xb0]).contiguous()
    out_b = torch.empty(S, n_b, T, D, device="cuda")
    rg.sampler.cobatched_loop(sc, x_all, n_a, out_b, inverted_a=inverted, guidance_iters=GI, guidance_lr=0.1, inseq_noise_a=noise)
    torch.cuda.synchronize()
    ea, eb = relerr(x_all[:n_a], ref_a), relerr(out_b, ref_b)
    print("co-batched vs separate: sampling %.3e, inversion levels %.3e (exact: %s, %s)"
          % (ea, eb, torch.equal(x_all[:n_a], ref_a), torch.equal(out_b, ref_b)))
    # rows never mix: the only differences can come from a different kernel variant at the larger row count
    assert ea <= 2e-3 and eb <= 2e-3
    assert relerr(x_all[n_a:], ref_b[S - 1]) <= 2e-3


@pytest.mark.parametrize("guided,with_in_seq", [(True, False), (True, True), (False, True), (False, False)])
def test_fused_glue_equals_the_four_launches(rg, guided, with_in_seq):
    """rg_cobatch_glue (one launch between two forwards: this step's two CFG + DDIM updates, the next step's guidance update
    and in-sequence replacement) against the four launches it replaces, bit for bit, over whole co-batched loops: guided and
    plain sampling, with and without a first-step in_seq."""
    L = 2
    cfg = rg.synth.default_model_cfg(num_layers=L)
    sch = rg.schedule.Schedule()
    W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, sch, "cuda", precision="bf16")
    S, T, D, n_a, n_b = sch.num_timesteps, 43, 512, 3, 5
    g = np.random.Generator(np.random.PCG64(11))
    rnd = lambda *s: torch.from_numpy(g.standard_normal(s).astype(np.float32)).cuda()
    da, db = rg.synth.synth_batch(n_a, seed=5), rg.synth.synth_batch(n_b, seed=6)
    ma, mb = torch.ones(n_a, T), torch.ones(n_b, T)
    qa = {c: torch.ones(n_a, T) for c in rg.denoiser.CONDS}
    qb = {c: torch.ones(n_b, T) for c in rg.denoiser.CONDS}
    xa0, xb0 = rnd(n_a, T, D), rnd(n_b, T, D)
    inverted = (rnd(S, n_a, T, D) * (torch.rand(S, n_a, T, 1, device="cuda") > 0.6)) if guided else None
    in_seq = (rnd(n_a, T, D) * (torch.rand(n_a, T, 1, device="cuda") > 0.5)) if with_in_seq else None
    noise = rnd(S, n_a, T, D) if (guided or with_in_seq) else None
    GI = [2] * 25 + [0] * 25
    outs = []
    for fused in (True, False):
        sc = rg.denoiser.DenoiserSession(W, n_a + n_b, engine="seq")
        sc.set_conditions(da["word"], da["audio"], da["speaker_ids"], ma, qa, offset=0, finalize=False)
        sc.set_conditions(db["word"], db["audio"], db["speaker_ids"], mb, qb, offset=n_a)
        x_all = torch.cat([xa0, xb0]).contiguous()
        out_b = torch.empty(S, n_b, T, D, device="cuda")
        rg.sampler.cobatched_loop(sc, x_all, n_a, out_b, inverted_a=inverted, guidance_iters=GI, guidance_lr=0.1,
                                  inseq_noise_a=noise, in_seq_a=in_seq, fused_glue=fused, tail_glue=False)
        torch.cuda.synchronize()
        outs.append((x_all.clone(), out_b.clone()))
    assert torch.isfinite(outs[0][0]).all() and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])



@pytest.mark.parametrize("form", [dict(seq_duo=True), dict(seq_duo=False), dict(seq_pairs=True, seq_duo=True), dict(seq_pairs=True, seq_duo=False)])
def test_tail_glue_equals_the_glue_launch(rg, form):
    """A co-batched loop whose forwards END with the step's update (rg_seq_args.glue_ctr: the workgroup that finishes the second
    of a clip's two sequences updates the clip, csrc/rg_tail.h) against the same loop with rg_cobatch_glue between the forwards:
    bit for bit, for every launch form of the two denoiser kernels (one or two sequences per workgroup, the classifier-free
    sequences in workgroups of their own or behind the conditional ones), odd group sizes (lone sequences), guided sampling with
    a first-step in_seq; three runs of the tail form (the arrival order of the workgroups differs from run to run)."""
    L = 2
    cfg = rg.synth.default_model_cfg(num_layers=L)
    sch = rg.schedule.Schedule()
    W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, sch, "cuda", precision="bf16")
    S, T, D, n_a, n_b = sch.num_timesteps, 43, 512, 5, 11
    g = np.random.Generator(np.random.PCG64(12))
    rnd = lambda *s: torch.from_numpy(g.standard_normal(s).astype(np.float32)).cuda()
    da, db = rg.synth.synth_batch(n_a, seed=5), rg.synth.synth_batch(n_b, seed=6)
    ma, mb = torch.ones(n_a, T), torch.ones(n_b, T)
    qa = {c: torch.ones(n_a, T) for c in rg.denoiser.CONDS}
    qb = {c: torch.ones(n_b, T) for c in rg.denoiser.CONDS}
    xa0, xb0 = rnd(n_a, T, D), rnd(n_b, T, D)
    inverted = rnd(S, n_a, T, D) * (torch.rand(S, n_a, T, 1, device="cuda") > 0.6)
    in_seq = rnd(n_a, T, D) * (torch.rand(n_a, T, 1, device="cuda") > 0.5)
    noise = rnd(S, n_a, T, D)
    GI = [2] * 25 + [0] * 25
    sc = rg.denoiser.DenoiserSession(W, n_a + n_b, engine="seq", **form)
    sc.set_conditions(da["word"], da["audio"], da["speaker_ids"], ma, qa, offset=0, finalize=False)
    sc.set_conditions(db["word"], db["audio"], db["speaker_ids"], mb, qb, offset=n_a)
    outs = []
    for tail in (False, True, True, True):
        x_all = torch.cat([xa0, xb0]).contiguous()
        out_b = torch.empty(S, n_b, T, D, device="cuda")
        rg.sampler.cobatched_loop(sc, x_all, n_a, out_b, inverted_a=inverted, guidance_iters=GI, guidance_lr=0.1,
                                  inseq_noise_a=noise, in_seq_a=in_seq, tail_glue=tail)
        torch.cuda.synchronize()
        assert sc.sq.args.glue_ctr is None or tail
        outs.append((x_all.clone(), out_b.clone()))
    assert torch.isfinite(outs[0][0]).all()
    for k in (1, 2, 3):
        assert torch.equal(outs[0][0], outs[k][0]) and torch.equal(outs[0][1], outs[k][1]), "run %d of the tail form differs" % k
    # a forward without the tail leaves x alone again
    x_keep = outs[0][0].clone()
    sc.forward(x_keep, 3)
    torch.cuda.synchronize()
    assert torch.equal(x_keep, outs[0][0])


@pytest.mark.parametrize("form", [dict(seq_duo=False), dict(seq_duo=True), dict(seq_pairs=True, seq_duo=False)])
def test_single_loops_with_tail_equal_the_launch_pairs(rg, form):
    """ddim_sample_loop (with an in_seq), ddim_guided_sample_loop and ddim_reverse_sample_loop with the step's update at the end
    of the forward (one launch per step) against forward + update launches: bit for bit, twice."""
    cfg = rg.synth.default_model_cfg(num_layers=2)
    sch = rg.schedule.Schedule()
    W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, sch, "cuda", precision="bf16")
    S, T, D, B = sch.num_timesteps, 43, 512, 5
    g = np.random.Generator(np.random.PCG64(13))
    rnd = lambda *s: torch.from_numpy(g.standard_normal(s).astype(np.float32)).cuda()
    d = rg.synth.synth_batch(B, seed=7)
    m = torch.ones(B, T)
    sc = rg.denoiser.DenoiserSession(W, B, engine="seq", **form)
    sc.set_conditions(d["word"], d["audio"], d["speaker_ids"], m, {c: torch.ones(B, T) for c in rg.denoiser.CONDS})
    x0 = rnd(B, T, D)
    inverted = rnd(S, B, T, D) * (torch.rand(S, B, T, 1, device="cuda") > 0.6)
    in_seq = rnd(B, T, D) * (torch.rand(B, T, 1, device="cuda") > 0.5)
    noise = rnd(S, B, T, D)
    GI = [3] * 20 + [0] * 30
    res = {}
    try:
        for tail in (False, True, True):
            rg.sampler.TAIL_GLUE = tail
            xs = rg.sampler.ddim_sample_loop(sc, x0.clone(), in_seq, noise)
            xg = rg.sampler.ddim_guided_sample_loop(sc, x0.clone(), inverted, GI, 0.1, noise, in_seq=in_seq)
            xg0 = rg.sampler.ddim_guided_sample_loop(sc, x0.clone(), inverted, GI, 0.1, noise)
            xr = x0.clone()
            lv = rg.sampler.ddim_reverse_sample_loop(sc, xr, torch.empty(S, B, T, D, device="cuda"))
            torch.cuda.synchronize()
            res.setdefault(tail, []).append((xs, xg, xg0, xr, lv))
    finally:
        rg.sampler.TAIL_GLUE = True
    ref = res[False][0]
    assert all(torch.isfinite(t).all() for t in ref) and torch.equal(ref[3], ref[4][S - 1])
    for run in res[True]:
        for name, a, b in zip(("sample", "guided+in_seq", "guided", "inverted x", "levels"), ref, run):
            assert torch.equal(a, b), name


def test_tail_arguments_are_checked(rg):
    """glue_ctr with a glue that does not cover the session's clips is refused by the entry points (and by the host wrapper)."""
    import ctypes
    cfg = rg.synth.default_model_cfg(num_layers=1)
    W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, rg.schedule.Schedule(), "cuda", precision="bf16")
    B, T, D = 2, 43, 512
    d = rg.synth.synth_batch(B, seed=3)
    x = torch.zeros(B, T, D, device="cuda")
    for form in (dict(seq_duo=False), dict(seq_duo=True)):
        sc = rg.denoiser.DenoiserSession(W, B, engine="seq", **form)
        sc.set_conditions(d["word"], d["audio"], d["speaker_ids"], torch.ones(B, T), {c: torch.ones(B, T) for c in rg.denoiser.CONDS})
        g = rg.sampler._glue_sampling(sc, x, 5, None, None, 0, 0.0)
        g.n_a = 1                                                  # one clip without an update
        with pytest.raises(rg.capi.RgError):
            sc.forward(x, 5, glue=g)
        a = sc.sq.args
        a.x, a.step, a.step_b, a.split = x.data_ptr(), 5, 5, B
        a.glue, a.glue_ctr = g, sc.sq.glue_ctr.data_ptr()
        rc = sc.sq._fn(sc.h._h, ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc != 0 and b"glue" in sc.h.lib.rg_last_error(sc.h._h)
        a.glue_ctr = None
    torch.cuda.synchronize()
