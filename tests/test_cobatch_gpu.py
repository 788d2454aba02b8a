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
                                  inseq_noise_a=noise, in_seq_a=in_seq, fused_glue=fused)
        torch.cuda.synchronize()
        outs.append((x_all.clone(), out_b.clone()))
    assert torch.isfinite(outs[0][0]).all() and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])

