"""GPU: the HIP denoiser (bf16 MFMA GEMMs, fp32 everything else) against
  (a) the golden vectors produced by the real reference (fp32 CPU), and
  (b) the oracle run with bf16-rounded GEMM operands and exact LayerNorm on -1e6 rows,
      which is what the kernels implement -> tighter tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser as od, diffusion as odf

pytestmark = pytest.mark.gpu
KEEP = [r for r in range(43) if r not in (10, 20, 30)]
STEP_OF_T = {999: 49, 514: 34, 99: 7, 0: 0}


def relerr(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def setup(rg):
    assert torch.cuda.is_available()
    out = {}
    for L in (2, 8):
        cfg = rg.synth.default_model_cfg(num_layers=L)
        P = rg.synth.synth_denoiser_state(0, cfg)
        sch = rg.schedule.Schedule()
        W = rg.denoiser.DenoiserWeights(P, cfg, sch, "cuda")
        out[L] = (cfg, P, W)
    out["fp32"] = (cfg, P, rg.denoiser.DenoiserWeights(P, cfg, sch, "cuda", precision="fp32"))
    return out


def _inputs(rg, B=2):
    data = rg.synth.synth_batch(B, seed=1234)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(99)).standard_normal((B, 43, 512)).astype(np.float32))
    mm = torch.ones(B, 43)
    mm[:, [10, 21, 32]] = 0
    return data, x, mm


def _hip_x0(rg, W, sess, x, step):
    xd = x.cuda()
    sess.forward(xd, step)
    x0 = torch.empty_like(xd)
    sch = W.schedule
    sess.cfg_ddim(xd, torch.empty_like(xd), step, sch.c_prev_a[step], sch.c_prev_b[step], x0_out=x0)
    torch.cuda.synchronize()
    return x0.cpu()


def test_timestep_tables_exact(rg, setup):
    """The load-time emb tables are fp32-exact restatements of time_embed + emb_layers."""
    cfg, P, W = setup[2]
    ts = torch.tensor(W.schedule.timestep_map)
    emb = od.linear(P, "time_embed.2", torch.nn.functional.silu(od.linear(P, "time_embed.0", od.timestep_embedding(ts, 512))))
    for l in range(2):
        for bi, blk in enumerate(rg.denoiser.BLOCKS):
            ref = od.linear(P, "temporal_decoder_blocks.%d.%s.proj_out.emb_layers.1" % (l, blk), torch.nn.functional.silu(emb))
            assert (W.ss[:, l, bi].cpu() - ref).abs().max() <= 2e-5


@pytest.mark.parametrize("L,tag", [(2, "L2_allenc"), (8, "L8_encdec")])
def test_forward_vs_reference_golden(rg, parity, setup, golden_dir, L, tag):
    cfg, P, W = setup[L]
    g = np.load(os.path.join(golden_dir, "denoiser_%s.npz" % tag))
    data, x, mm = _inputs(rg)
    sess = rg.denoiser.DenoiserSession(W, 2)
    for masks in ("ones", "real"):
        qm = od.make_query_masks(mm) if masks == "real" else None
        sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, qm)
        for t, step in STEP_OF_T.items():
            x0 = _hip_x0(rg, W, sess, x, step)
            ref = torch.from_numpy(g["den_%s_t%d" % (masks, t)])
            rows = KEEP if masks == "real" else list(range(43))
            e = relerr(x0[:, rows], ref[:, rows])
            # bf16 GEMM operands vs the fp32 reference; the stated tolerance is 2e-2 relative on x0
            parity.check("denoiser %s masks=%s t=%d (bf16): x0 vs reference golden" % (tag, masks, t), e, 1.5e-2)


def test_precise_mode_vs_reference_golden(rg, parity, setup, golden_dir):
    """fp32-equivalent (bf16x3) GEMM mode, L=8, against the real reference's fp32 output.
    Without query masks every row must agree tightly; this is the structural parity check
    (all hoists/fusions are exact algebra)."""
    cfg, P, W = setup["fp32"]
    g = np.load(os.path.join(golden_dir, "denoiser_L8_encdec.npz"))
    data, x, mm = _inputs(rg)
    sess = rg.denoiser.DenoiserSession(W, 2)
    sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, None)
    for t, step in STEP_OF_T.items():
        x0 = _hip_x0(rg, W, sess, x, step)
        ref = torch.from_numpy(g["den_ones_t%d" % t])
        e, ea = relerr(x0, ref), (x0 - ref).abs().max().item()
        parity.check("denoiser L8 fp32 mode t=%d: x0 vs reference golden" % t, e, 5e-5)


def test_precise_mode_masked_rows_vs_exact_ln_oracle(rg, parity, setup):
    """With the reference's query masks, rows 10/20/30 go through LayerNorm(y - 1e6).  The kernels
    evaluate that LayerNorm exactly on the fp32-quantised row; the oracle does the same when
    masked_ln="exact" (torch's fp32 LayerNorm differs there by a platform-dependent rounding of
    a mean near -1e6)."""
    cfg, P, W = setup["fp32"]
    data, x, mm = _inputs(rg)
    sess = rg.denoiser.DenoiserSession(W, 2)
    qm = od.make_query_masks(mm)
    sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, qm)
    xf = od.encode_conditions(P, data["word"], data["audio"], data["speaker_ids"])
    od.OPTS.update(masked_ln="exact")
    try:
        for t, step in ((999, 49), (99, 7)):
            ref = od.denoiser_forward(P, cfg, x, torch.full((2,), t, dtype=torch.long), mm, xf, qm)
            x0 = _hip_x0(rg, W, sess, x, step)
            e = relerr(x0, ref)
            parity.check("denoiser L8 fp32 mode, real masks, t=%d: x0 vs exact-LN oracle (all 43 rows)" % t, e, 1e-4)
    finally:
        od.OPTS.update(masked_ln="torch")


def test_sample_loop_graph_matches_eager_and_oracle(rg, setup):
    """50-step base loop: graph replay == eager launches bit for bit; final latent vs fp32 oracle."""
    cfg, P, W = setup[2]
    data, x, mm = _inputs(rg)
    sess = rg.denoiser.DenoiserSession(W, 2)
    qm = od.make_query_masks(mm)
    sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, qm)
    xe = x.cuda()
    rg.sampler.ddim_sample_loop(sess, xe)
    torch.cuda.synchronize()
    xg = x.cuda()
    xs = xg.clone()
    loop = rg.sampler.GraphedLoop(lambda: rg.sampler.ddim_sample_loop(sess, xg))
    xg.copy_(xs)
    loop.replay()
    torch.cuda.synchronize()
    assert torch.equal(xg, xe)
    xf = od.encode_conditions(P, data["word"], data["audio"], data["speaker_ids"])
    model = lambda a, t: od.denoiser_forward(P, cfg, a, t, mm, xf, qm)
    ref = odf.ddim_sample_loop(odf.SpacedSchedule(), model, x, lambda s: torch.zeros(s))
    e = relerr(xe.cpu()[:, KEEP], ref[:, KEEP])
    print("50-step loop rel err vs fp32 oracle %.3e" % e)
    assert e <= 2e-2


@pytest.mark.parametrize("B,duo", [(1, False), (3, False), (11, False), (1, True), (3, True), (11, True)])
def test_seq_forward_matches_launch_chain_and_oracle(rg, parity, setup, B, duo):
    """The sequence-stationary forward (rg_seq_forward: one workgroup per sequence, one launch; duo: rg_seq2_forward, two
    sequences of a kind per workgroup) against the per-op launch chain on the same weights / conditions / masks (both bf16
    MFMA operands: they differ by where bf16 roundings fall) and against the fp32 oracle; every sequence of the batch is checked."""
    cfg, P, W = setup[8]
    data = rg.synth.synth_batch(B, seed=77)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal((B, 43, 512)).astype(np.float32))
    mm = torch.ones(B, 43)
    mm[:, [10, 21, 32]] = 0
    if B > 1:
        mm[1, 5:9] = 0   # a clip with masked motion tokens
    qm = od.make_query_masks(mm)
    outs = {}
    for engine in ("seq", "chain"):
        sess = rg.denoiser.DenoiserSession(W, B, engine=engine, seq_duo=duo)
        assert (sess.sq is not None) == (engine == "seq") and (engine != "seq" or sess.sq.duo == duo)
        sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, qm)
        for step in (49, 7):
            outs[engine, step] = sess.forward(x.cuda(), step).clone()
            torch.cuda.synchronize()
        if engine == "seq":   # replay: same inputs, same bits
            again = sess.forward(x.cuda(), 7).clone()
            torch.cuda.synchronize()
            assert torch.equal(again, outs["seq", 7])
    xf = od.encode_conditions(P, data["word"], data["audio"], data["speaker_ids"])
    for step, t in ((49, 999), (7, 99)):
        a, b = outs["seq", step].view(2 * B, 43, 512).cpu(), outs["chain", step].view(2 * B, 43, 512).cpu()
        assert torch.isfinite(a).all()
        for r in range(2 * B):
            e = relerr(a[r], b[r])
            assert e <= 1.5e-2, (B, step, r, e)
        print("B=%d step=%d seq vs launch chain: rel err %.3e" % (B, step, relerr(a, b)))
    # against the fp32 oracle (CFG-mixed x0, exact LayerNorm on the -1e6 rows like the kernels), all 43 rows
    sess = rg.denoiser.DenoiserSession(W, B, engine="seq", seq_duo=duo)
    sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, qm)
    x0 = _hip_x0(rg, W, sess, x, 7)
    od.OPTS.update(masked_ln="exact")
    try:
        ref = od.denoiser_forward(P, cfg, x, torch.full((B,), 99, dtype=torch.long), mm, xf, qm)
    finally:
        od.OPTS.update(masked_ln="torch")
    e = relerr(x0, ref)
    emax = ((x0 - ref).norm(dim=-1) / ref.norm(dim=-1)).max().item()
    tag = "two sequences per workgroup" if duo else "one sequence per workgroup"
    parity.check("seq forward B=%d, %s (bf16): x0 vs fp32 oracle, all rows" % (B, tag), e, 1e-2)
    parity.check("seq forward B=%d, %s (bf16): x0 vs fp32 oracle, worst token row" % (B, tag), emax, 3e-2)


@pytest.mark.parametrize("duo,split", [(False, 2), (True, 2), (True, 3), (True, 0), (True, 5)])
def test_seq_forward_two_step_groups(rg, setup, duo, split):
    """Clips [split, B) at another diffusion step in the same launch (the co-batched pipeline) == two separate forwards,
    bit for bit: a workgroup's arithmetic does not depend on its neighbours (duo: pairs are formed inside a step group; odd
    groups end in a lone sequence)."""
    cfg, P, W = setup[8]
    B = 5
    data = rg.synth.synth_batch(B, seed=78)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(6)).standard_normal((B, 43, 512)).astype(np.float32)).cuda()
    mm = torch.ones(B, 43)
    mm[:, [10, 21, 32]] = 0
    sess = rg.denoiser.DenoiserSession(W, B, engine="seq", seq_duo=duo)
    sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, od.make_query_masks(mm))
    both = sess.forward(x, 40, 9, split).clone().view(2, B, 43, 512)
    a = sess.forward(x, 40).clone().view(2, B, 43, 512)
    b = sess.forward(x, 9).clone().view(2, B, 43, 512)
    torch.cuda.synchronize()
    assert torch.equal(both[:, :split], a[:, :split]) and torch.equal(both[:, split:], b[:, split:])
    assert not torch.equal(a, b)


@pytest.mark.parametrize("B", [1, 5, 16, 64])
def test_seq_forward_one_workgroup_per_clip_is_bit_identical(rg, setup, B):
    """All launch forms of the sequence-stationary forward give the same bits: one workgroup per sequence (the reference
    form here), per clip, and rg_seq2_forward's two sequences of a kind per workgroup with the
    classifier-free pairs in workgroups of their own or behind the conditional ones.
    DenoiserSession(seq_pairs=True): B workgroups, each running a clip's conditional sequence and then its classifier-free
    twin, instead of 2 B workgroups (what the pipeline picks for launches that would not fit the chip beside the other batch
    lanes').  Same bits: with two step groups, real masks, and over consecutive forwards of one session (nothing of a pass survives into the next)."""
    cfg, P, W = setup[8]
    data = rg.synth.synth_batch(B, seed=80)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(8)).standard_normal((B, 43, 512)).astype(np.float32)).cuda()
    mm = torch.ones(B, 43)
    mm[:, [10, 21, 32]] = 0
    mm[0, 30:] = 0
    outs = {}
    for pairs, n in ((False, 1), (True, 1), (False, "duo"), (True, "duo")):
        duo = n == "duo"
        sess = rg.denoiser.DenoiserSession(W, B, engine="seq", seq_pairs=pairs, seq_duo=duo)
        assert sess.sq.args.pairs == int(pairs) and sess.sq.duo == duo
        sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, od.make_query_masks(mm))
        outs[pairs, n] = [sess.forward(x, st, sb, sp).clone() for st, sb, sp in ((49, None, None), (23, 40, max(1, B // 3)), (0, None, None))]
        torch.cuda.synchronize()
    for key in ((True, 1), (False, "duo"), (True, "duo")):
        for i, (a, b) in enumerate(zip(outs[key], outs[False, 1])):
            assert torch.isfinite(a).all() and torch.equal(a, b), (B, key, i, (a - b).abs().max().item())
    assert not torch.equal(outs[False, 1][0], outs[False, 1][2])


@pytest.mark.parametrize("B,pairs", [(5, False), (16, False), (6, True)])
def test_device_chosen_launch_form_is_bit_identical(rg, setup, B, pairs):
    """rg_seqx_forward behind rg_lane_form (include/rg_gesture.h): a lane-arbitrated session runs two sequences per workgroup
    while the other lanes' workgroups leave no room for one per sequence, and one per sequence when they do -- decided on the
    device, launch by launch; whichever form runs, the bits are rg_seq_forward's.  The state the lanes share says what ran:
    state[lane][0] = workgroups held, state[lane][1] = the form flag; chain_end() marks the lane idle."""
    cfg, P, W = setup[8]
    data = rg.synth.synth_batch(B, seed=81)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal((B, 43, 512)).astype(np.float32)).cuda()
    mm = torch.ones(B, 43)
    mm[:, [10, 21, 32]] = 0
    mm[B - 1, 28:] = 0
    cases = ((49, None, None), (23, 40, max(1, B // 3)), (0, None, None))
    ref_sess = rg.denoiser.DenoiserSession(W, B, engine="seq", seq_pairs=False, seq_duo=False)
    ref_sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, od.make_query_masks(mm))
    ref = [ref_sess.forward(x, *c).clone() for c in cases]
    n, lane, budget = 8, 3, 256
    state = torch.zeros(n, rg.seqfwd.LANE_STRIDE, device="cuda", dtype=torch.int32)
    sess = rg.denoiser.DenoiserSession(W, B, engine="seq", seq_pairs=pairs, seq_duo=True, lane_dyn=(state, lane, n, budget))
    sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, od.make_query_masks(mm))
    for others, want_wide in ((0, True), (budget - 2 * B + 1, False), (budget - 2 * B, True), (10 ** 6, False)):
        state.zero_()
        state[0, 0] = others                   # what another lane's launches hold
        for c, r in zip(cases, ref):
            out = sess.forward(x, *c).clone()
            torch.cuda.synchronize()
            assert torch.isfinite(out).all() and torch.equal(out, r), (B, pairs, others, c, (out - r).abs().max().item())
            st = state.cpu()
            sp = B if c[2] is None else c[2]
            npc = (sp + 1) // 2 + (B - sp + 1) // 2
            assert int(st[lane, 1]) == int(want_wide), (others, st[:, :2].tolist())
            assert int(st[lane, 0]) == (2 * B if want_wide else (npc if pairs else 2 * npc)), (others, st[:, :2].tolist())
        sess.chain_end()
        torch.cuda.synchronize()
        assert int(state[lane, 0]) == 0
    # a fixed-form session (one workgroup per sequence) only publishes its load
    state.zero_()
    one = rg.denoiser.DenoiserSession(W, B, engine="seq", seq_pairs=False, seq_duo=False, lane_dyn=(state, 1, n, budget))
    one.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, od.make_query_masks(mm))
    assert torch.equal(one.forward(x, 49), ref[0]) and int(state[1, 0]) == 2 * B
    one.chain_end()
    assert int(state[1, 0]) == 0


@pytest.mark.parametrize("mode", ["fused", "grouped", "per-layer"])
def test_condition_side_attention_matrices_vs_oracle(rg, parity, setup, mode):
    """DenoiserSession.set_conditions: A[layer][condition][clip][head] = softmax_N(K)^T V (efficient_attention.py:74-90) of the
    text / audio / speaker conditions at full depth against the fp32 oracle -- for the fused projection + reduction (rg_cond_kv:
    one launch per condition, K | V never leave the registers; the default), for the grouped projections it replaces (bf16
    normalised rows, the layers' LayerNorm affines folded into stacked [key | value] weights, DenoiserWeights.KV_GROUP layers per
    GEMM, then rg_kv_reduce per layer) and for the per-layer fp32-A GEMMs with a LayerNorm prologue of rounds 1-3."""
    grouped = mode != "per-layer"
    import torch.nn.functional as F
    cfg, P, W = setup[8]
    data, x, mm = _inputs(rg, B=3)
    sess = rg.denoiser.DenoiserSession(W, 3, kv_grouped=grouped, kv_fused=mode == "fused")
    sess.a_pre.fill_(float("nan"))
    sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, None)
    torch.cuda.synchronize()
    got = sess.a_pre.cpu()                                                  # [L, 3, B, H, 32, 32]
    assert torch.isfinite(got).all()
    # the sequence-stationary forward's bf16 fragments: written by rg_cond_kv itself in the fused mode, packed from a_pre otherwise
    assert sess.sq is not None and torch.equal(sess.sq.afrag, rg.seqfwd.a_fragments(sess.a_pre))
    H, D = cfg["num_heads"], cfg["latent_dim"]
    xf = {"xf_text": od.linear(P, "text_pre_proj", data["word"].float()), "xf_audio": od.linear(P, "audio_pre_proj", data["audio"].float()),
          "xf_spk": P["speaker_embedding.weight"][data["speaker_ids"].long()].float()}
    worst = 0.0
    for l in range(cfg["num_layers"]):
        for ci, c in enumerate(od.COND_ORDER):
            name = "temporal_decoder_blocks.%d.ca_blocks.%s" % (l, c)
            xn = od.layer_norm(P, name + ".text_norm", xf[c])
            B, N = xn.shape[:2]
            key = F.softmax(od.linear(P, name + ".key", xn).view(B, N, H, -1), dim=1)
            val = od.linear(P, name + ".value", xn).view(B, N, H, -1)
            ref = torch.einsum("bnhd,bnhl->bhdl", key, val)
            worst = max(worst, relerr(got[l, ci], ref))
    parity.check("condition-side A = softmax_N(K)^T V, L8, %s projections (bf16 operands): worst (layer, condition) vs oracle"
                 % mode, worst, 6e-3)
