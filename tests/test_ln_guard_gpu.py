"""GPU: the folded LayerNorm under a residual-stream offset (ADVICE / VERDICT round 1).

The bf16 launch chain folds the attention LayerNorms into the consuming GEMM: rstd * (bf16(x) W'^T - mean c1) + c2 with x
UN-normalised, so the bf16 rounding of x scales with |row mean|, not with the row's spread.  A trained checkpoint can carry
a per-row offset in the residual stream; here it is injected through the joint-embedding bias (+ c on every column:
LayerNorm is shift invariant, so the exact result barely moves while every row sits c / std away from zero).
  * ln_mode="auto" must detect the offset on the session's first forward (rg_ln_guard) and use the LayerNorm pre-pass;
  * the pre-pass and the sequence-stationary forward (fp32 LayerNorm from fp32 rows by construction) stay at bf16 operand
    accuracy;
  * the folded form's error is reported next to them, and the guard stays silent on centred rows."""
import numpy as np
import pytest
import torch

from oracle import denoiser as od

pytestmark = pytest.mark.gpu
KEEP = [r for r in range(43) if r not in (10, 20, 30)]


def relerr(a, b):
    return ((a - b).norm() / b.norm()).item()


def _run(rg, W, sess, x, step):
    xd = x.cuda()
    sess.forward(xd, step)
    x0 = torch.empty_like(xd)
    sch = W.schedule
    sess.cfg_ddim(xd, torch.empty_like(xd), step, sch.c_prev_a[step], sch.c_prev_b[step], x0_out=x0)
    torch.cuda.synchronize()
    return x0.cpu()


@pytest.mark.parametrize("offset", [0.0, 8.0, 20.0])
def test_residual_offset(rg, offset):
    cfg = rg.synth.default_model_cfg(num_layers=2)
    P = dict(rg.synth.synth_denoiser_state(0, cfg))
    P["joint_embed.bias"] = P["joint_embed.bias"] + offset
    W = rg.denoiser.DenoiserWeights(P, cfg, rg.schedule.Schedule(), "cuda")
    B = 2
    data = rg.synth.synth_batch(B, seed=1234)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(99)).standard_normal((B, 43, 512)).astype(np.float32))
    mm = torch.ones(B, 43)
    mm[:, [10, 21, 32]] = 0
    qm = od.make_query_masks(mm)
    xf = od.encode_conditions(P, data["word"], data["audio"], data["speaker_ids"])
    od.OPTS.update(masked_ln="exact")
    try:
        ref = od.denoiser_forward(P, cfg, x, torch.full((B,), 99, dtype=torch.long), mm, xf, qm)
    finally:
        od.OPTS.update(masked_ln="torch")
    errs = {}
    for name, kw in (("folded", dict(engine="chain", ln_mode="folded")), ("prologue", dict(engine="chain", ln_mode="prologue")),
                     ("auto", dict(engine="chain", ln_mode="auto")), ("seq", dict(engine="seq"))):
        sess = rg.denoiser.DenoiserSession(W, B, **kw)
        sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, qm)
        errs[name] = relerr(_run(rg, W, sess, x, 7)[:, KEEP], ref[:, KEEP])
        if name == "auto":
            ratio = sess.ln_ratio
            print("offset %.0f: guard sees max |mean| / std = %.1f -> ln_mode %s" % (offset, ratio ** 0.5, sess.ln_mode))
            assert sess.ln_mode == ("folded" if offset == 0.0 else "prologue")
            # second forward of the settled session: same mode, no guard, same result as the explicit mode
            again = relerr(_run(rg, W, sess, x, 7)[:, KEEP], ref[:, KEEP])
            assert abs(again - errs["auto"]) <= 1e-6
    print("offset %.0f: rel err vs fp32 oracle  folded %.3e  prologue %.3e  auto %.3e  seq %.3e"
          % (offset, errs["folded"], errs["prologue"], errs["auto"], errs["seq"]))
    assert errs["prologue"] <= 2e-2 and errs["auto"] <= 2e-2 and errs["seq"] <= 2e-2
    if offset == 0.0:
        assert errs["folded"] <= 2e-2
