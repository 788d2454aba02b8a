"""GPU: sampler elementwise HIP kernels (through the C ABI) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import diffusion as odf

pytestmark = pytest.mark.gpu


def _rand(shape, seed):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32))


@pytest.fixture(scope="module")
def env(rg):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return rg.capi.get_handle(0), rg.schedule.Schedule(), odf.SpacedSchedule(), rg.capi.I64


@pytest.mark.parametrize("i", [49, 37, 25, 1, 0])
def test_ddim_update_and_reverse(env, i):
    h, sch, osch, I64 = env
    x, x0 = _rand((3, 43, 512), 1), _rand((3, 43, 512), 2)
    ref, _ = odf.ddim_sample(osch, lambda a, t: x0, x, i, lambda s: torch.zeros(s))
    xd, x0d = x.cuda(), x0.cuda()
    out = torch.empty_like(xd)
    h.call("ddim_update", xd, x0d, out, I64(xd.numel()), float(sch.c_recip[i]), float(sch.c_recipm1[i]),
           float(sch.c_prev_a[i]), float(sch.c_prev_b[i]))
    assert torch.equal(out.cpu(), ref)  # bit-exact: same fp32 op order, no FMA contraction
    ref = odf.ddim_reverse_sample(osch, lambda a, t: x0, x, i)
    h.call("ddim_update", xd, x0d, out, I64(xd.numel()), float(sch.c_recip[i]), float(sch.c_recipm1[i]),
           float(sch.c_next_a[i]), float(sch.c_next_b[i]))
    d = (out.cpu() - ref).abs()
    print("step", i, "mismatches", int((d > 0).sum()), "max", float(d.max()))
    assert float(d.max()) <= 2e-6


def test_inseq_replace_and_guidance(env):
    h, sch, osch, I64 = env
    B, T, D, i = 4, 43, 512, 30
    x, noise = _rand((B, T, D), 3), _rand((B, T, D), 4)
    in_seq = torch.zeros(B, T, D)
    in_seq[0, 1:4] = _rand((3, D), 5)
    in_seq[0, 12:15] = _rand((3, D), 6)
    in_seq[2, 7:9] = _rand((2, D), 7)
    in_seq[3, 0, 17] = 0.5  # a single non-zero element marks the whole row
    nz = (in_seq != 0).any(-1, keepdim=True).float()
    ref = x * (1 - nz) + odf.q_sample(osch, in_seq, i, noise) * nz
    xd = x.cuda()
    h.call("inseq_replace", xd, in_seq.cuda(), noise.cuda(), B * T, D, float(sch.s_ab[i]), float(sch.s_1mab[i]))
    assert torch.equal(xd.cpu(), ref)
    for g_iter in (0, 1, 7, 24):
        ref = odf.retrieval_guidance_update(x, in_seq, g_iter, 0.1)
        xd = x.cuda()
        h.call("guidance_update", xd, in_seq.cuda(), B * T, D, g_iter, 0.1)
        assert (xd.cpu() - ref).abs().max() <= 1e-6
        assert torch.equal(xd.cpu() * (1 - nz), x * (1 - nz))  # unmasked rows untouched


def test_guidance_matches_autograd(env):
    """The closed-form gradient equals what the reference obtains with autograd
    (gaussian_diffusion.py:1351-1378)."""
    x = _rand((2, 43, 512), 8)
    in_seq = torch.zeros(2, 43, 512)
    in_seq[1, 3:6] = _rand((3, 512), 9)
    mask = (in_seq != 0).any(-1)
    s = x.clone().requires_grad_(True)
    for _ in range(5):
        loss = torch.nn.functional.mse_loss(s * mask.unsqueeze(-1).float(), in_seq)
        gr = torch.autograd.grad(loss, [s], retain_graph=True)[0]
        s = s - 0.1 * gr
    assert (odf.retrieval_guidance_update(x, in_seq, 5, 0.1) - s.detach()).abs().max() <= 1e-6


def test_cfg_ddim_and_splice(env):
    h, sch, osch, I64 = env
    B, T, D, i = 3, 43, 512, 20
    out2, x = _rand((2 * B, T, D), 10), _rand((B, T, D), 11)
    js = torch.ones(T)
    js[11:21] = 1.25
    w_c, w_u = sch.cfg_weights(dict(coarse_scale=6.5, both_coef=0.52351, text_coef=-0.28419, retr_coef=2.39872), i)
    jt = js.view(1, T, 1)
    x0 = out2[:B] * w_c * jt + out2[B:] * w_u * (1 / jt)
    ref, _ = odf.ddim_sample(osch, lambda a, t: x0, x, i, lambda s: torch.zeros(s))
    xo, x0o = torch.empty(B, T, D, device="cuda"), torch.empty(B, T, D, device="cuda")
    h.call("cfg_ddim_update", out2.cuda(), x.cuda(), xo, x0o, js.cuda(), B, T, D, w_c, w_u, float(sch.c_recip[i]),
           float(sch.c_recipm1[i]), float(sch.c_prev_a[i]), float(sch.c_prev_b[i]))
    assert (x0o.cpu() - x0).abs().max() <= 1e-5
    assert (xo.cpu() - ref).abs().max() <= 1e-4
    src, dst = _rand((2, T, D), 12), _rand((3, T, D), 13)
    exp = dst.clone()
    exp[2, 1:4] = src[1, 2:5]
    exp[2, 12:15] = src[1, 13:16]
    dd = dst.cuda()
    h.call("splice_rows", src.cuda(), dd, T, D, 10, 1, 2, 2, 5, 1, 4)
    assert torch.equal(dd.cpu(), exp)


@pytest.mark.parametrize("guided", [True, False])
def test_splice_many_equals_the_splices_one_by_one(env, guided):
    """rg_splice_many (every exemplar of a batch in one launch: level `lvl` into the start noise, all levels into the guidance
    target) against rg_splice_rows / rg_splice_rows_rep per exemplar (diffusion_architecture.py:386-407), bit for bit, with
    entries of different lengths, two exemplars in one clip and an empty entry."""
    import ctypes
    import importlib
    rg = importlib.import_module("rag-gesture_amd")
    h = env[0]
    S, Ep, B, T, D, n_lat, lvl = 6, 5, 4, 43, 512, 10, 4
    inv = _rand((S, Ep, T, D), 21).cuda()
    entries = [(0, 2, 1, 0, 3), (1, 2, 5, 6, 2), (3, 0, 0, 7, 3), (4, 3, 2, 2, 0), (2, 1, 9, 9, 1)]      # (e, b, r0, q0, nrows)
    sn_ref, sn = _rand((B, T, D), 22).cuda(), None
    invl_ref = _rand((S, B, T, D), 23).cuda() if guided else None
    sn, invl = sn_ref.clone(), (invl_ref.clone() if guided else None)
    for e, b, r0, q0, n in entries:
        h.call("splice_rows", inv[lvl], sn_ref, T, D, n_lat, e, b, r0, r0 + n, q0, q0 + n)
        if guided:
            h.call("splice_rows_rep", inv, invl_ref, T, D, n_lat, e, b, r0, r0 + n, q0, q0 + n, S, Ep, B)
    tab = rg.sampler.SpliceTable()
    tab.n = len(entries)
    for i, (e, b, r0, q0, n) in enumerate(entries):
        tab.e[i], tab.b[i], tab.r0[i], tab.q0[i], tab.nrows[i] = e, b, r0, q0, n
    rc = h.lib.rg_splice_many(h._h, ctypes.byref(tab), inv.data_ptr(), sn.data_ptr(), None if invl is None else invl.data_ptr(),
                              T, D, n_lat, lvl, S, Ep, B, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(sn, sn_ref) and (not guided or torch.equal(invl, invl_ref))
    assert not torch.equal(sn, _rand((B, T, D), 22).cuda())

