"""GPU: fused bf16-MFMA GEMM (rg_gemm) against a plain PyTorch fp32 reference of the same op,
fed the same bf16-rounded operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.Generator(np.random.PCG64(seed)).standard_normal(shape) * scale).astype(np.float32))


def bf(x):
    return x.bfloat16().float()


@pytest.fixture(scope="module")
def h(rg):
    assert torch.cuda.is_available()
    return rg.capi.get_handle(0)


@pytest.mark.parametrize("M,N,K", [(86, 512, 512), (2752, 1536, 512), (300, 1024, 2048), (129, 512, 768)])
def test_plain_bias_fp32_A(rg, h, M, N, K):
    G = rg.gemm
    a, w, b = _rand((M, K), 1), _rand((N, K), 2, 0.05), _rand((N,), 3)
    out = torch.empty(M, N, device="cuda")
    G.gemm(h, M=M, N=N, K=K, W=G.pack_weight(w, "cuda"), out=out, segs=[G.Seg(a.cuda())], seg_len=K, bias=b.cuda())
    ref = F.linear(bf(a), bf(w), b)
    assert (out.cpu() - ref).abs().max() <= 2e-3 * max(1.0, ref.abs().max().item())


def test_bf16_A_gelu_bf16_out(rg, h):
    G = rg.gemm
    M, N, K = 2752, 1024, 512
    a, w, b = _rand((M, K), 4), _rand((N, K), 5, 0.05), _rand((N,), 6)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    G.gemm(h, M=M, N=N, K=K, W=G.pack_weight(w, "cuda"), out=out, A=a.cuda().bfloat16(), bias=b.cuda(), act=1)
    ref = F.gelu(F.linear(bf(a), bf(w), b))
    got = out.float().cpu()
    d = (got - ref).abs()
    m, n = divmod(int(d.argmax()), N)
    # a bf16 result: half an ulp of the largest magnitude this seed produces (7.3 -> ulp 2^-5) plus the fp32 accumulation slack
    assert d.max() <= 2 ** -6 + 1e-3, "max abs %.4e at (%d, %d): got %.6f, fp32 reference %.6f; %d elements beyond the bound" % (
        d.max(), m, n, got[m, n], ref[m, n], int((d > 2 ** -6 + 1e-3).sum()))


def test_ragged_vae_shapes(rg, h):
    """K = 78 (not a multiple of 8, odd row stride) and N = 61: the VAE's skel_embedding / final_layer."""
    G = rg.gemm
    M = 170
    a, w, b = _rand((M, 78), 7), _rand((512, 78), 8, 0.1), _rand((512,), 9)
    out = torch.empty(M, 512, device="cuda")
    G.gemm(h, M=M, N=512, K=78, W=G.pack_weight(w, "cuda"), out=out, segs=[G.Seg(a.cuda())], bias=b.cuda())
    assert (out.cpu() - F.linear(bf(a), bf(w), b)).abs().max() <= 2e-3
    a2, w2, b2 = _rand((M, 512), 10), _rand((61, 512), 11, 0.05), _rand((61,), 12)
    out2 = torch.full((M, 61), 7.0, device="cuda")
    G.gemm(h, M=M, N=61, K=512, W=G.pack_weight(w2, "cuda"), out=out2, segs=[G.Seg(a2.cuda())], seg_len=512, bias=b2.cuda())
    assert (out2.cpu() - F.linear(bf(a2), bf(w2), b2)).abs().max() <= 2e-3


def test_ln_prologue_softmax_stats_residual(rg, h):
    """LayerNorm-on-load from partial stats, per-head softmax on the first 512 columns,
    partial-stats output, residual, token-periodic bias, row duplication."""
    G = rg.gemm
    M, N, K, T = 172, 1536, 512, 43
    x = _rand((M, K), 13) * 1.7 + 0.3
    gam, bet = _rand((K,), 14, 0.1) + 1, _rand((K,), 15, 0.1)
    w, b = _rand((N, K), 16, 0.05), _rand((N,), 17)
    xs = x.view(M, 8, 64)
    stats = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], dim=-1).contiguous()  # [M,8,2]
    out = torch.empty(M, N, device="cuda")
    st_out = torch.zeros(M, N // 128, 2, device="cuda")
    G.gemm(h, M=M, N=N, K=K, W=G.pack_weight(w, "cuda"), out=out,
           segs=[G.Seg(x.cuda(), mode=G.A_LN, stats=stats.cuda(), gamma=gam.cuda(), beta=bet.cuda())], seg_len=512,
           bias=b.cuda(), softmax_cols=512, stats_out=st_out)
    xn = F.layer_norm(x, (K,), gam, bet)
    ref = F.linear(bf(xn), bf(w), b)
    ref[:, :512] = F.softmax(ref[:, :512].view(M, 16, 32), dim=-1).view(M, 512)
    assert (out.cpu() - ref).abs().max() <= 3e-3
    o = out.cpu().view(M, N // 128, 128)
    assert (st_out.cpu()[..., 0] - o.sum(-1)).abs().max() <= 1e-3
    assert (st_out.cpu()[..., 1] - (o * o).sum(-1)).abs().max() <= 1e-3
    # residual + tbias + row duplication (embed-style)
    Mh = M // 2
    res, tb = _rand((M, 512), 18), _rand((T, 512), 19)
    w2 = _rand((512, K), 20, 0.05)
    out2 = torch.empty(M, 512, device="cuda")
    G.gemm(h, M=M, N=512, K=K, W=G.pack_weight(w2, "cuda"), out=out2, segs=[G.Seg(x[:Mh].contiguous().cuda())],
           seg_len=512, a_row_mod=Mh, residual=res.cuda(), tbias=tb.cuda(), tb_period=T)
    ref2 = F.linear(bf(x[:Mh]), bf(w2)).repeat(2, 1) + res + tb.repeat(M // T, 1)
    assert (out2.cpu() - ref2).abs().max() <= 3e-3


def test_styl_prologue_segments(rg, h):
    """StylizationBlock front half on load, 3 STYL segments + 1 identity segment (the ca_mix GEMM
    shape), per-N-tile gamma/beta selection (the CA query GEMM)."""
    G = rg.gemm
    M, D = 215, 512
    y3 = _rand((M, 3 * D), 21)
    x1 = _rand((M, D), 22)
    gam, bet = _rand((3, D), 23, 0.1) + 1, _rand((3, D), 24, 0.1)
    ss = _rand((3, 2 * D), 25, 0.3)
    w, b = _rand((D, 4 * D), 26, 0.03), _rand((D,), 27)
    stats = []
    for c in range(3):
        yc = y3[:, c * D:(c + 1) * D].reshape(M, 4, 128)
        stats.append(torch.stack([yc.sum(-1), (yc * yc).sum(-1)], dim=-1).contiguous().cuda())
    y3d = y3.cuda()
    segs = [G.Seg(y3d, ld=3 * D, mode=G.A_STYL, stats=stats[c], gamma=gam[c].cuda(), beta=bet[c].cuda(),
                  scale_shift=ss[c].cuda(), col_offset=c * D) for c in range(3)]
    segs.append(G.Seg(x1.cuda()))
    out = torch.empty(M, D, device="cuda")
    G.gemm(h, M=M, N=D, K=4 * D, W=G.pack_weight(w, "cuda"), out=out, segs=segs, seg_len=D, bias=b.cuda())
    parts = []
    for c in range(3):
        hc = F.layer_norm(y3[:, c * D:(c + 1) * D], (D,), gam[c], bet[c])
        parts.append(F.silu(hc * (1 + ss[c, :D]) + ss[c, D:]))
    parts.append(x1)
    ref = F.linear(bf(torch.cat(parts, -1)), bf(w), b)
    assert (out.cpu() - ref).abs().max() <= 5e-3
    # per-N-group gamma/beta: N = 1536 in 3 groups of 512
    x = _rand((M, D), 28)
    xs = x.view(M, 8, 64)
    st = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], dim=-1).contiguous().cuda()
    w3 = _rand((3 * D, D), 29, 0.05)
    out3 = torch.empty(M, 3 * D, device="cuda")
    G.gemm(h, M=M, N=3 * D, K=D, W=G.pack_weight(w3, "cuda"), out=out3,
           segs=[G.Seg(x.cuda(), mode=G.A_LN, stats=st, gamma=gam.cuda(), beta=bet.cuda())], seg_len=D,
           gb_group=512, gb_stride=512)
    ref3 = torch.cat([F.linear(bf(F.layer_norm(x, (D,), gam[c], bet[c])), bf(w3[c * D:(c + 1) * D])) for c in range(3)], -1)
    assert (out3.cpu() - ref3).abs().max() <= 3e-3


@pytest.mark.parametrize("M,K", [(688, 512), (1376, 2048), (100, 1024)])
def test_narrow_tiles_bf16_A_epilogue_features(rg, h, M, K):
    """tile_n = 64 (64x64 tiles): bias, residual, partial statistics per 64 columns, bf16 copy, folded LayerNorm."""
    G = rg.gemm
    N = 512
    a, w, b = _rand((M, K), 21), _rand((N, K), 22, 0.05), _rand((N,), 23)
    res = _rand((M, N), 24)
    out = torch.empty(M, N, device="cuda")
    o2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = torch.zeros(M, N // 64, 2, device="cuda")
    G.gemm(h, M=M, N=N, K=K, W=G.pack_weight(w, "cuda"), out=out, A=a.cuda().bfloat16(), bias=b.cuda(), residual=res.cuda(),
           stats_out=st, out2=o2, tile_n=64)
    ref = F.linear(bf(a), bf(w), b) + res
    assert (out.cpu() - ref).abs().max() <= 3e-3 * max(1.0, ref.abs().max().item())
    assert (o2.float().cpu() - out.cpu()).abs().max() <= 2e-2 * max(1.0, ref.abs().max().item())
    o = out.cpu().view(M, N // 64, 64)
    assert (st.cpu()[..., 0] - o.sum(-1)).abs().max() <= 2e-3
    assert (st.cpu()[..., 1] - (o * o).sum(-1)).abs().max() <= 2e-2
    # same result as the default 64x128 tiles
    out128 = torch.empty(M, N, device="cuda")
    G.gemm(h, M=M, N=N, K=K, W=G.pack_weight(w, "cuda"), out=out128, A=a.cuda().bfloat16(), bias=b.cuda(), residual=res.cuda())
    assert torch.equal(out128, out)


@pytest.mark.parametrize("waves,path,M,N,K", [
    (0, 0, 1376, 512, 512),     # one round: ring of 4, 8 waves
    (0, 0, 4128, 512, 512),     # 260 workgroups: two 8-wave workgroups per CU (ring of 3, 128 VGPRs)
    (0, 0, 4128, 1536, 512),    # 780 workgroups: the same variant over three rounds
    (16, 0, 688, 512, 2048),    # that variant forced on a small grid
    (4, 0, 4128, 1024, 512),    # 4-wave workgroups, ring of 2
    (0, 4, 4128, 1536, 512),    # the 128x256 big-tile kernel
    (0, 0, 16640, 4096, 512),   # >= 8 rounds of 64x128 tiles: the big-tile kernel by policy
])
def test_bf16_A_kernel_variants_epilogue_features(rg, h, waves, path, M, N, K):
    """Every kernel the dispatch can pick for a bf16 A operand, with the whole fused epilogue: LayerNorm folded through
    row statistics (fetched by DMA in the LDS-DMA kernels), bias, residual, partial output statistics, bf16 copy."""
    G = rg.gemm
    x = _rand((M, K), 31) + 0.3                       # rows with a mean: the folded LayerNorm has something to remove
    gam, bet = _rand((K,), 32) * 0.2 + 1.0, _rand((K,), 33) * 0.1
    w, b, res = _rand((N, K), 34, 0.05), _rand((N,), 35), _rand((M, N), 36)
    # W' = W diag(gamma), c1 = rowsum(W'), bias' = b + W beta  (denoiser.py builds the same at load time)
    wp = G.pack_weight(w * gam[None, :], "cuda")
    c1 = wp.hi[:N, :K].float().sum(-1).contiguous()
    bias = (b + w @ bet).cuda()
    xs = x.cuda()
    parts = K // 128
    st_in = torch.stack([xs.view(M, parts, 128).sum(-1), (xs * xs).view(M, parts, 128).sum(-1)], dim=-1).contiguous()
    out = torch.empty(M, N, device="cuda")
    o2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = torch.zeros(M, N // 128, 2, device="cuda")
    h.lib.rg_set_gemm_waves(h._h, waves)
    h.lib.rg_set_gemm_path(h._h, path)
    try:
        G.gemm(h, M=M, N=N, K=K, W=wp, out=out, A=xs.bfloat16(), bias=bias, residual=res.cuda(), stats_out=st, out2=o2,
               ln_stats=st_in, ln_c1=c1)
        torch.cuda.synchronize()
    finally:
        h.lib.rg_set_gemm_waves(h._h, 0)
        h.lib.rg_set_gemm_path(h._h, 0)
    mu, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    rs = torch.rsqrt(var + 1e-5)
    ref = rs * (F.linear(bf(x), wp.hi[:N, :K].float().cpu()) - mu * c1.cpu()[None, :]) + bias.cpu() + res
    scale = max(1.0, ref.abs().max().item())
    assert (out.cpu() - ref).abs().max() <= 3e-3 * scale
    assert (o2.float().cpu() - out.cpu()).abs().max() <= 2e-2 * scale
    o = out.cpu().view(M, N // 128, 128)
    assert (st.cpu()[..., 0] - o.sum(-1)).abs().max() <= 4e-3 * scale
    assert (st.cpu()[..., 1] - (o * o).sum(-1)).abs().max() <= 4e-2 * scale * scale


@pytest.mark.parametrize("M,N,K,bf16_a", [(5760, 1536, 512, True), (192, 512, 512, True), (2400, 1024, 512, False), (576, 512, 1024, True)])
def test_grouped_launch_equals_single_launches(rg, h, M, N, K, bf16_a):
    """rg_gemm_grouped / rg_layernorm_grouped / rg_add_rows_grouped / rg_copy_rows_grouped / rg_mha_bf16_grouped: four
    problems of one shape in one launch (the four body-part VAEs) must give the bits of four single launches; descriptors
    of different shapes fall back to single launches inside the call."""
    import ctypes
    G = rg.gemm
    dev = "cuda"
    ws = [G.pack_weight(_rand((N, K), 20 + i, 0.05), dev) for i in range(4)]
    a = [_rand((M, K), 30 + i).to(dev) for i in range(4)]
    b = [_rand((N,), 40 + i).to(dev) for i in range(4)]
    res = [_rand((M, N), 50 + i).to(dev) for i in range(4)]

    def desc(i, out):
        kw = dict(M=M, N=N, K=K, W=ws[i], out=out, bias=b[i], residual=res[i], act=1 if i % 2 else 0)
        if bf16_a:
            kw["A"] = a[i].bfloat16()
        else:
            kw.update(segs=[G.Seg(a[i])], seg_len=K)
        return G.make_desc(**kw), kw

    single = [torch.empty(M, N, device=dev) for _ in range(4)]
    keep = []
    for i in range(4):
        d, kw = desc(i, single[i])
        keep.append(kw)
        G.launch(h, d)
    grouped = [torch.zeros(M, N, device=dev) for _ in range(4)]
    ds = [desc(i, grouped[i]) for i in range(4)]
    arr = (G.GemmDesc * 4)(*[d for d, _ in ds])
    s = torch.cuda.current_stream().cuda_stream
    assert h.lib.rg_gemm_grouped(h._h, arr, 4, ctypes.c_void_p(s)) == 0
    torch.cuda.synchronize()
    for i in range(4):
        assert torch.equal(single[i], grouped[i]), i
    # a group whose members differ in shape is launched one by one (same results)
    d_small, kw_small = G.make_desc(M=64, N=N, K=K, W=ws[0], out=(o_small := torch.zeros(64, N, device=dev)),
                                    **({"A": a[0][:64].bfloat16().contiguous()} if bf16_a else {"segs": [G.Seg(a[0][:64].contiguous())], "seg_len": K})), None
    grouped2 = torch.zeros(M, N, device=dev)
    d1, kw1 = desc(1, grouped2)
    arr2 = (G.GemmDesc * 2)(d_small, d1)
    assert h.lib.rg_gemm_grouped(h._h, arr2, 2, ctypes.c_void_p(s)) == 0
    torch.cuda.synchronize()
    assert torch.equal(grouped2, single[1]) and o_small.abs().sum().item() > 0

    # row-wise grouped ops
    vp = lambda ts: (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
    D = 512
    x = [_rand((M, D), 60 + i).to(dev) for i in range(4)]
    g_, b_ = [_rand((D,), 70 + i).to(dev) for i in range(4)], [_rand((D,), 80 + i).to(dev) for i in range(4)]
    o1, o16a = [torch.empty(M, D, device=dev) for _ in range(4)], [torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(4)]
    o2, o16b = [torch.empty(M, D, device=dev) for _ in range(4)], [torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(4)]
    for i in range(4):
        h.call("layernorm", x[i], g_[i], b_[i], o1[i], M, D, o16a[i])
    h.call("layernorm_grouped", 4, vp(x), vp(g_), vp(b_), vp(o2), M, D, vp(o16b))
    pos = [_rand((64, D), 90 + i).to(dev) for i in range(4)]
    s1, s2 = [torch.empty(M, D, device=dev) for _ in range(4)], [torch.empty(M, D, device=dev) for _ in range(4)]
    for i in range(4):
        h.call("add_rows", x[i], pos[i], s1[i], M * D, 64 * D)
    h.call("add_rows_grouped", 4, vp(x), vp(pos), vp(s2), M * D, 64 * D)
    torch.cuda.synchronize()
    for i in range(4):
        assert torch.equal(o1[i], o2[i]) and torch.equal(o16a[i], o16b[i]) and torch.equal(s1[i], s2[i]), i
