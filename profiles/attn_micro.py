"""Micro-benchmark of the non-GEMM denoiser kernels (self-attention, stylization pre-pass) as graph-replayed chains on
rotating buffers at the row counts of the guided workload (R = clips x 2 CFG rows)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rg = importlib.import_module("rag-gesture_amd")
h = rg.capi.get_handle(0)
T, D, CHAIN, ROT = 43, 512, 160, 4


def chain(fn):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(ROT):
            fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for i in range(CHAIN):
                fn(i % ROT)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / CHAIN)
    return best


for R in (16, 32, 48, 96):
    M = R * T
    qkv = [torch.randn(M, 3 * D, device="cuda") for _ in range(ROT)]
    mask = torch.ones(R, T, device="cuda")
    y = [torch.empty(M, D, device="cuda") for _ in range(ROT)]
    st = [torch.empty(M, 8, 2, device="cuda") for _ in range(ROT)]
    t_sa = chain(lambda i: h.call("sa_attention", qkv[i], 3 * D, mask, y[i], D, st[i], R, T, D, None, 0, 1))
    t_sa0 = chain(lambda i: h.call("sa_attention", qkv[i], 3 * D, mask, y[i], D, st[i], R, T, D, None, 0, 0))
    print("R=%3d (M=%4d): sa_attention mfma %.2f us, valu %.2f us" % (R, M, t_sa, t_sa0), flush=True)
