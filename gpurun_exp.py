"""Temporary: time ca_stylize vs ca_attention + stylize."""
import importlib, torch
rg = importlib.import_module("rag-gesture_amd")
G = rg.gemm
h = rg.capi.get_handle(0)
T, D = 43, 512
def timeit(fn, n=100):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); g.replay(); e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (16, 48):
    R = 2 * B; M = R * T
    q3 = torch.rand(B * T, 3 * D, device="cuda"); apre = torch.randn(3, B, 16, 32, 32, device="cuda")
    qm = torch.ones(3, R, T, device="cuda"); qm[:, :, [10, 20, 30]] = 0
    gam = torch.ones(3, D, device="cuda"); bet = torch.zeros(3, D, device="cuda"); ss = torch.randn(3, 2 * D, device="cuda") * 0.1
    tab = torch.randn(2, 3 * D, device="cuda").bfloat16(); hcat = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
    apt = torch.empty(3, B, 16, 2, 32, 32, device="cuda", dtype=torch.bfloat16)
    h.call("split_transpose_bf16", apre, apt, 3 * B * 16)
    t = timeit(lambda: h.call("ca_stylize", q3, apt, qm, gam, bet, ss, tab, hcat, 4 * D, B, B, T, D, 3))
    print(f"B={B}: ca_stylize {t:.1f} us", flush=True)
