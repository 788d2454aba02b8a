"""Diagnostic (RG_DIAG=1 build): in-kernel phase stamps of sa_attention (workgroup 0, wave 0; 100 MHz clock), median over
launches in a graph-replayed chain, next to the event-timed duration per launch."""
import ctypes, importlib, os, sys
import numpy as np
import torch
os.environ["RG_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rg = importlib.import_module("rag-gesture_amd")
h = rg.capi.get_handle(0)
T, D = 43, 512
NAMES = ["entry->dma issued", "dma issued->landed (vmcnt 0)", "syncthreads", "k softmax", "A = P^T V (lds reads, split, 24 mfma)",
         "y = q A (q reads, 24 mfma)", "y -> lds", "y store + stats", "syncthreads", ]
for R in (16, 32, 96):
    M = R * T
    qkv = [torch.randn(M, 3 * D, device="cuda") for _ in range(4)]
    mask = torch.ones(R, T, device="cuda")
    y = [torch.empty(M, D, device="cuda") for _ in range(4)]
    st = [torch.empty(M, 8, 2, device="cuda") for _ in range(4)]
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    h.lib.rg_debug_set_stamp_buffer3(ctypes.c_void_p(buf.data_ptr()))
    rows = []
    for it in range(40):
        for i in range(4):
            h.call("sa_attention", qkv[i], 3 * D, mask, y[i], D, st[i], R, T, D, None, 0, 1)
        torch.cuda.synchronize()
        rows.append(buf.cpu().numpy()[:10].copy())
    d = np.diff(np.array(rows), axis=1) * 10.0   # ns
    med = np.median(d, axis=0)
    print("R=%d: in-kernel span of wave 0 / WG 0: %.2f us" % (R, med.sum() / 1e3))
    for n, v in zip(NAMES, med):
        print("    %-44s %6.2f us" % (n, v / 1e3))
h.lib.rg_debug_set_stamp_buffer3(ctypes.c_void_p(0))
