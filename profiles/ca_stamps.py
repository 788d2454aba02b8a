"""Diagnostic (RG_DIAG=1 build): in-kernel phase stamps of ca_stylize (workgroup 0, thread 0; 100 MHz clock) inside real
denoiser forward steps (the last ca_stylize launch of a step leaves its stamps)."""
import ctypes, importlib, os, sys
import numpy as np
import torch
os.environ["RG_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rg = importlib.import_module("rag-gesture_amd")
h = rg.capi.get_handle(0)
cfg = rg.synth.default_model_cfg(num_layers=8)
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, rg.schedule.Schedule(), "cuda")
NAMES = ["entry -> loads issued (q DMA, A frags, params, mask)", "loads landed", "q reads, split, 24 mfma", "y -> lds (mask select)",
         "row statistics of the head tile", "workgroup barrier", "combine statistics + barrier", "LN + stylization + SiLU -> bf16 store"]
for B in (8, 16, 48):
    sess = rg.denoiser.DenoiserSession(W, B, engine="chain")
    d = rg.synth.synth_batch(B, seed=1)
    mask = torch.ones(B, 43)
    sess.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, {c: torch.ones(B, 43) for c in rg.denoiser.CONDS})
    x = torch.randn(B, 43, 512, device="cuda")
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    h.lib.rg_debug_set_stamp_buffer3(ctypes.c_void_p(buf.data_ptr()))
    rows = []
    for it in range(30):
        sess.forward(x, 30)
        torch.cuda.synchronize()
        rows.append(buf.cpu().numpy()[:9].copy())
    h.lib.rg_debug_set_stamp_buffer3(ctypes.c_void_p(0))
    dd = np.diff(np.array(rows), axis=1) * 10.0
    med = np.median(dd, axis=0)
    print("B=%d: in-kernel span of WG 0: %.2f us" % (B, med.sum() / 1e3))
    for n, v in zip(NAMES, med):
        print("    %-54s %6.2f us" % (n, v / 1e3))
