"""Per-launch time of the block-fused decoder (one part, 64 sequences = 256 workgroups = one per CU, alone on the chip): the
launches differ in their unit count (3, 8, 10, 5), which separates the cost per unit GEMM from the fixed part of a launch."""
import importlib, os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
vcfg = rg.synth.default_vae_cfg("upper")
sd = rg.synth.synth_vae_state(101, vcfg, prefix="")
vae = rg.vae.TransformerVAE(sd, vcfg, "cuda", "bf16")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B * 160, 512, device="cuda"); pos = torch.randn(B * 160, 512, device="cuda")
h, st = vae.h, vae.vdec.st
nb = st.nb
V = rg.vencfwd.VdecArgs
qimg = torch.empty(4 * B * 48 * 1024, device="cuda", dtype=torch.uint8)
kbuf = torch.empty(2 * B * 160 * 512, device="cuda", dtype=torch.bfloat16); vt = torch.empty_like(kbuf)
xbuf = torch.empty(4 * B * nb * 8 * 12 * 64 * 4, device="cuda")
def launch(step):
    a = V(); a.wstream, a.pstream = st.wstream.data_ptr(), st.pstream.data_ptr()
    a.x, a.pos, a.qimg, a.kbuf, a.vt, a.xbuf = (t.data_ptr() for t in (x, pos, qimg, kbuf, vt, xbuf))
    a.dump = None; a.nseq, a.nb, a.step = B, nb, step
    h.call("vdec_step", ctypes.byref(a))
for _ in range(2):
    for s in range(2 * nb + 2): launch(s)
torch.cuda.synchronize()
for s in range(2 * nb + 2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): launch(s)
    e1.record(); torch.cuda.synchronize()
    units = (0 if s == 0 else 5) + (0 if s == 2 * nb + 1 else (2 if s > nb else 0) + 3)
    print("step %2d: %2d units  %.1f us per launch (B=%d sequences, %d workgroups)" % (s, units, e0.elapsed_time(e1) / 10 * 1e3, B, 4 * B), flush=True)
