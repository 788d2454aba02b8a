"""In-kernel phase stamps of the block-fused VAE decoder (RG_DIAG=1 build: csrc/rg_vdec.hip VSTAMP): where the ~85 us of a launch
that are not its unit GEMMs go.  One part, B sequences (4 B workgroups), every launch of a decode; mean over workgroups and waves.
    RG_DIAG=1 python profiles/dbg/vdec_stamps.py [B]"""
import importlib, os, sys, ctypes
import torch
assert os.environ.get("RG_DIAG") == "1", "needs the diagnostic build: RG_DIAG=1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
vcfg = rg.synth.default_vae_cfg("upper")
vae = rg.vae.TransformerVAE(rg.synth.synth_vae_state(101, vcfg, prefix=""), vcfg, "cuda", "bf16")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B * 160, 512, device="cuda"); pos = torch.randn(B * 160, 512, device="cuda")
h, st = vae.h, vae.vdec.st
nb = st.nb
qimg = torch.empty(4 * B * 48 * 1024, device="cuda", dtype=torch.uint8)
kbuf = torch.empty(2 * B * 160 * 512, device="cuda", dtype=torch.bfloat16); vt = torch.empty_like(kbuf)
xbuf = torch.empty(4 * B * nb * 8 * 12 * 64 * 4, device="cuda")
dump = torch.zeros(4 * B * 8 * 16, device="cuda")
NAMES = ["descriptors + Q image", "attention", "x rows in", "out_proj + norm1", "FFN + norm2 (+ skip push)", "skip linear + x rows out",
         "Q / K / V projections", "K / V / Q image stores"]


def launch(step, stamp):
    a = rg.vencfwd.VdecArgs()
    a.wstream, a.pstream = st.wstream.data_ptr(), st.pstream.data_ptr()
    a.x, a.pos, a.qimg, a.kbuf, a.vt, a.xbuf = (t.data_ptr() for t in (x, pos, qimg, kbuf, vt, xbuf))
    a.dump, a.pad_ = (dump.data_ptr(), 99) if stamp else (None, 0)
    a.nseq, a.nb, a.step = B, nb, step
    h.call("vdec_step", ctypes.byref(a))


for _ in range(2):
    for s in range(2 * nb + 2):
        launch(s, False)
torch.cuda.synchronize()
for s in range(2 * nb + 2):
    dump.zero_()
    launch(s, True)
    torch.cuda.synchronize()
    t = dump.view(4 * B, 8, 16)[:, :, :9].double() * 0.01            # us since stamp 0, per (workgroup, wave)
    parts, prev = [], torch.zeros_like(t[:, :, 0])
    for i in range(1, 9):
        cur = t[:, :, i]
        hit = cur > 0
        if hit.any():
            parts.append("%s %.1f" % (NAMES[i - 1], (cur - prev)[hit].mean().item()))
            prev = torch.where(hit, cur, prev)
    print("launch %2d (%d workgroups): whole %.1f us in the kernel | %s" % (s, 4 * B, t[:, :, 8].mean().item(), " | ".join(parts)), flush=True)
