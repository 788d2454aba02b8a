import importlib, sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/oracle") else os.getcwd())
import numpy as np, torch
rg = importlib.import_module("rag-gesture_amd")
from oracle import vae as ovae, rotation as orot
torch.set_num_threads(32)
NAMES = ("upper", "lower", "face", "hands", "transl", "exps", "contact"); ROT = NAMES[:4]
def relerr(a,b): return ((a-b).norm()/b.norm()).item()
def rot_relerr(a,b):
    ma, mb = orot.axis_angle_to_matrix(a.reshape(-1,3)), orot.axis_angle_to_matrix(b.reshape(-1,3)); return ((ma-mb).norm()/mb.norm()).item()
for L in (2, 8):
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder", normalize_before=True, num_layers=L)
    P = {}
    for i, part in enumerate(rg.synth.PARTS):
        P.update(rg.synth.synth_vae_state(101 + i, vae_cfgs[part], prefix="gesture_rep_encoder.%s_vae." % part))
    B = 4
    g = np.random.Generator(np.random.PCG64(98))
    z = torch.from_numpy(g.standard_normal((B, 43, 512)).astype(np.float32)); z[:, [10, 21, 32]] = 0
    with torch.no_grad():
        ref = ovae.gesture_decode(P, vae_cfgs, z)
    for prec in ("fp32", "bf16"):
        gre = rg.vae.GestureRepEncoder(P, vae_cfgs, "cuda", prec, part_streams=False, grouped=True)
        d = rg.synth.synth_batch(B, seed=5)
        tape = rg.synth.NoiseTape(1)
        gre.encode(d["motion_upper"], d["motion_lower"], d["motion_face"], d["motion_hands"], d["trans"], d["facial"], d["contact"], d["motion_mask"],
                   [tape.draw((B * 10, 1, 512)) for _ in range(4)])     # (the joint counts are taken from the encode inputs)
        dec = gre.decode(z.cuda())
        print(L, prec, ["%s %.2e" % (nm, rot_relerr(a.cpu(), r) if nm in ROT else relerr(a.cpu(), r)) for nm, a, r in zip(NAMES, dec, ref)], flush=True)
