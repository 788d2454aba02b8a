"""Run-to-run determinism of the VAE launch paths: the same decode / encode repeated, compared bitwise with the first result.
(bench.py's verification and the jitter test both saw 48 elements of a decoded output differ between two runs of the SAME
synchronous call; the latent never differed.)"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
P = {}
for i, part in enumerate(rg.synth.PARTS):
    P.update(rg.synth.synth_vae_state(101 + i, vae_cfgs[part], prefix="gesture_rep_encoder.%s_vae." % part))
NAMES = ("upper", "lower", "face", "hands", "transl", "exps", "contact")
g = np.random.Generator(np.random.PCG64(99))
z = torch.from_numpy(g.standard_normal((B, 43, 512)).astype(np.float32)).cuda()
data = rg.synth.synth_batch(B, seed=1234, device="cuda")
tape = rg.synth.NoiseTape(5)
eps = [tape.draw((B * 10, 1, 512)).cuda() for _ in range(4)]
f = lambda t: t.float().contiguous()
for path, kw in (("part_streams", dict(part_streams=True)), ("grouped", dict(part_streams=False, grouped=True)),
                 ("single_chain", dict(part_streams=False, grouped=False))):
    gre = rg.vae.GestureRepEncoder(P, vae_cfgs, "cuda", "bf16", **kw)
    first_d = first_e = None
    bad_d = bad_e = 0
    t0 = time.time()
    for it in range(iters):
        lat, _ = gre.encode_device(f(data["motion_upper"]), f(data["motion_lower"]), f(data["motion_face"]), f(data["motion_hands"]),
                                   f(data["trans"]), f(data["facial"]), f(data["contact"]), eps)
        dec = gre.decode(z)
        torch.cuda.synchronize()
        if first_d is None:
            first_d, first_e = [d.clone() for d in dec], lat.clone()
            continue
        if not torch.equal(lat, first_e):
            bad_e += 1
            d = (lat != first_e).nonzero()
            print("%s iter %d ENCODE: %d elements differ, max %.3e, first %s last %s" % (path, it, d.shape[0], (lat - first_e).abs().max(), d[0].tolist(), d[-1].tolist()), flush=True)
        for nm, a, b in zip(NAMES, dec, first_d):
            if not torch.equal(a, b):
                bad_d += 1
                d = (a != b).nonzero()
                rows = sorted(set((int(x[0]), int(x[1])) for x in d.tolist()))
                cols = sorted(set(int(x[2]) for x in d.tolist()))
                print("%s iter %d DECODE %s: %d elements differ, max %.3e, rows %s cols %s" % (path, it, nm, d.shape[0], (a - b).abs().max(), rows[:8], cols[:24]), flush=True)
    print("%s: %d iterations, %.1f s, decode mismatches %d, encode mismatches %d" % (path, iters, time.time() - t0, bad_d, bad_e), flush=True)
