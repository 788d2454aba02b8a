"""Device time of the four-part VAE encode alone on the chip: fused one-launch encoder stacks (rg_venc_forward) against the
per-op launch chains (grouped launches), 16 clips and 48 exemplars (BASELINE config 3), HIP-graph replays."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
P = {}
for i, part in enumerate(rg.synth.PARTS):
    P.update(rg.synth.synth_vae_state(101 + i, vae_cfgs[part], prefix="gesture_rep_encoder.%s_vae." % part))
f = lambda t: t.float().contiguous()
for B in (16, 48):
    data = rg.synth.synth_batch(B, seed=1234, device="cuda")
    tape = rg.synth.NoiseTape(5)
    eps = [tape.draw((B * 10, 1, 512)).cuda() for _ in range(4)]
    for name, kw in (("fused", dict(part_streams=False, grouped=True)), ("chains (grouped)", dict(part_streams=False, grouped=True, fused_encoder=False))):
        gre = rg.vae.GestureRepEncoder(P, vae_cfgs, "cuda", "bf16", **kw)
        run = lambda: gre.encode_device(f(data["motion_upper"]), f(data["motion_lower"]), f(data["motion_face"]), f(data["motion_hands"]),
                                        f(data["trans"]), f(data["facial"]), f(data["contact"]), eps)
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run()
            with rg.capi.capture(g):
                run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): g.replay()
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        print("encode B=%d %-18s %.3f ms per call (graph replay, alone on the chip)" % (B, name, e0.elapsed_time(e1) / 20), flush=True)

import numpy as np
z = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal((16, 43, 512)).astype("float32")).cuda()
for name, kw in (("fused", dict(part_streams=False, grouped=True)), ("chains (grouped)", dict(part_streams=False, grouped=True, fused_decoder=False))):
    gre = rg.vae.GestureRepEncoder(P, vae_cfgs, "cuda", "bf16", **kw)
    data = rg.synth.synth_batch(1, seed=1, device="cuda")
    gre.encode_device(f(data["motion_upper"]), f(data["motion_lower"]), f(data["motion_face"]), f(data["motion_hands"]),
                      f(data["trans"]), f(data["facial"]), f(data["contact"]), [torch.zeros(10, 1, 512, device="cuda")] * 4)      # (joint counts)
    run = lambda: gre.decode(z)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
        with rg.capi.capture(g):
            run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): g.replay()
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("decode B=16 %-18s %.3f ms per call (graph replay, alone on the chip)" % (name, e0.elapsed_time(e1) / 20), flush=True)
