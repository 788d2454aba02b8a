"""Kernels of ONE fused four-part VAE encode (B clips) and one decode, by name and duration, from a rocprofv3 kernel trace of
graph replays: python profiles/dbg/enc_kernels.py B  (run under rocprofv3 --kernel-trace; see profiles/dbg/r06_enc_kernels.sh)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
P = {}
for i, part in enumerate(rg.synth.PARTS):
    P.update(rg.synth.synth_vae_state(101 + i, vae_cfgs[part], prefix="gesture_rep_encoder.%s_vae." % part))
f = lambda t: t.float().contiguous()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
data = rg.synth.synth_batch(B, seed=1234, device="cuda")
tape = rg.synth.NoiseTape(5)
eps = [tape.draw((B * 10, 1, 512)).cuda() for _ in range(4)]
gre = rg.vae.GestureRepEncoder(P, vae_cfgs, "cuda", "bf16", part_streams=False, grouped=True)
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
torch.cuda._sleep(50_000_000)      # a marker gap in the trace: what follows is exactly 5 replays
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
