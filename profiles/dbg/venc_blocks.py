"""Block by block: the fused VAE encoder (rg_venc_forward, dump_block) against the oracle's skip encoder on the same input."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
from oracle import vae as ovae
part = "upper"
vcfg = rg.synth.default_vae_cfg(part)
sd = rg.synth.synth_vae_state(101, vcfg, prefix="")
vae = rg.vae.TransformerVAE(sd, vcfg, "cuda", "bf16")
assert vae.venc is not None
nseq, S = int(sys.argv[1]) if len(sys.argv) > 1 else 6, 17
g = np.random.Generator(np.random.PCG64(5))
x = torch.from_numpy(g.standard_normal((nseq, S, 512)).astype(np.float32))
nb = vae.venc.st.nb
# oracle states behind every block (sequence-first layout [S, B, D])
xs_o = x.permute(1, 0, 2)
act = ovae._act(vcfg["transformer_activation"])
states, stack = [], []
with torch.no_grad():
    t = xs_o
    for i in range(nb):
        t = ovae.encoder_layer(sd, "encoder.input_blocks.%d" % i, t, vcfg["num_heads"], act, False)
        stack.append(t); states.append(t)
    t = ovae.encoder_layer(sd, "encoder.middle_block", t, vcfg["num_heads"], act, False); states.append(t)
    for i in range(nb):
        t = ovae._lin(sd, "encoder.linear_blocks.%d" % i, torch.cat([t, stack.pop()], dim=-1))
        t = ovae.encoder_layer(sd, "encoder.output_blocks.%d" % i, t, vcfg["num_heads"], act, False); states.append(t)
    final = ovae._ln(sd, "encoder.norm", t)
xd = x.reshape(nseq * S, 512).cuda().contiguous()
out = vae.venc.run(xd, nseq, S)
torch.cuda.synchronize()
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
print("final: rel err %.3e  (nan %d)" % (rel(out.cpu().view(nseq, S, 512), final.permute(1, 0, 2)), int(torch.isnan(out).sum())))
for blk in range(2 * nb + 1):
    dump = torch.zeros((nseq + 1) // 2, 48, 512, device="cuda")
    vae.venc.run(xd, nseq, S, dump=dump, dump_block=blk)
    torch.cuda.synchronize()
    d = dump.cpu()
    got = torch.stack([d[s // 2, (s % 2) * 24:(s % 2) * 24 + S] for s in range(nseq)])     # [nseq, S, 512]
    ref = states[blk].permute(1, 0, 2)
    print("block %d: rel err %.3e  worst sequence %.3e  pad rows finite %s" % (blk, rel(got, ref), max(rel(got[s], ref[s]) for s in range(nseq)),
          bool(torch.isfinite(d).all())))
