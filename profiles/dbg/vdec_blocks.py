"""Block by block: the block-fused VAE decoder (rg_vdec_step, dump) against the oracle's skip encoder with pos on the same input."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
from oracle import vae as ovae
vcfg = rg.synth.default_vae_cfg("upper")
sd = rg.synth.synth_vae_state(101, vcfg, prefix="")
vae = rg.vae.TransformerVAE(sd, vcfg, "cuda", "bf16")
assert vae.vdec is not None
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = np.random.Generator(np.random.PCG64(5))
z = torch.from_numpy(g.standard_normal((B, 10, 512)).astype(np.float32))
nb, heads = vae.vdec.st.nb, vcfg["num_heads"] * 8
act = ovae._act(vcfg["transformer_activation"])
with torch.no_grad():
    xseq = torch.cat((z.permute(1, 0, 2), torch.zeros(150, B, 512)), dim=0)          # [160, B, D]
    pos = xseq + sd["query_pos_decoder.pe"][:160]
    states, stack, t = [xseq], [], xseq
    for i in range(nb):
        t = ovae.encoder_layer(sd, "decoder.input_blocks.%d" % i, t, heads, act, False, None, pos); stack.append(t); states.append(t)
    t = ovae.encoder_layer(sd, "decoder.middle_block", t, heads, act, False, None, pos); states.append(t)
    for i in range(nb):
        t = ovae._lin(sd, "decoder.linear_blocks.%d" % i, torch.cat([t, stack.pop()], dim=-1))
        t = ovae.encoder_layer(sd, "decoder.output_blocks.%d" % i, t, heads, act, False, None, pos); states.append(t)
    final = ovae._ln(sd, "decoder.norm", t)
xd = xseq.permute(1, 0, 2).reshape(B * 160, 512).cuda().contiguous()
pd = pos.permute(1, 0, 2).reshape(B * 160, 512).cuda().contiguous()
dumps = []
out = vae.vdec.run(xd, pd, B, dumps=dumps)
torch.cuda.synchronize()
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
print("final: rel err %.3e  (nan %d)" % (rel(out.cpu().view(B, 160, 512), final.permute(1, 0, 2)), int(torch.isnan(out).sum())))
for k, d in enumerate(dumps):
    got = d.cpu().view(B, 4, 48, 512)[:, :, :40].reshape(B, 160, 512)
    ref = states[k].permute(1, 0, 2)
    print("launch %d (state behind block %d): rel err %.3e  finite %s" % (k, k - 1, rel(got, ref), bool(torch.isfinite(d).all())))
