"""Diagnostics for the sequence-stationary forward (rg_seq_forward): head output against the launch chain and the oracle
(bf16 operands), and the staged register dumps of layer 0 against a torch restatement.
    python profiles/dbg/seq_check.py [L] [B]"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rg = importlib.import_module("rag-gesture_amd")
from oracle import denoiser as od  # noqa: E402


def relerr(a, b):
    return ((a - b).norm() / b.norm()).item()


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cfg = rg.synth.default_model_cfg(num_layers=L)
    P = rg.synth.synth_denoiser_state(0, cfg)
    sch = rg.schedule.Schedule()
    W = rg.denoiser.DenoiserWeights(P, cfg, sch, "cuda")
    data = rg.synth.synth_batch(B, seed=1234)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(99)).standard_normal((B, 43, 512)).astype(np.float32))
    mm = torch.ones(B, 43)
    mm[:, [10, 21, 32]] = 0
    qm = od.make_query_masks(mm)
    step, t = 34, 514
    chain = rg.denoiser.DenoiserSession(W, B, engine="chain")
    seq = rg.denoiser.DenoiserSession(W, B, engine="seq")
    for masks in ("ones", "real"):
        q = qm if masks == "real" else None
        chain.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, q)
        seq.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, q)
        hc = chain.forward(x.cuda(), step).clone().view(2 * B, 43, 512).cpu()
        hs = seq.forward(x.cuda(), step).clone().view(2 * B, 43, 512).cpu()
        torch.cuda.synchronize()
        print("masks=%s  seq vs chain: rel %.3e  cond rows %.3e  uncond rows %.3e  finite %s"
              % (masks, relerr(hs, hc), relerr(hs[:B], hc[:B]), relerr(hs[B:], hc[B:]), bool(torch.isfinite(hs).all())))
        # oracle with bf16 operands on the head output (before the CFG mix): run the layers by hand
        od.OPTS.update(bf16=True, masked_ln="exact")
        try:
            xf = od.encode_conditions(P, data["word"], data["audio"], data["speaker_ids"])
            ts = torch.full((B,), t, dtype=torch.long)
            emb = od.linear(P, "time_embed.2", F.silu(od.linear(P, "time_embed.0", od.timestep_embedding(ts, 512))))
            h = od.embed_input(P, x, 10).repeat(2, 1, 1)
            stages = {1: h.clone()}
            cond_type = torch.cat([torch.ones(B, 1, 1), torch.zeros(B, 1, 1)], dim=0)
            xf2 = {k: v.repeat(2, 1, 1) for k, v in xf.items()}
            emb2 = emb.repeat(2, 1)
            sm = mm.clone().unsqueeze(-1).repeat(2, 1, 1)
            qm2 = {k: v.repeat(2, 1) for k, v in q.items()} if q is not None else None
            for l in range(L):
                name = "temporal_decoder_blocks.%d" % l
                h1 = od.efficient_self_attention(P, name + ".sa_block", h, sm, emb2, 16)
                outs = []
                for c in xf2.keys():
                    outs.append(od.efficient_cross_attention(P, name + ".ca_blocks." + c, h1, xf2[c], emb2,
                                                             qm2[c] if qm2 is not None else None, cond_type, 16))
                h2 = od.linear(P, name + ".ca_mix", torch.cat(outs, dim=-1))
                y = od.linear(P, name + ".ffn.linear2", F.gelu(od.linear(P, name + ".ffn.linear1", h2)))
                h3 = h2 + od.stylization_block(P, name + ".ffn.proj_out", y, emb2)
                if l == 0:
                    stages.update({2: h1.clone(), 3: h2.clone(), 4: h3.clone()})
                h = h3
            ref = od.linear(P, "out", h)
        finally:
            od.OPTS.update(bf16=False, masked_ln="torch")
        print("          seq vs bf16 oracle: %.3e   chain vs bf16 oracle: %.3e" % (relerr(hs, ref), relerr(hc, ref)))
        dump = torch.zeros(2 * B, 48, 512, device="cuda")
        for stg in (1, 2, 3, 4):
            dump.zero_()
            seq.sq.run(x.cuda(), step, dump=dump, dump_stage=stg, dump_layer=0)
            torch.cuda.synchronize()
            d = dump[:, :43].cpu()
            r = stages[stg]
            print("          stage %d (layer 0): rel %.3e  cond %.3e  uncond %.3e" % (stg, relerr(d, r), relerr(d[:B], r[:B]), relerr(d[B:], r[B:])))


if __name__ == "__main__":
    main()
