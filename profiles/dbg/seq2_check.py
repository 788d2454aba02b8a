"""Two sequences per workgroup (rg_seq2_forward) against one per workgroup (rg_seq_forward): bit for bit, every sequence, every
launch form; then launch times.   python profiles/dbg/seq2_check.py [L]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rg = importlib.import_module("rag-gesture_amd")
from oracle import denoiser as od  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cfg = rg.synth.default_model_cfg(num_layers=L)
    P = rg.synth.synth_denoiser_state(0, cfg)
    W = rg.denoiser.DenoiserWeights(P, cfg, rg.schedule.Schedule(), "cuda")
    bad = 0
    for B in (1, 2, 3, 5, 16):
        data = rg.synth.synth_batch(B, seed=80 + B)
        x = torch.from_numpy(np.random.Generator(np.random.PCG64(8)).standard_normal((B, 43, 512)).astype(np.float32)).cuda()
        mm = torch.ones(B, 43)
        mm[:, [10, 21, 32]] = 0
        mm[0, 30:] = 0
        outs = {}
        for key, kw in (("one", dict(seq_duo=False)), ("duo", dict(seq_duo=True)), ("duo_pairs", dict(seq_duo=True, seq_pairs=True))):
            sess = rg.denoiser.DenoiserSession(W, B, engine="seq", **kw)
            sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, od.make_query_masks(mm))
            outs[key] = [sess.forward(x, st, sb, sp).clone() for st, sb, sp in ((49, None, None), (23, 40, max(1, B // 3)), (0, None, None), (7, 30, 0))]
            torch.cuda.synchronize()
        for key in ("duo", "duo_pairs"):
            for i, (a, b) in enumerate(zip(outs[key], outs["one"])):
                a, b = a.view(2 * B, 43, 512), b.view(2 * B, 43, 512)
                eq = [bool(torch.equal(a[r], b[r])) for r in range(2 * B)]
                fin = bool(torch.isfinite(a).all())
                if not all(eq) or not fin:
                    bad += 1
                    print("B=%d %s case %d: finite %s, sequences differing %s, max abs diff %.3e (ref max %.3e)"
                          % (B, key, i, fin, [r for r in range(2 * B) if not eq[r]], (a - b).abs().max().item(), b.abs().max().item()))
        print("B=%d compared" % B, flush=True)
    print("MISMATCHES: %d" % bad)
    # launch times, 128 sequences
    B = 64
    data = rg.synth.synth_batch(B, seed=3)
    x = torch.randn(B, 43, 512, device="cuda")
    mm = torch.ones(B, 43)
    mm[:, [10, 21, 32]] = 0
    for key, kw in (("one workgroup per sequence", dict(seq_duo=False)), ("one per clip (old pairs)", dict(seq_duo=False, seq_pairs=True)),
                    ("duo", dict(seq_duo=True)), ("duo, twin pairs behind", dict(seq_duo=True, seq_pairs=True))):
        sess = rg.denoiser.DenoiserSession(W, B, engine="seq", **kw)
        sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, od.make_query_masks(mm))
        for _ in range(3):
            sess.forward(x, 20, 30, 16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            sess.forward(x, 20, 30, 16)
        e1.record()
        torch.cuda.synchronize()
        print("B=64 (128 sequences) %-32s %8.1f us per forward" % (key, e0.elapsed_time(e1) * 1e3 / 20), flush=True)


if __name__ == "__main__":
    main()
