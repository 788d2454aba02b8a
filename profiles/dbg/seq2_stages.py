"""Localise a difference between rg_seq2_forward and rg_seq_forward with the staged register dumps."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rg = importlib.import_module("rag-gesture_amd")
from oracle import denoiser as od  # noqa: E402
L = 2
cfg = rg.synth.default_model_cfg(num_layers=L)
P = rg.synth.synth_denoiser_state(0, cfg)
W = rg.denoiser.DenoiserWeights(P, cfg, rg.schedule.Schedule(), "cuda")
B = 3
data = rg.synth.synth_batch(B, seed=83)
x = torch.from_numpy(np.random.Generator(np.random.PCG64(8)).standard_normal((B, 43, 512)).astype(np.float32)).cuda()
mm = torch.ones(B, 43); mm[:, [10, 21, 32]] = 0; mm[0, 30:] = 0
sess = {}
for key, duo in (("one", False), ("duo", True)):
    sess[key] = rg.denoiser.DenoiserSession(W, B, engine="seq", seq_duo=duo)
    sess[key].set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, od.make_query_masks(mm))
for layer in range(L):
    for stage in (1, 10, 2, 11, 12, 13, 3, 4):
        d = {}
        for key in sess:
            dump = torch.zeros(2 * B, 48, 512, device="cuda")
            sess[key].sq.run(x, 23, dump=dump, dump_stage=stage, dump_layer=layer)
            torch.cuda.synchronize()
            d[key] = dump[:, :43].cpu()
        diff = (d["duo"] - d["one"]).abs().amax(dim=(1, 2))
        print("layer %d stage %2d: max abs diff per sequence %s   (ref max %.3e)" % (layer, stage, ["%.2e" % v for v in diff.tolist()], d["one"].abs().max().item()))
