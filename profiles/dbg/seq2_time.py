"""Launch time of the denoiser forward kernels at 128 sequences (B = 64, two step groups).  python profiles/dbg/seq2_time.py [forms]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
rg = importlib.import_module("rag-gesture_amd")
from oracle import denoiser as od  # noqa: E402
forms = sys.argv[1].split(",") if len(sys.argv) > 1 else ["one", "one_pairs", "duo", "duo_pairs"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = rg.synth.default_model_cfg(num_layers=8)
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, rg.schedule.Schedule(), "cuda")
data = rg.synth.synth_batch(B, seed=3)
x = torch.randn(B, 43, 512, device="cuda")
mm = torch.ones(B, 43); mm[:, [10, 21, 32]] = 0
KW = dict(one=dict(seq_duo=False), one_pairs=dict(seq_duo=False, seq_pairs=True), duo=dict(seq_duo=True), duo_pairs=dict(seq_duo=True, seq_pairs=True))
for key in forms:
    sess = rg.denoiser.DenoiserSession(W, B, engine="seq", **KW[key])
    sess.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, od.make_query_masks(mm))
    for _ in range(3):
        sess.forward(x, 20, 30, B // 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        sess.forward(x, 20, 30, B // 4)
    e1.record()
    torch.cuda.synchronize()
    print("%s B=%d (%d sequences) %-10s %8.1f us per forward" % (os.environ.get("RG_LIB_TAG", "product"), B, 2 * B, key, e0.elapsed_time(e1) * 1e3 / 20), flush=True)
