"""In-kernel wall-clock shares of the sequence-stationary forward (diagnostic build: RG_DIAG=1 python rag-gesture_amd/build.py;
RG_DIAG=1 python profiles/dbg/seq_stamps.py [B])."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
assert os.environ.get("RG_DIAG") == "1"
rg = importlib.import_module("rag-gesture_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = rg.synth.default_model_cfg(num_layers=8)
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, rg.schedule.Schedule(), "cuda")
sess = rg.denoiser.DenoiserSession(W, B, engine="seq")
d = rg.synth.synth_batch(B, seed=1)
mask = torch.ones(B, 43); mask[:, [10, 21, 32]] = 0
sess.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, {c: torch.ones(B, 43) for c in rg.denoiser.CONDS})
x = torch.randn(B, 43, 512, device="cuda")
dump = torch.zeros(2 * B * 48 * 512, device="cuda")
for _ in range(3):
    sess.sq.run(x, 30, dump=dump, dump_stage=99)
torch.cuda.synchronize()
t = dump[:2 * B * 64].view(2 * B, 8, 8).cpu() / 100.0      # us
names = ["unit GEMMs", "row statistics", "barriers", "styl params+panel", "k softmax", "kernel"]
for tag, sl in (("conditional", slice(0, B)), ("classifier-free", slice(B, 2 * B))):
    m = t[sl].mean(dim=(0, 1))
    print("%s sequences (mean over workgroups and waves, us): " % tag + "  ".join("%s %.1f" % (names[i], m[i]) for i in (0, 1, 2, 3, 4, 5))
          + "  rest %.1f" % (m[5] - m[0] - m[1] - m[2] - m[3] - m[4]) + "   kernel min / max over workgroups %.1f / %.1f" % (t[sl, :, 5].min(), t[sl, :, 5].max()))
print("per wave (conditional workgroups, mean us): " + "  ".join("w%d: gemm %.0f stats %.0f bar %.0f" % (w, t[:B, w, 0].mean(), t[:B, w, 1].mean(), t[:B, w, 2].mean()) for w in range(8)))
