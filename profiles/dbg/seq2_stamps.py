"""In-kernel wall-clock shares of the two-sequence denoiser forward (diagnostic build: RG_DIAG=1 python rag-gesture_amd/build.py;
RG_DIAG=1 python profiles/dbg/seq2_stamps.py [B] [pairs])."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
assert os.environ.get("RG_DIAG") == "1"
rg = importlib.import_module("rag-gesture_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pairs = len(sys.argv) > 2 and sys.argv[2] == "pairs"
cfg = rg.synth.default_model_cfg(num_layers=8)
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, rg.schedule.Schedule(), "cuda")
sess = rg.denoiser.DenoiserSession(W, B, engine="seq", seq_duo=True, seq_pairs=pairs)
d = rg.synth.synth_batch(B, seed=1)
mask = torch.ones(B, 43); mask[:, [10, 21, 32]] = 0
sess.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, {c: torch.ones(B, 43) for c in rg.denoiser.CONDS})
x = torch.randn(B, 43, 512, device="cuda")
dump = torch.zeros(2 * B * 48 * 512, device="cuda")
for _ in range(3):
    sess.sq.run(x, 30, dump=dump, dump_stage=99)
torch.cuda.synchronize()
nwg = B // 2 if pairs else B
t = dump[:nwg * 96].view(nwg, 8, 12).cpu() / 100.0      # us; pairs: the stamps of the LAST pass (classifier-free pair)
names = ["unit GEMMs", "row statistics", "barriers", "params+panel", "attention math", "pass", "xbuf/gbuf", "consume waits",
         "panel writes outside units", "mix_x scalings / table adds", "prologue", "-"]
# workgroup -> kind (rg_seq2_kernel: XCD-interleaved when the pair count per kind is a multiple of 4)
npc = B // 2
kinds = [((b & 7) >= 4) if npc % 4 == 0 else (b >= npc) for b in range(nwg)] if not pairs else [True] * nwg
summary = {}
for tag, k in (("conditional pairs", False), ("classifier-free pairs", True)):
    idx = [b for b in range(nwg) if kinds[b] == k]
    if not idx:
        continue
    m = t[idx].mean(dim=(0, 1))
    summary["classifier_free_pass_us" if k else "conditional_pass_us"] = round(float(m[5]), 1)
    summary[("classifier_free" if k else "conditional") + "_categories_us"] = {names[i]: round(float(m[i]), 1) for i in range(11)}
    rest = m[5] - m[0] - m[1] - m[2] - m[3] - m[4] - m[6] - m[8] - m[9] - m[10]
    print("%s (mean over workgroups and waves, us): " % tag + "  ".join("%s %.1f" % (names[i], m[i]) for i in (0, 1, 2, 3, 4, 6, 8, 9, 10, 7, 5))
          + "  rest %.1f" % rest + "   pass min / max over workgroups %.1f / %.1f" % (t[idx][:, :, 5].min(), t[idx][:, :, 5].max()))
    print("   per wave: " + "  ".join("w%d: gemm %.0f stats %.0f bar %.0f" % (w, t[idx][:, w, 0].mean(), t[idx][:, w, 1].mean(), t[idx][:, w, 2].mean()) for w in range(8)))

# per gemm_frags call (wave 0 and wave 4 of the first conditional and the first classifier-free workgroup), layer 3
log = dump[(1 << 20):(1 << 20) + nwg * 8 * 512].view(nwg, 8, 512).cpu() / 100.0
for tag, k in (("conditional", False), ("classifier-free", True)):
    idx = [b for b in range(nwg) if kinds[b] == k]
    if not idx:
        continue
    b = idx[0]
    n = int((log[b, 0] > 0).sum())
    per_layer = (n - 2) // 8
    print("%s pair, workgroup %d: %d gemm_frags calls (%d per layer); layer 3, us per call [wave 0 | wave 4 | mean over the kind's workgroups, waves 0-3 | 4-7]:" % (tag, b, n, per_layer))
    for c in range(1 + 3 * per_layer, 1 + 4 * per_layer):
        print("   call %2d: %6.2f | %6.2f | %6.2f | %6.2f" % (c - 1 - 3 * per_layer, log[b, 0, c], log[b, 4, c], log[idx][:, :4, c].mean(), log[idx][:, 4:, c].mean()))
    print("   embed %.2f  head %.2f" % (log[b, 0, 0], log[b, 0, n - 1]))

if os.environ.get("STAMPS_JSON"):
    import hashlib, json
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "rag-gesture_amd", "csrc", "rg_seq2.hip")
    summary["kernel_source_sha256"] = hashlib.sha256(open(src, "rb").read()).hexdigest()
    summary["what"] = "in-kernel wall-clock stamps of rg_seq2_kernel (diagnostic build, RG_DIAG=1), mean over workgroups and waves, B = %d, pairs %s" % (B, pairs)
    with open(os.environ["STAMPS_JSON"], "w") as f:
        json.dump(summary, f, indent=1)
