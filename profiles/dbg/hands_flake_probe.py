"""The eager jitter scenario of tests/test_async_gpu.py with the decoder outputs kept (GestureRepEncoder.debug_keep): when a
decoded pred_hands differs from the synchronous run, is the 6D decoder output `d` (final_layer GEMM) different too, or only
its axis-angle conversion?"""
import importlib, os, random, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
KEYS = ("pred_upper", "pred_lower", "pred_facepose", "pred_hands", "pred_transl", "pred_exps", "prev_latentout")
GI = [2] * 25 + [0] * 25
dev = torch.device("cuda", 0)
cfg = rg.synth.default_model_cfg(num_layers=2)
vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
db = rg.synth.SyntheticDataset(512, seed=11, device=dev, feat_device=dev)
model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=db, device=dev)
model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
model.eval()
model.use_graphs = False
gre = model.model.gesture_rep_encoder
B, N = 4, 6
batches = []
for i in range(N):
    d = rg.synth.synth_batch(B, seed=900 + i, device=dev)
    qs = [rg.synth.synth_query(50 * i + j) for j in range(B)]
    d["discourse"] = [q["discourse"] for q in qs]; d["prominence"] = [q["prominence"] for q in qs]
    d["text_features"] = [q["text_features"].to(dev) for q in qs]
    d["speaker_ids"] = torch.tensor([[q["speaker_id"]] * 150 for q in qs], device=dev)
    batches.append(d)
guided = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
def args(i, flags):
    d = dict(batches[i]); d["trans"] = batches[i]["trans"].clone()
    return dict(d, retrieval_method="discourse", inference_kwargs=dict(flags, noise_tape=rg.synth.NoiseTape(4500 + i)))
want, want_d = {}, {}
for name, flags in (("guided", guided), ("base", {})):
    want[name], want_d[name] = [], []
    for i in range(N):
        gre.debug_keep = []
        out = model(**args(i, flags)); torch.cuda.synchronize()
        want[name].append({k: out[k].clone() for k in KEYS})
        want_d[name].append({p: t.clone() for p, t in gre.debug_keep if t.shape[0] == B * 150})
model.async_results = True
gre.debug_poison = os.environ.get("POISON") == "1"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
for rep in range(reps):
    rng = random.Random(100 + rep)
    def jit(stream, tag):
        if rng.random() < 0.5:
            with torch.cuda.stream(stream):
                torch.cuda._sleep(rng.randrange(1, 6_000_000))
    model._jitter = jit
    for name, flags in (("guided", guided), ("base", {})):
        gre.debug_keep = []
        outs = []
        for i in range(N):
            o = model.submit(**args(i, flags))
            if o is not None: outs.append(o)
        outs += model.flush()
        got = [{k: o[k].clone() for k in KEYS} for o in outs]
        torch.cuda.synchronize()
        kept = [(p, t) for p, t in gre.debug_keep if t.shape[0] == B * 150]      # decodes in tail order = submission order
        assert len(kept) == 4 * N, len(kept)
        for i in range(N):
            for k in KEYS:
                if not torch.equal(got[i][k], want[name][i][k]):
                    bad += 1
                    dd = {p: t for p, t in kept[4 * i:4 * i + 4]}
                    msg = []
                    for p, t in dd.items():
                        w = want_d[name][i][p]
                        ne = (t != w).nonzero()
                        msg.append("%s d differs in %d elements%s" % (p, ne.shape[0], (" rows %s cols %d..%d" % (sorted(set(ne[:, 0].tolist()))[:6], ne[:, 1].min(), ne[:, 1].max())) if ne.shape[0] else ""))
                    nz = (got[i][k] != want[name][i][k]).nonzero()
                    orig = dict.__getitem__(outs[i], k)          # the result tensor itself, after the device-wide sync
                    msg.append("result tensor re-read after sync %s the synchronous run" % ("EQUALS" if torch.equal(orig, want[name][i][k]) else "still differs from"))
                    msg.append("NaNs in the result: %d" % int(torch.isnan(orig).sum()))
                    pk = {"pred_hands": "hands", "pred_upper": "upper"}.get(k)
                    if pk is not None:
                        os.makedirs("gpurun_out", exist_ok=True)
                        np.savez_compressed("gpurun_out/aa_flake_%d_%s_%d_%s.npz" % (rep, name, i, k), got=orig.cpu().numpy(), want=want[name][i][k].cpu().numpy(),
                                            d=dict(kept[4 * i:4 * i + 4])[pk].cpu().numpy(), d_all=torch.stack([dict(kept[4 * b:4 * b + 4])[pk] for b in range(N)]).cpu().numpy(),
                                            want_d_all=torch.stack([want_d[nm2][b][pk] for nm2 in ("guided", "base") for b in range(N)]).cpu().numpy())
                    # whose data is in the wrong region?  f(d) of every kept decoder output of this pass at the same positions
                    from oracle import rotation as orot
                    part = {"pred_hands": "hands", "pred_upper": "upper", "pred_lower": "lowertrans", "pred_facepose": "face"}.get(k)
                    if part is not None and k in ("pred_hands", "pred_upper"):
                        nj = got[i][k].shape[-1] // 3
                        idx = nz.cpu()
                        g_bad = orig.cpu()[idx[:, 0], idx[:, 1], idx[:, 2]]
                        for bi in range(N):
                            cand = dict(kept[4 * bi:4 * bi + 4])[part].cpu().float().reshape(B, 150, -1)[:, :, :nj * 6]
                            aa = orot.sixd_to_aa(cand, nj)
                            c_bad = aa[idx[:, 0], idx[:, 1], idx[:, 2]]
                            err = (c_bad - g_bad).abs().max().item()
                            msg.append("vs f(d of batch %d): max abs %.2e" % (bi, err))
                        # neighbouring rows / clips of the same d
                        aa = orot.sixd_to_aa(dict(kept[4 * i:4 * i + 4])[part].cpu().float().reshape(B, 150, -1)[:, :, :nj * 6], nj).reshape(-1)
                        flat = (idx[:, 0] * 150 + idx[:, 1]) * (nj * 3) + idx[:, 2]
                        for sh in (-3 * 256, -3 * 64, -3 * 16, 3 * 16, 3 * 64, 3 * 256, -nj * 3, nj * 3):
                            f2 = (flat + sh).clamp(0, aa.numel() - 1)
                            msg.append("shift %d: %.2e" % (sh, (aa[f2] - g_bad).abs().max().item()))
                    print("rep %d %s batch %d %s: %d elements differ (first %s) | %s" % (rep, name, i, k, nz.shape[0], nz[0].tolist(), "; ".join(msg)), flush=True)
print("reps %d: mismatching (batch, key) pairs %d" % (reps, bad))
