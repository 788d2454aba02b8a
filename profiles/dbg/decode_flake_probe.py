"""Reproduce the rare 48-element mismatch of a decoded output (jitter test, eager base lanes; bench sync-vs-sync) and print
where it is: key, clip, frames, columns, magnitude; whether the latent differs; whether a second synchronous run agrees."""
import importlib, os, random, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
KEYS = ("pred_upper", "pred_lower", "pred_facepose", "pred_hands", "pred_transl", "pred_exps", "prev_latentout")
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
use_graphs = (sys.argv[2] != "eager") if len(sys.argv) > 2 else False
mode = sys.argv[3] if len(sys.argv) > 3 else "submit"          # submit | sync
dev = torch.device("cuda", 0)
cfg = rg.synth.default_model_cfg(num_layers=2)
vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs), database=None, device=dev)
model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
model.eval()
model.use_graphs = use_graphs
B, N = 4, 6
batches = [rg.synth.synth_batch(B, seed=900 + i, device=dev) for i in range(N)]

def args(i):
    d = dict(batches[i]); d["trans"] = batches[i]["trans"].clone()
    return dict(d, retrieval_method="discourse", inference_kwargs=dict(noise_tape=rg.synth.NoiseTape(4500 + i)))

want = []
for i in range(N):
    out = model(**args(i)); torch.cuda.synchronize()
    want.append({k: out[k].clone() for k in KEYS})
rng = random.Random(3)
def jit(stream, tag):
    if rng.random() < 0.5:
        with torch.cuda.stream(stream):
            torch.cuda._sleep(rng.randrange(1, 6_000_000))
bad = 0
t0 = time.time()
for rep in range(passes):
    if mode == "submit":
        model.async_results = True
        model._jitter = jit
        outs = []
        for i in range(N):
            o = model.submit(**args(i))
            if o is not None: outs.append(o)
        outs += model.flush()
        got = [{k: o[k].clone() for k in KEYS} for o in outs]
    else:
        model.async_results = False
        got = []
        for i in range(N):
            o = model(**args(i))
            got.append({k: o[k].clone() for k in KEYS})
    torch.cuda.synchronize()
    for i in range(N):
        for k in KEYS:
            a, b = got[i][k], want[i][k]
            if not torch.equal(a, b):
                bad += 1
                d = (a != b).nonzero().tolist()
                clips = sorted(set(x[0] for x in d)); rows = sorted(set(x[1] for x in d)); cols = sorted(set(x[2] for x in d))
                print("pass %d batch %d %s: %d elements, max %.3e | clips %s rows %s cols %s" % (rep, i, k, len(d), (a - b).abs().max(), clips, rows, cols), flush=True)
print("mode %s graphs %s: %d passes, %.0f s, mismatching (batch, key) pairs: %d" % (mode, use_graphs, passes, time.time() - t0, bad))
