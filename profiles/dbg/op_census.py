"""Which Python call sites of the package issue the small device operations of a submit(): counts of torch calls (copy_, clone,
to, cat, stack, __setitem__, __getitem__ on device tensors, zeros / empty / randn ...) per call site over N steady-state submissions."""
import collections, importlib, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
B, N = 16, 6
GI = [2] * 25 + [0] * 25
cfg = rg.synth.default_model_cfg(num_layers=8)
vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
database = rg.synth.SyntheticDataset(32768, seed=2025, device=dev, feat_device=dev)
model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=database, device=dev, calibrate_lanes=False)
model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
model.eval()
model.async_results = True
data = rg.synth.synth_batch(B, seed=1234, device=dev)
qs = [rg.synth.synth_query(i) for i in range(B)]
data["discourse"] = [q["discourse"] for q in qs]
data["prominence"] = [q["prominence"] for q in qs]
data["text_features"] = [q["text_features"].to(dev) for q in qs]
data["speaker_ids"] = torch.tensor([[q["speaker_id"]] * 150 for q in qs], device=dev)
trans0 = data["trans"].clone()
noise = rg.pipeline.DeviceNoise(dev, seed=1)


def step():
    d = dict(data)
    d["trans"] = trans0.clone()
    model.model.database.test_indexes.clear()
    return model.submit(**dict(d, retrieval_method="discourse", inference_kwargs=dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1, noise_tape=noise)))


for _ in range(10):
    step()
model.flush()
for _ in range(6):
    step()
torch.cuda.synchronize()
if os.environ.get("TRACE"):      # under rocprofv3 --kernel-trace: N submissions between two marker kernels (profiles/dbg/r06_main_census.sh)
    torch.cuda._sleep(30_000_000)
    for _ in range(N):
        step()
    torch.cuda._sleep(30_000_000)
    torch.cuda.synchronize()
    model.flush()
    torch.cuda.synchronize()
    sys.exit(0)
counts = collections.Counter()


def wrap(owner, name):
    fn = getattr(owner, name)

    def w(*a, **k):
        fr = [f for f in traceback.extract_stack()[:-1] if "rag-gesture_amd" in f.filename]
        if fr:
            f = fr[-1]
            counts[("%s.%s" % (getattr(owner, "__name__", "Tensor"), name), "%s:%d" % (os.path.basename(f.filename), f.lineno))] += 1
        return fn(*a, **k)
    setattr(owner, name, w)


for owner, names in ((torch.Tensor, ("copy_", "clone", "to", "contiguous", "__setitem__", "__getitem__", "index_copy_", "index_select", "float", "long",
                                     "fill_", "zero_", "view", "expand", "unsqueeze", "record_stream", "cuda", "cpu", "item", "tolist")),
                     (torch, ("cat", "stack", "zeros", "empty", "ones", "tensor", "randn", "zeros_like", "empty_like", "full"))):
    for nm in names:
        wrap(owner, nm)
for _ in range(N):
    step()
torch.cuda.synchronize()
print("per submission (over %d), top call sites:" % N)
for (op, site), c in counts.most_common(60):
    print("  %6.1f  %-22s %s" % (c / N, op, site))
