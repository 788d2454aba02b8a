"""Diagnostic: host-side cost centres of RetrievalDatabase.forward in the guided bench workload (cProfile, top entries)."""
import cProfile
import importlib
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
B = 16
GI = [2] * 25 + [0] * 25
cfg = rg.synth.default_model_cfg(num_layers=8)
vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
database = rg.synth.SyntheticDataset(32768, seed=2025, device=dev, feat_device=dev)
model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=database, device=dev)
model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
model.eval()
data = rg.synth.synth_batch(B, seed=1234, device=dev)
qs = [rg.synth.synth_query(i) for i in range(B)]
data["discourse"] = [q["discourse"] for q in qs]
data["prominence"] = [q["prominence"] for q in qs]
data["text_features"] = [q["text_features"].to(dev) for q in qs]
data["speaker_ids"] = torch.tensor([[q["speaker_id"]] * 150 for q in qs], device=dev)
trans0 = data["trans"].clone()


def one_step():
    d = dict(data)
    d["trans"] = trans0.clone()
    model.model.database.test_indexes.clear()
    ikw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
    model(**dict(d, retrieval_method="discourse", inference_kwargs=ikw))


one_step(); one_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    one_step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
