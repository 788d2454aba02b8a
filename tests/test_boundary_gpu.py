"""GPU: the drop-in boundary as the reference's tools use it (tools/visualize.py:138-147, 189-200):

    model = build_architecture(cfg.model, database=train_dataset)      # registry lookup by cfg.model.type
    load_checkpoint(model, args.checkpoint, map_location="cpu")        # mmcv.runner
    model = MMDataParallel(model, device_ids=[0]); model.eval()
    with torch.no_grad(): output = model(**data)

mmcv is not installed here, so its three pieces are restated from mmcv 1.7.2 (requirements.txt:11) in a few lines each:
`Registry.get`, `runner.checkpoint.load_checkpoint` (strip `module.`, walk `_load_from_state_dict` over the module
tree) and MMDataParallel's single-device path (= torch.nn.DataParallel: scatter kwargs to the device, call the module).
Also: body-part VAE checkpoints in the reference's own format ({"model_state": ...}, `module.`-prefixed) next to
their YAMLs (diffusion_transformer.py:151-188)."""
import os
import re

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


class Registry:   # mmcv.utils.Registry: name -> class
    def __init__(self):
        self._module_dict = {}

    def register_module(self, name=None, module=None, force=False):
        if not force and name in self._module_dict:
            raise KeyError("%s is already registered" % name)
        self._module_dict[name] = module

    def get(self, key):
        return self._module_dict.get(key)


def build_architecture(cfg, registry, **kwargs):   # mogen/models/builder.py:22-26
    cfg = dict(cfg)
    return registry.get(cfg.pop("type"))(**cfg, **kwargs)


def mmcv_load_checkpoint(model, filename, map_location="cpu", strict=False):
    """mmcv/runner/checkpoint.py (1.7.2): load_checkpoint -> load_state_dict."""
    checkpoint = torch.load(filename, map_location=map_location)
    state_dict = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
    state_dict = {re.sub(r"^module\.", "", k): v for k, v in state_dict.items()}
    unexpected_keys, all_missing_keys, err_msg = [], [], []

    def load(module, prefix=""):
        module._load_from_state_dict(state_dict, prefix, {}, True, all_missing_keys, unexpected_keys, err_msg)
        for name, child in module._modules.items():
            if child is not None:
                load(child, prefix + name + ".")

    load(model)
    assert not all_missing_keys and not unexpected_keys and not err_msg, (all_missing_keys, unexpected_keys, err_msg)
    return checkpoint


def test_tools_plumbing_runs_unchanged(rg, tmp_path):
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder", num_layers=2)
    P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
    model_cfg = rg.synth.reference_style_model_cfg(cfg, vae_cfgs)
    assert model_cfg["type"] == "MotionDiffusion"

    # ---- the VAEs as the reference ships them: one YAML + one checkpoint per body part in a directory
    vae_dir = tmp_path / "vae"
    vae_dir.mkdir()
    vcfg = dict(model_cfg["model"]["vae_cfg"])
    for part in rg.vae.PARTS:
        pre = "gesture_rep_encoder.%s_vae." % part
        sd = {("module." + k[len(pre):]): v for k, v in P.items() if k.startswith(pre)}
        assert sd
        torch.save({"model_state": sd, "epoch": 3}, vae_dir / ("%s.bin" % part))
        y = dict(vae_cfgs[part], test_ckpt="/somewhere/else/%s.bin" % part)   # only the basename counts (:156)
        with open(vae_dir / ("%s.yaml" % part), "w") as f:
            yaml.safe_dump(y, f)
        vcfg["%s_cfg" % part] = str(vae_dir / ("%s.yaml" % part))
    model_cfg = dict(model_cfg, model=dict(model_cfg["model"], vae_cfg=vcfg))

    # ---- the diffusion checkpoint in mmcv format WITHOUT the VAE weights
    den = {"model." + k: v for k, v in P.items() if not k.startswith("gesture_rep_encoder.")}
    ckpt = tmp_path / "epoch_1.pth"
    torch.save({"meta": {}, "state_dict": den}, ckpt)

    # ---- tools/visualize.py:138-147 with this implementation registered under the reference's names
    registry = Registry()
    registry.register_module(name="MotionDiffusion", module=rg.MotionDiffusion, force=True)
    registry.register_module(name="ReGestureTransformer", module=rg.ReGestureTransformer, force=True)
    model = build_architecture(model_cfg, registry, database=None)
    assert isinstance(model, torch.nn.Module)
    mmcv_load_checkpoint(model, str(ckpt), map_location="cpu")
    model = torch.nn.DataParallel(model.cuda(), device_ids=[0])
    model.eval()
    assert model.module.training is False

    # ---- :189-200
    data = rg.synth.synth_batch(2, seed=5)          # CPU tensors + python lists, as the collate function hands them over
    ikw = dict(noise_tape=rg.synth.NoiseTape(1))
    with torch.no_grad():
        out = model(**dict(data, retrieval_method="discourse", inference_kwargs=ikw))
    assert out["pred_upper"].shape == (2, 150, 39) and out["pred_upper"].is_cuda

    # ---- the same model loaded the direct way from the full state: identical results
    ref = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs), database=None)
    keys = ref.load_state_dict(P)
    assert keys.missing_keys == [] and keys.unexpected_keys == []
    ref.eval()
    out2 = ref(**dict(rg.synth.synth_batch(2, seed=5), retrieval_method="discourse",
                      inference_kwargs=dict(noise_tape=rg.synth.NoiseTape(1))))
    for k in ("pred_upper", "pred_hands", "pred_transl", "prev_latentout"):
        assert torch.equal(out[k], out2[k]), k
    # state_dict(): the reference-format tensors, `model.`-prefixed like the reference module's own keys
    sd = model.module.state_dict()
    assert set(k for k in sd if not k.startswith("model.gesture_rep_encoder.")) == set(den)
    assert any(k.startswith("model.gesture_rep_encoder.upper_vae.") for k in sd)


def test_training_mode_is_refused(rg):
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder", num_layers=2)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs), database=None)
    with pytest.raises(rg.capi.RgError):
        model.train()
    assert model.eval() is model
