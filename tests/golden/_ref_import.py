"""Import harness for the upstream reference (runs ONLY in the build container).

The reference at /root/reference needs mmcv, cv2, fairseq, lmdb, ... which are not
installed here.  None of them contribute arithmetic to the inference hot path
(SURVEY.md section 8c), so this module registers minimal stand-in *modules* (a
registry, BaseModule = nn.Module, no-op names) in ``sys.modules`` and then imports
the reference's hot-path modules by file path under the ``mogen.*`` names.

This file is test infrastructure for generating golden vectors.  It is never
imported by the product, by ``-m gpu`` tests, by smoke() or by bench.py, and it
does nothing useful on the GPU box (where /root/reference does not exist).
"""
import importlib
import importlib.machinery
import os
import sys
import types

# /root/reference is read-only by contract: from the moment this harness is imported no import writes a __pycache__
# (set here, at import time, so that it precedes the first reference import whatever script drives the harness)
sys.dont_write_bytecode = True

REF_ROOT = os.environ.get("RG_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "mogen", "models"))


class _Registry:
    """20-line stand-in for mmcv.utils.Registry (register_module/get/build)."""

    def __init__(self, name, parent=None, build_func=None):
        self.name = name
        self._parent = parent
        self._modules = parent._modules if parent is not None else {}
        self.build_func = build_func or _build_from_cfg

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            self._modules[name or cls.__name__] = cls
            return cls

        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self._modules[key]

    def build(self, cfg, default_args=None):
        return self.build_func(cfg, self, default_args)


def _build_from_cfg(cfg, registry, default_args=None):
    cfg = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            cfg.setdefault(k, v)
    cls = registry.get(cfg.pop("type"))
    return cls(**cfg)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=True)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


_loaded = None


def load_reference():
    """Returns a namespace with the reference's hot-path modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True  # never write __pycache__ into the reference mount
    import torch.nn as nn
    import transformers  # noqa: F401  (the reference imports it at module scope)

    models_registry = _Registry("model")
    _stub("mmcv")
    _stub("mmcv.cnn", MODELS=models_registry)
    _stub("mmcv.utils", Registry=_Registry)
    _stub("mmcv.runner", BaseModule=_BaseModule(nn))
    _stub("mmcv.parallel")
    _stub("cv2", norm=lambda *a, **k: None)
    _stub("fairseq")
    _stub("lmdb")
    _stub("librosa")
    _stub("fuzzywuzzy", fuzz=types.SimpleNamespace(partial_ratio=lambda a, b: 0))
    _stub("dotenv", load_dotenv=lambda *a, **k: None)
    _stub("openai", OpenAI=object)
    _stub("kornia")
    _stub("kornia.filters")
    _stub("kornia.filters.kernels", laplacian_1d=lambda *a, **k: None)

    mroot = os.path.join(REF_ROOT, "mogen")
    _pkg("mogen", mroot)
    _pkg("mogen.models", os.path.join(mroot, "models"))
    for sub in ("utils", "attentions", "transformers", "architectures", "losses"):
        _pkg("mogen.models." + sub, os.path.join(mroot, "models", sub))
    _pkg("mogen.models.transformers.rag", os.path.join(mroot, "models", "transformers", "rag"))

    ns = types.SimpleNamespace()
    ns.builder = importlib.import_module("mogen.models.builder")
    ns.gd = importlib.import_module("mogen.models.utils.gaussian_diffusion")
    ns.rc = importlib.import_module("mogen.models.utils.rotation_conversions")
    ns.detr = importlib.import_module("mogen.models.utils.detr_utils")
    ns.styl = importlib.import_module("mogen.models.utils.stylization_block")
    ns.attn = importlib.import_module("mogen.models.attentions.efficient_attention")
    ns.vae = importlib.import_module("mogen.models.transformers.gesture_vae")
    ns.dt = importlib.import_module("mogen.models.transformers.diffusion_transformer")
    ns.rag_utils = importlib.import_module("mogen.models.transformers.rag.utils")
    ns.discourse = importlib.import_module("mogen.models.transformers.rag.discourse_retrieval")
    ns.raggesture = importlib.import_module("mogen.models.transformers.raggesture")
    importlib.import_module("mogen.models.losses.mse_loss")
    ns.arch = importlib.import_module("mogen.models.architectures.diffusion_architecture")
    import torch

    torch.autograd.set_detect_anomaly(False)  # the reference turns it on at import
    _loaded = ns
    return ns


def _BaseModule(nn):
    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

    return BaseModule
