"""MI355X-native RAG-Gesture inference hot path."""
from . import synth, schedule, capi, gemm, denoiser, sampler, vae, pipeline, retrieval, dist, packing, longform, features  # noqa: F401
from .pipeline import MotionDiffusion, ReGestureTransformer, build_architecture, register_with_mmcv  # noqa: F401

register_with_mmcv(force=False)   # no-op without mmcv; never replaces the reference's own classes unless asked to
from . import smoke  # noqa: F401
