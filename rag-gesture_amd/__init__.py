"""MI355X-native RAG-Gesture inference hot path."""
from . import synth, schedule, capi, gemm, denoiser, sampler, vae, pipeline, retrieval, dist, packing, longform  # noqa: F401
from .pipeline import MotionDiffusion, ReGestureTransformer, build_architecture  # noqa: F401
from . import smoke  # noqa: F401
