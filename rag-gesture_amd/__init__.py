"""MI355X-native RAG-Gesture inference hot path."""
from . import synth, schedule, capi, gemm, denoiser, sampler  # noqa: F401
from . import smoke  # noqa: F401
