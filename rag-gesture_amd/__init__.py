"""MI355X-native RAG-Gesture inference hot path."""
import os as _os

# The asynchronous pipeline keeps up to six streams busy at once (the caller's, up to four lanes, the retrieval search);
# the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that
# share a queue serialise.  Read when the runtime initialises (the first HIP call); a value the user has set wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from . import synth, schedule, capi, gemm, denoiser, sampler, vae, pipeline, retrieval, dist, packing, longform, features  # noqa: F401
from .pipeline import MotionDiffusion, ReGestureTransformer, build_architecture, register_with_mmcv  # noqa: F401

register_with_mmcv(force=False)   # no-op without mmcv; never replaces the reference's own classes unless asked to
from . import smoke  # noqa: F401
