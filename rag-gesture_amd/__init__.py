"""MI355X-native RAG-Gesture inference hot path."""
import os as _os

# The asynchronous pipeline keeps up to eleven streams busy at once (the caller's, eight batch lanes, the retrieval search,
# the decode stream);
# the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that
# share a queue serialise.  Read when the runtime initialises (the first HIP call); a value the user has set wins.
# The package must therefore be imported BEFORE the host application's first HIP call (bench.py and `bench.py --gpus N`
# also put the variable into the environment of every rank before torch is imported).  If HIP is already up, setting the
# variable now would change nothing: it is left alone and the shortfall is reported instead (MotionDiffusion.lane_report
# then shows how many of the lane streams were measured to be concurrent when calibrate_lanes=True).
import sys as _sys
_torch = _sys.modules.get("torch")
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    if _torch is not None and _torch.cuda.is_initialized():
        import warnings as _warnings
        _warnings.warn("rag-gesture_amd imported after the HIP runtime was initialised: GPU_MAX_HW_QUEUES keeps the runtime's "
                       "default (4 hardware queues); lanes that share a queue serialise (slower, never incorrect)")
    else:
        _os.environ["GPU_MAX_HW_QUEUES"] = "16"
from . import synth, schedule, capi, gemm, denoiser, sampler, vae, pipeline, retrieval, dist, packing, longform, features  # noqa: F401
from .pipeline import MotionDiffusion, ReGestureTransformer, build_architecture, register_with_mmcv  # noqa: F401

register_with_mmcv(force=False)   # no-op without mmcv; never replaces the reference's own classes unless asked to
from . import smoke  # noqa: F401
