"""Drop-in for the reference's inference entry points:

    model = build_architecture(cfg.model, database=train_dataset)   # mogen/models/builder.py:22-26
    model.load_state_dict(ckpt["state_dict"]); model.eval()
    with torch.no_grad(): out = model(**data)                       # tools/visualize.py:189-200

`MotionDiffusion.forward(**kwargs)` mirrors the eval branch of
mogen/models/architectures/diffusion_architecture.py:117-176, 213-582 (same kwargs, same
`inference_kwargs` flags and compatibility asserts, same result keys; the input dict itself is
returned, mutated, exactly like the reference's `results = kwargs`).  All arithmetic runs in the
HIP extension; there is no CPU fallback.

Differences that are deliberate and documented in DESIGN.md:
  * exemplar DDIM inversions run batched (the reference loops at batch 1);
  * clips are independent between VAE encode and VAE decode, and at 43 latent tokens per clip a
    single kernel chain cannot fill 256 CUs (most kernels are latency-, not throughput-bound): the
    batch is cut into `lanes` groups of clips, each with its own HIP stream (checked to sit on its own
    hardware queue), denoiser session and captured graphs (inversion -> splice -> sampling);
  * randomness: by default torch's generator on the device; `inference_kwargs["noise_tape"]`
    (an object with draw(shape)) replays explicit noise in the reference's consumption order
     for parity tests.
"""
import collections
import ctypes
import contextlib
import copy
import os
import time
import types

import torch
import yaml

from . import capi, denoiser, sampler, schedule as sched_mod, seqfwd, vae as vae_mod

MODELS = {}


def register_module(cls):
    MODELS[cls.__name__] = cls
    return cls


def register_with_mmcv(force=False):
    """Put MotionDiffusion / ReGestureTransformer into the registries the reference's tools build from
    (mogen/models/builder.py:11-26: `MODELS = Registry('models', parent=mmcv.cnn.MODELS)`, ARCHITECTURES = SUBMODULES =
    MODELS; classes are looked up by `cfg.model.type`).  force=True replaces the reference's own classes of the same
    names: after that one call `tools/visualize.py:138-147` (build_architecture, load_checkpoint, MMDataParallel,
    model.eval()) and `:189-200` (model(**data)) run unchanged on this implementation.  Returns the registries touched
    (empty when mmcv is not installed)."""
    regs = []
    try:
        from mmcv.cnn import MODELS as mmcv_models
        regs.append(mmcv_models)
    except ImportError:
        return regs
    try:
        from mogen.models.builder import MODELS as mogen_models
        regs.append(mogen_models)
    except ImportError:
        pass
    for reg in regs:
        for cls in (MotionDiffusion, ReGestureTransformer):
            try:
                reg.register_module(name=cls.__name__, module=cls, force=force)
            except (KeyError, TypeError):   # already registered and not forced
                pass
    return regs


def read_vae_checkpoint(path):
    """State dict of one body-part VAE as the reference stores it: {"model_state": ...}, keys optionally prefixed
    `module.` by DataParallel (diffusion_transformer.py:169-188 `load_checkpoints`)."""
    states = torch.load(path, map_location="cpu")
    sd = states["model_state"] if "model_state" in states else states
    if sd and all(k.startswith("module.") for k in sd):
        sd = collections.OrderedDict((k[7:], v) for k, v in sd.items())
    return sd


def build_architecture(cfg, **kwargs):
    """Same contract as mogen/models/builder.py:22-26 (pops `type`, instantiates)."""
    cfg = dict(cfg)
    return MODELS[cfg.pop("type")](**cfg, **kwargs)


def build_submodule(cfg, **kwargs):
    cfg = dict(cfg)
    return MODELS[cfg.pop("type")](**cfg, **kwargs)


def guidance_iters_preset(name, steps=50):
    """The named per-timestep guidance schedules of the reference's tools (tools/visualize.py:74-95,
    tools/longform_synthesis.py: same table); index = respaced timestep (49 = noisiest), value = gradient steps."""
    half = steps // 2
    table = {
        "all_one": [1] * steps, "all_zero": [0] * steps, "all_10": [10] * steps,
        "decreasing": list(range(steps)),
        "increasing": list(range(steps - 1, -1, -1)),
        "drop_decreasing_till_25": [0] * half + list(range(steps))[half:steps],
        "step_increasing_from_25": list(range(steps - 1, -1, -1))[:half] + [0] * half,
        "decreasing_till_25": [0] * half + list(range(half)),
        "increasing_from_25": list(range(half - 1, -1, -1)) + [0] * half,
    }
    if name not in table:
        raise ValueError("Invalid guidance_iters value")
    return table[name]


E_BUCKET = 4   # exemplar counts are padded to a multiple of this: bounded graph / session caches under real retrieval


def bucket(n, m=None):
    m = E_BUCKET if m is None else m     # (read at call time: tests compare against an unpadded run)
    return -(-n // m) * m


def pad_rows(t, n):
    """t [E, ...] -> [n, ...]: padding rows repeat row 0 (valid data: the padded exemplars are inverted like the real
    ones and never spliced anywhere)."""
    return t if t.shape[0] == n else torch.cat([t, t[:1].expand(n - t.shape[0], *t.shape[1:])], dim=0)


class _TorchNoise:
    order_free = True  # draws come from torch's generator: consumers may batch / reorder them

    def __init__(self, device, generator=None):
        self.device, self.generator = device, generator

    def draw(self, shape, device=None):
        return torch.randn(*shape, device=self.device, generator=self.generator)


class DeviceNoise(_TorchNoise):
    """`inference_kwargs["noise_tape"]` drawing on the device from a generator of its own: a run can be repeated with the
    same noise (`state()` before the call, `set_state()` before the repetition) without host-side noise tensors --
    every draw of a batch is made inside forward() / submit(), in one fixed order, whichever schedule then executes it.
    bench.py verifies the batches it times this way."""

    def __init__(self, device, seed=0):
        g = torch.Generator(device=device)
        g.manual_seed(int(seed))
        super().__init__(torch.device(device), g)

    def state(self):
        return self.generator.get_state()

    def set_state(self, state):
        self.generator.set_state(state)


@register_module
class ReGestureTransformer:
    """Configuration holder mirroring raggesture.py:887-922 / diffusion_transformer.py:335-420
    constructor keys; the packed device weights are built in load_state_dict()."""

    def __init__(self, retrieval_cfg=None, scale_func_cfg=None, per_joint_scale=None, retrieval_train=False,
                 use_retrieval_for_test=False, input_feats=None, max_seq_len=240, frame_chunk_size=16,
                 latent_dim=512, time_embed_dim=2048, num_layers=8, sa_block_cfg=None, ca_block_cfg=None,
                 vae_cfg=None, ffn_cfg=None, text_encoder=None, audio_encoder=None, speaker_embedding=None,
                 use_cache_for_text=False, init_cfg=None, body_part_cat_axis="time", database=None, device="cuda",
                 **_unused):
        if _unused:
            raise capi.RgError("ReGestureTransformer: unknown configuration key(s) %s" % ", ".join(sorted(_unused)))
        capi.require(not retrieval_train, "retrieval_train is a training-time switch (raggesture.py:900 asserts it off)")
        capi.require(body_part_cat_axis == "time", "Only time axis is supported for body part categorization")
        for enc in (text_encoder, audio_encoder):
            capi.require(enc is None or (enc.get("pretrained_model") is None and enc.get("num_layers", 0) == 0
                                   and not enc.get("use_text_proj", False)),
                    "only the shipped configuration (pre-extracted features, pre_proj only) is on the hot path")
        self.cfg = dict(
            latent_dim=latent_dim, time_embed_dim=time_embed_dim, num_layers=num_layers,
            num_heads=(sa_block_cfg or {}).get("num_heads", 16), ff_size=(ffn_cfg or {}).get("ffn_dim", 1024),
            max_seq_len=max_seq_len, frame_chunk_size=frame_chunk_size,
            text_latent_dim=(text_encoder or {}).get("latent_dim", 768),
            num_speakers=(speaker_embedding or {}).get("num_speakers", 25),
            scale_func_cfg=scale_func_cfg, per_joint_scale=per_joint_scale,
        )
        capi.require(scale_func_cfg is not None, "the shipped config always runs the CFG mix (scale_func_cfg)")
        self.vae_cfgs, self.vae_states = self._read_vae_cfgs(vae_cfg)
        self.retrieval_cfg, self.use_retrieval_for_test = retrieval_cfg, use_retrieval_for_test
        self.database = None
        if retrieval_cfg is not None and use_retrieval_for_test:
            from . import retrieval
            self.database = retrieval.RetrievalDatabase(**retrieval_cfg, dataset=database, device=device)
        self.weights = self.gesture_rep_encoder = None

    @staticmethod
    def _read_vae_cfgs(vae_cfg):
        """vae_cfg holds YAML paths (diffusion_transformer.py:151-154) or, for synthetic models,
        the dicts themselves under the same keys.  Like the reference's `load_vae` (:151-168) the YAML's `test_ckpt`
        names the VAE checkpoint that sits NEXT TO the YAML; its weights are read here and used for every
        `gesture_rep_encoder.<part>_vae.*` key the diffusion checkpoint does not carry itself."""
        out, states = {}, {}
        for part in vae_mod.PARTS:
            v = vae_cfg["%s_cfg" % part]
            base = None
            if isinstance(v, str):
                base = os.path.dirname(v)
                with open(v, "r", encoding="utf-8") as f:
                    v = yaml.safe_load(f)
            out[part] = dict(v)
            out[part].setdefault("frame_chunk_size", vae_cfg.get("frame_chunk_size", 15))
            ck = out[part].get("test_ckpt")
            path = None
            if ck:
                path = os.path.join(base, os.path.basename(ck)) if base is not None else ck
            states[part] = read_vae_checkpoint(path) if path and os.path.exists(path) else None
        return out, states

    def post_process(self, motion):
        return motion


IncompatibleKeys = collections.namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])


class PendingLatent:
    """The final latent `prev_latentout` of a batch that has been submitted but not sampled yet, as a value for a LATER
    batch's `inference_kwargs["prev_latent"]` (long-form synthesis: window k + 1 takes window k's latent, but only its
    sampling loop needs it -- its retrieval and exemplar inversion do not, and share their launches with window k's sampling).
    Bound when the consuming batch's sampling loop is queued; `select(rows)` picks clips of the producing batch."""

    def __init__(self, state, rows=None):
        self.state, self.rows = state, rows

    def select(self, rows):
        rows = [int(r) for r in rows]
        return self if rows == list(range(self.state.B)) else PendingLatent(self.state, rows)


class AsyncResults(dict):
    """Result dict of an asynchronously submitted batch (MotionDiffusion(async_results=True), submit() / flush()): the
    tensors are produced on the batch's own stream, and the FIRST READ of any entry makes the reader's current stream wait for
    the batch (`done_event`) -- once per reading stream, a device-side wait, the host is not blocked.  Code written against the
    reference's synchronous results (tools/visualize.py:201-260 indexes `output[...]` right after the call) therefore reads
    finished tensors without knowing about the pipeline; `done_event` / `done_stream` themselves are read without waiting."""
    _PLAIN = ("done_event", "done_stream")

    def _ready(self):
        ev = dict.get(self, "done_event")
        if ev is None:
            return
        st = torch.cuda.current_stream()
        seen = self.__dict__.setdefault("_waited", set())
        if st.cuda_stream not in seen:
            seen.add(st.cuda_stream)
            st.wait_event(ev)
            MotionDiffusion._used_on(st, *[v for v in dict.values(self) if torch.is_tensor(v)])

    def __getitem__(self, k):
        if k not in self._PLAIN:
            self._ready()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        if k not in self._PLAIN:
            self._ready()
        return dict.get(self, k, default)

    def values(self):
        self._ready()
        return dict.values(self)

    def items(self):
        self._ready()
        return dict.items(self)

    def pop(self, k, *default):
        if k not in self._PLAIN:
            self._ready()
        return dict.pop(self, k, *default)
    def popitem(self):
        self._ready()
        return dict.popitem(self)

    def setdefault(self, k, default=None):
        if k not in self._PLAIN:
            self._ready()
        return dict.setdefault(self, k, default)

    # dict(out), {**out}, f(**out) and other.update(out) copy a dict SUBCLASS entry by entry in C, without calling any
    # of the methods above, as long as the subclass inherits dict's own __iter__: overriding it sends them through
    # keys() + __getitem__ (CPython dict_merge), i.e. through the wait.
    def __iter__(self):
        self._ready()
        return dict.__iter__(self)

    def copy(self):
        """A plain dict of finished tensors (the wait is issued on the current stream)."""
        self._ready()
        return dict(dict.items(self))

    __copy__ = copy

    def __deepcopy__(self, memo):
        self._ready()
        return {k: copy.deepcopy(v, memo) for k, v in dict.items(self) if k not in self._PLAIN}

    def __reduce__(self):
        self._ready()
        return (dict, (), None, None, iter([(k, v) for k, v in dict.items(self) if k not in self._PLAIN]))


@register_module
class MotionDiffusion(torch.nn.Module):
    """nn.Module like the reference's class (diffusion_architecture.py:64), so the tools' plumbing works on it as
    written: `mmcv.runner.load_checkpoint(model, path)` (walks `_load_from_state_dict`), `.cuda()` / `.eval()`,
    `MMDataParallel(model, device_ids=[0])(**data)`, `model.state_dict()`.  It owns no nn.Parameters: the weights live
    packed in HBM (DenoiserWeights / GestureRepEncoder) and `state_dict()` hands back the tensors it was loaded from.

    Options beyond the reference's keys (constructor arguments, not environment variables):
      precision   "bf16" (production) | "fp32" (bf16x3 operands, parity checks)
      lanes       concurrent clip groups per forward, each on its own hardware queue (measured on MI355X: 2 lanes
                  149.7 vs 156 ms guided B=16, 71.3 vs 73.0 ms base B=32; 3-4 lanes no better)
      async_results     False: forward() returns tensors that are valid on the caller's stream (the reference's semantics).
                        True: forward() returns as soon as the batch is queued; `results["done_event"]` marks its completion
                        (wait for it on the consuming stream, `wait_results(results)` does so on the current one; or queue
                        the consumer on `results["done_stream"]`, which needs no wait).  The
                        front end of the next batch (conditions, VAE encodes, retrieval: ~13 ms of a 131 ms guided step)
                        and the decode of this one then run beside the dependent inversion -> sampling chain instead of in
                        front of / behind it.  Batches alternate between `slots` sets of sessions and graph buffers so that
                        a front end never writes what the chain in flight still reads; at most `max_inflight` batches
                        are queued before forward() blocks on the oldest.
      batch_lanes       submit() of batches WITH exemplar inversion (the co-batched pipeline): whole batches rotate over this
                        many lanes, each running [sampling of its pending batch || inversion of the new one] as one chain of 50
                        launches; the launch form follows from the lanes (_seq_form_auto).  Default 4: 4 lanes x 64 workgroups
                        of rg_seq2_kernel (two sequences of a kind per workgroup; the 32 classifier-free workgroups leave after
                        1.05 of the launch's 1.66 ms and the front end of the next batch runs on their compute units).  8 lanes
                        x 32 workgroups (the classifier-free pair behind the conditional one in the same workgroup, 2.6 ms per
                        launch) hold every compute unit all the time: 2 % more batches per second in a long run (33.1 vs 33.9
                        ms per guided step of 16 clips over 40 steps), but twice the batches in flight, twice the latency per
                        batch and a longer fill and drain (36.4 vs 35.5 ms over 20 steps; profiles/r05w_lanes_forms.txt)
      lane_streams      the caller's own streams for the lanes / the search / the base and batch lanes
                        (max(lanes, base_lanes, batch_lanes) + 2 of them)
      calibrate_lanes   (default) pick streams that were measured to run concurrently (distinct hardware queues) first --
                        two lanes that share a hardware queue run their chains one after the other (61 instead of 41 ms per
                        guided step, profiles/r04i_lane_timeline.txt); performance only: the NUMBER of lanes and the presence of
                        the search stream never depend on a measurement.  False: the next streams of torch's pool, unmeasured
      session_options   keyword arguments of denoiser.DenoiserSession (engine, ln_mode, ...)
      vae_options       keyword arguments of vae.GestureRepEncoder (part_streams, chain)"""

    def __init__(self, model=None, loss_recon=None, loss_gen=None, loss_contact=None, loss_laplace=None,
                 diffusion_train=None, diffusion_test=None, init_cfg=None, inference_type="ddpm",
                 genloss_acceleration_weight=True, genloss_hands_weight=2, genloss_smooth=True,
                 body_part_lossweights=None, device="cuda", precision="bf16", lanes=2, sample_lanes=None, session_options=None,
                 vae_options=None, async_results=False, slots=2, max_inflight=2, cobatch_lanes="batch", base_lanes=8,
                 batch_lanes=4, lane_streams=None, calibrate_lanes=True, decode_stream=False, dynamic_forms=False, dynamic_budget=None, invert_alone_wide=True, tail_glue=True, **kwargs):
        super().__init__()
        # loss_* / diffusion_train / body_part_lossweights are training-only keys: accepted, unused
        self.model = build_submodule(model, device=device, **kwargs)
        dt = dict(diffusion_test)
        self.schedule = sched_mod.Schedule(beta_scheduler=dt["beta_scheduler"], diffusion_steps=dt["diffusion_steps"],
                                           respace=dt.get("respace"))
        capi.require(dt.get("model_mean_type", "start_x") == "start_x" and dt.get("classifier_free_guidance_scale", 0) == 0,
                "unsupported argument: requires dt.get(\"model_mean_type\", \"start_x\") == \"start_x\" and dt.get(\"classifier_free_guidance_scale\", 0) == 0")
        self.inference_type = inference_type
        capi.require(inference_type in ("ddim", "ddpm"), "inference_type is 'ddim' (shipped config) or 'ddpm'")
        self.device, self.precision = torch.device(device), precision
        self.training = False
        self.session_options, self.vae_options = dict(session_options or {}), dict(vae_options or {})
        self._state = None
        self._sessions = {}
        self._graphs = {}
        self.use_graphs = True  # capture the fixed launch sequences (loops, VAEs) into HIP graphs
        self.graph_captures = self.graph_cross_stream_waits = 0     # counters (tests, bench line)
        self._jitter = None     # test hook: callable(stream, tag) run before a graph use / a tail is queued on `stream`
        self.profile_phases, self.phase_ms = False, {}
        self.lanes = int(lanes)
        # submit() of batches WITHOUT exemplar inversion (base diffusion: 2 B sequences per launch, 64 at B = 32) lets whole
        # batches alternate between this many lanes -- a launch holds one CU per sequence, so three or four such chains fit the chip (3 measured as good as 4: profiles/r03g)
        self.base_lanes = max(1, int(base_lanes))
        self.batch_lanes = max(1, int(batch_lanes))
        self.sample_lanes = None if sample_lanes is None else int(sample_lanes)
        # asynchronous submission (see forward): off = the reference's semantics (results valid on the caller's stream)
        self.async_results, self.slots, self.max_inflight = bool(async_results), max(1, int(slots)), max(1, int(max_inflight))
        # dynamic_forms: the denoiser launches of submit()'s chains choose their form ON THE DEVICE, launch by launch, from the
        # workgroups the other lanes hold (include/rg_gesture.h: rg_lane_form / rg_seqx_forward): two sequences per workgroup
        # while the chip is full, one per workgroup (0.6 of the time per launch) while the pipeline fills or drains.  Same bits.
        # OFF by default: measured on the headline run (profiles/r06s_ab.txt, r06m_ab.txt, r06n_ab.txt: 31.3-31.7 against
        # 30.8-31.1 ms per step, median batch latency 266-270 against 279-286 ms) it buys latency, not throughput -- with the
        # chains running wide the caller's stream (clip encode -> retrieval -> exemplar encode -> condition projections, in
        # series, beside the lanes' launches) paces the pipeline instead of the chains (profiles/r06k_timed_region.txt).
        self.dynamic_forms, self._lane_state = bool(dynamic_forms), None
        self.invert_alone_wide = bool(invert_alone_wide)      # an inversion ALONE (a filling pipeline) as one workgroup per sequence
        self.tail_glue = bool(tail_glue)      # co-batched chains: a loop step's update at the end of its forward (sampler.cobatched_loop)
        self.dynamic_budget = None if dynamic_budget is None else int(dynamic_budget)     # workgroups the lanes may hold together (None: the chip's compute units)
        self.rotation_lanes = None     # submit(): lanes in the rotation of whole batches (None: batch_lanes / base_lanes); longform.py pins 1
        self.form_lanes = None         # ... and how many of their chains really run side by side (None: all of them); longform.py: 2
        self._slot, self._inflight, self._graph_owner, self._slot_done = 0, collections.deque(), {}, {}
        # co-batched pipeline (submit / flush): the batch whose exemplars are inverted and whose sampling is still to come
        # (one pipeline per lane when whole batches alternate between the lanes, cobatch_lanes="batch"; else one, key None)
        self._pend, self._cob, self._ready, self._tail_turn = {}, None, collections.deque(), 0
        self._slots, self._submitted, self._last_out_seq = {}, 0, -1
        self.cobatch_lanes = str(cobatch_lanes)
        capi.require(self.cobatch_lanes in ("batch", "split"),
                "unsupported argument: requires self.cobatch_lanes in (\"batch\", \"split\")")
        self._lane_streams, self._search_stream, self._decode_stream, self._lanes_calibrated = [], None, None, None
        # decode_stream=True: asynchronous batches decode on a stream of their own instead of the lane's.  On the lane's stream the
        # decode (4.9 ms alone, 8-15 ms beside the other lane's chain) sits between two chains of that lane
        # (profiles/r04i_lane_timeline.txt); on its own stream it competes with BOTH chains for compute units and the step gets
        # slower (43.1 vs 41.0 ms, profiles/r04j): off by default, as round 3 found for the same experiment
        self.decode_on_own_stream = bool(decode_stream)
        self._given_streams = None if lane_streams is None else list(lane_streams)
        self.calibrate_lanes, self.lane_report = bool(calibrate_lanes), None

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state, strict=True):
        """Accepts the reference's key names: `model.` prefixed (mmcv checkpoint of MotionDiffusion,
        tools/visualize.py:141, bare or wrapped in {"state_dict": ...}) or un-prefixed ReGestureTransformer keys.
        Body-part VAE weights missing from `state` are taken from the VAE checkpoints named by the YAMLs
        (diffusion_transformer.py:151-188).  Returns (missing_keys, unexpected_keys) like nn.Module."""
        if "state_dict" in state and not torch.is_tensor(state["state_dict"]):
            state = state["state_dict"]
        if any(k.startswith("model.") for k in state):
            state = {k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}
        else:
            state = dict(state)
        m = self.model
        for part in vae_mod.PARTS:
            pre = "gesture_rep_encoder.%s_vae." % part
            sd = (getattr(m, "vae_states", None) or {}).get(part)
            if sd is not None and not any(k.startswith(pre) for k in state):
                state.update({pre + k: v for k, v in sd.items()})
        m.weights = denoiser.DenoiserWeights(state, m.cfg, self.schedule, self.device, precision=self.precision)
        m.gesture_rep_encoder = vae_mod.GestureRepEncoder(state, m.vae_cfgs, self.device, self.precision, **self.vae_options)
        m.gesture_rep_encoder.graph_runner = self._graph_run
        self._state = state
        self._sessions = {}
        self._graphs = {}
        return IncompatibleKeys([], [])

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """The hook `mmcv.runner.load_checkpoint` (mmcv/runner/checkpoint.py `load_state_dict`) and
        `nn.Module.load_state_dict` walk: called once on this module with the whole checkpoint dict."""
        sd = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        try:
            self.load_state_dict(sd, strict=strict)
        except KeyError as e:
            missing_keys.append(prefix + str(e).strip("'"))

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        """The reference-format tensors this model was loaded from (`model.`-prefixed like MotionDiffusion's own)."""
        out = collections.OrderedDict() if destination is None else destination
        for k, v in (self._state or {}).items():
            out[prefix + "model." + k] = v
        return out

    _RECORD = True

    @staticmethod
    def _used_on(stream, *tensors):
        """The caching allocator must not hand a tensor's memory out again before `stream` (not the stream it was
        allocated on) is done with it: every tensor that crosses streams is marked, so that freeing it on the host while a
        lane is still behind (asynchronous submission: by a whole batch) is safe."""
        if not MotionDiffusion._RECORD:
            return
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)

    def _graph_run(self, key, inputs, fn, owner=None):
        """Run fn(static_inputs) -> outputs through a cached HIP graph: `inputs` (dict of device
        tensors or None) are copied into static buffers, the captured launch sequence is replayed
        and clones of the outputs are returned.  Falls back to eager launches if use_graphs is off.
        owner: key of the session whose buffers the graph is bound to (evicted together).

        A graph owns ONE set of static inputs, intermediates and outputs, so its uses must not overlap: each use
        (copy-in, replay, clone-out) ends with an event, and a use on another stream than the previous one waits for it
        first.  Uses on the same stream are ordered by the stream.  (Keys that name a lane and a slot are only ever
        replayed on that lane's stream; the chain is what makes every other key -- VAE encode / decode -- safe whatever
        stream the caller or the tail of a batch happens to run on.)"""
        cur = torch.cuda.current_stream()
        self._used_on(cur, *inputs.values())
        if not self.use_graphs:
            return fn(inputs)
        if owner is not None:      # the graph captures the owner session's launches: one graph per launch form (_session)
            key = key + self._form_tag(self._session_opts(owner[0], owner[1], owner[2]))
        ent = self._graphs.pop(key, None)
        if ent is not None:
            self._graphs[key] = ent                               # most recently used goes last
        if ent is None:
            torch.cuda.synchronize()  # capture from a quiet device; an evicted graph must not be in flight when it dies
            while len(self._graphs) >= self.MAX_GRAPHS:
                old = next(iter(self._graphs))
                del self._graphs[old]
                self._graph_owner.pop(old, None)
            if owner is not None:
                self._graph_owner[key] = owner
            static = {k: (None if v is None else torch.empty(v.shape, dtype=v.dtype, device=v.device).copy_(v))
                      for k, v in inputs.items()}
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):  # warm-up outside capture (lazy module loads, allocator)
                fn(static)
            cur.wait_stream(side)
            for k, v in inputs.items():
                if v is not None:
                    static[k].copy_(v)
            graph = torch.cuda.CUDAGraph()
            with capi.capture(graph):
                outs = fn(static)
            ent = self._graphs[key] = (graph, static, outs, [None, None])
            self.graph_captures += 1
        graph, static, outs, last = ent
        if last[0] is not None and last[1] != cur.cuda_stream:
            cur.wait_event(last[0])                               # the previous use ran on another stream
            self.graph_cross_stream_waits += 1
        if self._jitter is not None:
            self._jitter(cur, key)
        for k, v in inputs.items():
            if v is not None:
                static[k].copy_(v)
        graph.replay()
        res = tuple(o.clone() for o in outs)
        last[0], last[1] = cur.record_event(), cur.cuda_stream
        return res

    @staticmethod
    def wait_results(results, stream=None):
        """Asynchronous submission: make `stream` (default: the current one) wait for the batch behind `results` and mark
        its tensors as used there; a no-op for synchronous results.  Returns `results`."""
        ev = results.get("done_event") if isinstance(results, dict) else None
        if ev is not None:
            stream = torch.cuda.current_stream() if stream is None else stream
            stream.wait_event(ev)
            MotionDiffusion._used_on(stream, *[v for v in results.values() if torch.is_tensor(v)])
        return results

    def train(self, mode=True):
        if mode:
            raise capi.RgError("this is the inference hot path: training is out of scope")
        self.training = False
        return self

    @contextlib.contextmanager
    def _phase(self, name):
        """Wall time of one phase of forward() into self.phase_ms (only when self.profile_phases is
        set: it synchronises the device at phase boundaries, which the production path never does)."""
        if not getattr(self, "profile_phases", False):
            yield
            return
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        yield
        torch.cuda.synchronize()
        self.phase_ms[name] = self.phase_ms.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

    def cuda(self, *a, **k):
        return self

    def cpu(self):
        return self   # weights stay packed in HBM; forward() raises without a GPU (no CPU fallback)

    def to(self, *a, **k):
        return self

    MAX_SESSIONS, MAX_GRAPHS = 192, 768  # LRU caps (a session holds ~45 MB of activations per 16 clips, a graph its statics;
    #                                       ten batch lanes x two slots x three roles must fit: an evicted session re-captures its graphs)

    def _session(self, B, role="sample", lane=0):
        # The launch form the CURRENT rotation resolves to is part of the key (and of the keys of the graphs that capture the
        # session's launches, _graph_run): a long-form run that pins two lanes after a four-lane run, base and guided batches
        # of one size on one lane, a synchronous forward() between submit()s each get the session of their own form instead
        # of whichever was built first.  Sessions are never dropped while their launches may be in flight.
        opts = self._session_opts(B, role, lane)
        key = (B, role, lane, self._slot) + self._form_tag(opts)
        if key not in self._sessions:
            while len(self._sessions) >= self.MAX_SESSIONS:      # evict the least recently used session and its graphs
                old = next(iter(self._sessions))
                torch.cuda.synchronize()                         # (nothing of it is in flight when its buffers go)
                del self._sessions[old]
                for gk in [g for g, o in self._graph_owner.items() if o == old[:4]]:
                    self._graphs.pop(gk, None)
                    del self._graph_owner[gk]
            self._sessions[key] = denoiser.DenoiserSession(self.model.weights, B, **opts)
        else:
            self._sessions[key] = self._sessions.pop(key)        # most recently used goes last
        return self._sessions[key]

    LANE_SLOTS = 32          # lanes the shared arbitration state has room for

    @staticmethod
    def _form_tag(opts):
        return (bool(opts.get("seq_pairs")), opts.get("seq_duo"), "lane_dyn" in opts)

    def _session_opts(self, B, role, lane):
        """Constructor options of the session (B, role, lane): the launch form resolved from the CURRENT rotation."""
        opts = dict(self.session_options)
        opts.setdefault("tail_glue", self.tail_glue)
        if opts.get("seq_pairs", "auto") == "auto":
            opts["seq_pairs"], duo = self._seq_form_auto(B)
            if opts.get("seq_duo") is None:
                opts["seq_duo"] = duo
            if opts["seq_pairs"] and opts["seq_duo"] and role == "invert":
                # an inversion ALONE runs only while the pipeline fills (in the steady state it shares the launches of a
                # pending batch's sampling): the chip is emptying or empty then, so the classifier-free pairs get workgroups
                # of their own (B workgroups for 1.6 ms instead of B / 2 for 2.6 ms per launch; same bits)
                opts["seq_pairs"], opts["seq_duo"] = False, True
            if role == "invert" and self.invert_alone_wide and self._cob is not None:
                opts["seq_pairs"], opts["seq_duo"] = False, False
        cob = self._cob
        seq = opts.get("engine") != "chain" and getattr(self.model.weights, "seq_streams", None) is not None
        if (self.dynamic_forms and seq and self.async_results and cob is not None and cob.get("lane") is not None
                and 0 <= lane < self.LANE_SLOTS and "lane_dyn" not in opts):
            if self._lane_state is None:
                self._lane_state = torch.zeros(self.LANE_SLOTS, seqfwd.LANE_STRIDE, device=self.device, dtype=torch.int32)
            cus = self.dynamic_budget or torch.cuda.get_device_properties(self.device).multi_processor_count
            opts["lane_dyn"] = (self._lane_state, lane, self.LANE_SLOTS, cus)
        return opts

    def _seq_form_auto(self, B, cus=None):
        """(seq_pairs, seq_duo) of a session of B clips: the WIDEST launch form that still fits the chip beside the other lanes'
        launches -- 2 B workgroups (one per sequence, 1.0 ms per launch), B (rg_seq2: two sequences of a kind per workgroup,
        1.66 ms; the classifier-free half of the workgroups leaves after 1.05), or B / 2 (rg_seq2, the classifier-free pair
        behind the conditional one in the same workgroup, 2.6 ms) -- counted over the lanes that whole batches currently rotate
        over (batch_lanes; base_lanes for batches without inversion; what longform.py pins).  All forms give the same bits."""
        cob = self._cob
        if not (self.async_results and cob is not None and cob.get("lane") is not None):
            return False, False
        if cus is None:
            cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        rot = getattr(self, "_cur_rot", None) or self.batch_lanes
        if 2 * B * rot <= cus:
            return False, False
        if B * rot <= cus:
            return False, True
        return True, True

    def _set_conditions(self, B, role, lane, word, audio, speaker_ids, motion_mask, query_masks):
        """DenoiserSession.set_conditions through the graph cache: its ~55 launches (pre-projections, per-layer K/V
        GEMMs and reductions) cost the host ~1.2 ms when issued one by one, and the exemplars' call sits between
        their VAE encode and the inversion loop, where the device would wait for it."""
        sess, dev = self._session(B, role, lane), self.model.weights.dev
        ins = dict(word=word.to(dev).float(), audio=audio.to(dev).float(), spk=speaker_ids.to(dev).long(),
                   mask=motion_mask.to(dev).float())
        for c in denoiser.CONDS:
            ins["q_" + c] = query_masks[c].to(dev).float()

        def run(st):
            sess.set_conditions(st["word"], st["audio"], st["spk"], st["mask"], {c: st["q_" + c] for c in denoiser.CONDS})
            return ()

        self._graph_run(("cond", B, role, lane, self._slot, tuple(ins["word"].shape), tuple(ins["audio"].shape),
                         tuple(ins["spk"].shape)), ins, run, owner=(B, role, lane, self._slot))
        return sess

    def _concurrent_streams(self, n):
        """Up to n HIP streams that were MEASURED to run concurrently (calibrate_lanes=True only; a performance
        refinement, never a correctness condition).  ROCm multiplexes streams onto a few hardware queues and two
        streams on the same queue serialise -- observed for the first two side streams of a process.  Candidates are
        checked pairwise with a short spin kernel on each (concurrent: ~1x the spin time, same queue: ~2x) and a mutually
        concurrent set is kept; the caller pads it to the count it needs."""
        spin = 2_000_000  # cycles (~1 ms): long against launch latency, short against anything else

        def together(a, b):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for st in (a, b):
                with torch.cuda.stream(st):
                    torch.cuda._sleep(spin)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        def alone(a):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(a):
                torch.cuda._sleep(spin)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        # (torch hands out streams from a pool of 32 per priority and wraps around: more candidates than that would alias
        #  earlier ones as distinct Python objects -- ADVICE r04 -- so the list is capped and de-duplicated by HIP stream)
        cands, seen = [], {torch.cuda.current_stream().cuda_stream}
        for _ in range(min(4 * n + 4, 32)):
            c = torch.cuda.Stream(device=self.device)
            if c.cuda_stream not in seen:
                seen.add(c.cuda_stream)
                cands.append(c)
        alone(cands[0])  # warm-up
        base = min(alone(cands[0]) for _ in range(3))
        fixed = [torch.cuda.current_stream()]   # the caller's stream keeps working beside the lanes
        chosen = []
        for c in cands:
            if len(chosen) == n:
                break
            if all(min(together(c, o) for _ in range(2)) < 1.5 * base for o in fixed + chosen):
                chosen.append(c)
        rest = [c for c in cands if all(c.cuda_stream != o.cuda_stream for o in chosen)]
        return chosen, rest

    def _make_streams(self):
        """The stream topology: `lanes` lane streams, one search stream, then the additional lanes of the base workload --
        max(lanes, base_lanes, batch_lanes) + 2 streams, ALWAYS that many, whatever the machine, the load on the host or the number of
        ranks starting at once (the schedule a test pins is the schedule every box runs).  Sources, in order: the
        `lane_streams=` constructor argument (the caller's own streams); `calibrate_lanes=True`: streams measured to be
        concurrent first, unmeasured ones to make up the count; otherwise fresh streams."""
        lanes = max(1, int(self.lanes))
        need = max(lanes, self.base_lanes, self.batch_lanes) + 2     # lanes (+ base / batch lanes), the search stream, the decode stream
        report = dict(lanes=lanes, base_lanes=self.base_lanes, batch_lanes=self.batch_lanes, streams=need, source="fresh",
                      measured_concurrent=None)
        if self._given_streams is not None:
            found = list(self._given_streams)
            capi.require(len(found) >= need, "lane_streams: %d streams needed (max(lanes, base_lanes, batch_lanes) + the search stream + the decode stream), got %d"
                         % (need, len(found)))
            report["source"] = "caller"
        elif self.calibrate_lanes:
            chosen, rest = self._concurrent_streams(need)
            found = (chosen + rest)[:need]
            report.update(source="calibrated", measured_concurrent=len(chosen))
        else:
            found = [torch.cuda.Stream(device=self.device) for _ in range(need)]
        found = found[:need]
        self._search_stream, self._decode_stream = found[lanes], found[lanes + 1]
        self._lane_streams = found[:lanes] + found[lanes + 2:]
        self._lanes_calibrated = (lanes, self.base_lanes, self.batch_lanes)
        self.lane_report = report

    def stream_set(self):
        """The streams of this model in the order the `lane_streams=` constructor argument takes them: several models of one
        process (bench.py builds six) can share one measured set instead of each measuring its own."""
        lanes = max(1, int(self.lanes))
        if self._lanes_calibrated != (lanes, self.base_lanes, self.batch_lanes):
            self._make_streams()
        return self._lane_streams[:lanes] + [self._search_stream, self._decode_stream] + self._lane_streams[lanes:]

    def _lane_plan(self, B, n_lanes=None):
        """[(lane index, stream, b0, b1)]: contiguous, near-equal groups of clips (n_lanes of them, default self.lanes)."""
        lanes = max(1, int(self.lanes))
        if self._lanes_calibrated != (lanes, self.base_lanes, self.batch_lanes):
            self._make_streams()
        want = max(lanes, max(self.base_lanes, self.batch_lanes) if self.async_results else 1)
        n = max(1, min(lanes if n_lanes is None else int(n_lanes), want, B, len(self._lane_streams)))
        cuts = [(B * i) // n for i in range(n + 1)]
        return [(i, self._lane_streams[i], cuts[i], cuts[i + 1]) for i in range(n)]

    # ------------------------------------------------------------------ forward (eval)
    def forward(self, **kwargs):
        if self.model.weights is None:
            raise capi.RgError("weights not loaded: call load_state_dict() first")
        dev = self.device
        W, S = self.model.weights, self.schedule.num_timesteps
        h = W.h
        inference_kwargs = kwargs.get("inference_kwargs", {})
        use_outpaint = inference_kwargs.pop("outpaint", False)
        use_inversion = inference_kwargs.pop("use_inversion", False)
        inversion_start_time = inference_kwargs.pop("inversion_start_time", -1)
        visualize_inversion = inference_kwargs.pop("visualize_inversion", False)
        use_insertion_guidance = inference_kwargs.pop("insertion_guidance", False)
        guidance_iters = inference_kwargs.pop("guidance_iters", [10] * 50)
        if isinstance(guidance_iters, str):     # the tools' --guidance_iters names
            guidance_iters = guidance_iters_preset(guidance_iters)
        guidance_lr = inference_kwargs.pop("guidance_lr", 0.1)
        use_prev_latent = inference_kwargs.pop("use_prev_latent", False)
        prev_latent = inference_kwargs.pop("prev_latent", None)
        tape = inference_kwargs.pop("noise_tape", None) or _TorchNoise(dev)
        if use_prev_latent:
            capi.require(not use_outpaint, "use_prev_latent excludes use_outpaint (diffusion_architecture.py:227-241)")
        if use_outpaint:
            capi.require(not use_inversion, "use_outpaint excludes use_inversion")
            capi.require(not use_insertion_guidance, "use_outpaint excludes insertion_guidance")
        if use_insertion_guidance:
            capi.require(not use_outpaint, "insertion_guidance excludes use_outpaint")
            capi.require(use_inversion, "insertion_guidance needs use_inversion")

        gre = self.model.gesture_rep_encoder
        ev_inputs = torch.cuda.Event()
        ev_inputs.record(torch.cuda.current_stream())   # everything the caller queued for the inputs
        B = kwargs["motion_upper"].shape[0]
        D = gre.vae_latent_dim
        eps_list = [tape.draw((B * 10, 1, D)) for _ in range(4)]
        f = lambda t: t.to(dev).float().contiguous()
        motion_mask = gre.latent_mask(kwargs["motion_mask"].float())
        T = motion_mask.shape[1]
        n_lat = (T - 3) // 4
        up_i, ha_i, fa_i, lt_i = (list(range(0, n_lat)), list(range(n_lat + 1, 2 * n_lat + 1)),
                                  list(range(2 * n_lat + 2, 3 * n_lat + 2)), list(range(3 * n_lat + 3, T)))
        # cross-attention query masks: the reference's index arithmetic gives rows 10, 20, 30
        # ([(T-3)//4, 2*(T-3)//4, 3*(T-3)//4], diffusion_architecture.py:155), not the separators.
        qmask = torch.ones_like(motion_mask)
        for r in ((T - 3) // 4, 2 * (T - 3) // 4, 3 * (T - 3) // 4):
            qmask[:, r] = 0      # (a LIST index would go through a synchronous host-to-device copy of the index tensor,
            #                       which on ROCm waits for every stream of the device: the host then cannot queue ahead)
        query_masks = {c: qmask for c in denoiser.CONDS}

        # the conditioning projections (K/V of every layer) need the inputs only: they run on the lane streams
        # while the main stream encodes the motion
        plan = self._lane_plan(B)                            # exemplar inversion
        plan_s = self._lane_plan(B, self.sample_lanes)       # sampling loops
        main = torch.cuda.current_stream()
        # asynchronous submission: this batch's sessions / graph buffers are those of its slot; its front end (everything
        # up to the lanes) stays on the caller's stream and the search stream, because the lane streams may still be
        # busy with the previous batch's chain
        run_async = self.async_results and self.slots > 1 and not getattr(self, "profile_phases", False)
        cob = self._cob if run_async else None      # submit(): this batch's sampling is deferred to a later call
        pid = None
        if cob is not None and cob.get("lane") is not None:
            # whole batches rotate over the lanes: this one runs (inversion now, sampling one rotation later) on one
            # lane's stream while the batches submitted before it are busy on the others
            # (base: nothing to share launches with, more and smaller chains; with inversion: the co-batched chains)
            # (the lane is a function of the submission count and the lane count ONLY: a batch of fewer clips than lanes must
            #  not change the modulus, or it lands on a lane whose pending batch is not the oldest)
            self._lane_plan(B)                                   # (makes the streams on first use)
            n_rot = getattr(self, "rotation_lanes", None) or (self.batch_lanes if use_inversion else self.base_lanes)
            n_rot = max(1, min(int(n_rot), len(self._lane_streams)))
            self._cur_rot = getattr(self, "form_lanes", None) or n_rot      # chains that really run side by side (launch forms)
            pid = cob["lane"] % n_rot
            plan = plan_s = [(pid, self._lane_streams[pid], 0, B)]
        if run_async:
            self._slot = self._take_slot(pid, main, self._lanes_of(plan, plan_s))
        else:
            self._slot = 0
            self._wait_slot(main, self._lanes_of(plan, plan_s), 0)   # (asynchronous batches of an earlier mode still in flight)
        gre.concurrent_parts(not run_async)   # one launch chain per VAE graph when graphs are queued behind a running batch
        word, audio, spk = kwargs["word"], kwargs["audio"], kwargs["speaker_ids"]
        with self._phase("conditions"):
            for lane, stream, b0, b1 in ([] if cob is not None else plan_s):
                cstream = main if run_async else stream
                cstream.wait_stream(main)
                with torch.cuda.stream(cstream):
                    self._set_conditions(b1 - b0, "sample", lane, word[b0:b1], audio[b0:b1], spk[b0:b1],
                                         motion_mask[b0:b1], {c: qmask[b0:b1] for c in denoiser.CONDS})
        with self._phase("vae_encode"):
            motion, tr_rel = gre.encode_device_graphed(
                f(kwargs["motion_upper"]), f(kwargs["motion_lower"]), f(kwargs["motion_face"]), f(kwargs["motion_hands"]),
                f(kwargs["trans"]), f(kwargs["facial"]), f(kwargs["contact"]), [f(e) for e in eps_list])
        kwargs["trans"].copy_(tr_rel.to(kwargs["trans"].device))  # the reference's in-place re-zeroing
        capi.require(motion.shape[1] == T, "unsupported argument: requires motion.shape[1] == T")
        kwargs.update({"motion_mask": motion_mask, "text": kwargs["word"], "raw_text": kwargs.get("raw_word"),
                       "text_times": kwargs.get("text_segments")})
        retrieval_dict = kwargs.get("re_dict")
        early_cond = {}   # lane -> number of exemplars whose conditions were already projected

        def exemplar_conditions_early(ex, recs, fork):
            """Called by RetrievalDatabase.forward once the exemplars are known, before it VAE-encodes them: their
            K/V projections (text / audio / speaker of the retrieved samples) go to the lane streams meanwhile."""
            if not use_inversion or getattr(self, "profile_phases", False):
                return
            if cob is not None and self._pend.get(pid) is not None:
                return      # co-batched with the pending batch's sampling: the conditions go into the shared sessions
            for lane, stream, b0, b1 in plan:
                sel = [e for e, (b, _, _, placed) in enumerate(ex) if placed is not None and b0 <= b < b1]
                if not sel:
                    continue
                cstream = main if run_async else stream
                cstream.wait_event(fork)   # main's state before the exemplar VAE encode was queued
                with torch.cuda.stream(cstream):
                    Ep = bucket(len(sel))
                    st = lambda k: pad_rows(torch.stack([recs[e][k] for e in sel]).to(dev), Ep)
                    eqm = {c: pad_rows(torch.stack([qmask[ex[e][0]] for e in sel]), Ep) for c in denoiser.CONDS}
                    self._set_conditions(Ep, "invert", lane, st("word"), st("audio"), st("speaker_id"),
                                         gre.latent_mask(st("motion_mask").float()), eqm)
                early_cond[lane] = len(sel)

        if retrieval_dict is None and self.model.database is not None:
            with self._phase("retrieval"):
                retrieval_dict = self.model.database(kwargs, kwargs.get("motion_length"), dev, idx=kwargs.get("sample_name"),
                                                     retrieval_method=kwargs.get("retrieval_method", "discourse"),
                                                     gesture_rep_encoder=gre, noise=tape,
                                                     on_exemplars=exemplar_conditions_early,
                                                     search_stream=None if getattr(self, "profile_phases", False)
                                                     else self._search_stream, inputs_ready=ev_inputs)
        results = kwargs
        results["retrieval_dict"] = copy.copy(retrieval_dict)

        if use_outpaint:
            rml = retrieval_dict["raw_motion_latents"]
            capi.require(rml.shape[1] == 1, "unsupported argument: requires rml.shape[1] == 1")
            retrieval_motion_latents = rml.squeeze(1).to(dev).float().contiguous()
        prev_future = None
        if use_prev_latent and isinstance(prev_latent, PendingLatent):
            capi.require(len(prev_latent.rows) == B if prev_latent.rows is not None else prev_latent.state.B == B,
                         "prev_latent: the pending latent must hold one row per clip of this batch")
            prev_future, prev_latent = prev_latent, torch.zeros(B, T, D, device=dev)    # filled by _fill_prev
        elif use_prev_latent and prev_latent is not None:
            prev_latent = prev_latent.to(dev).float()
            masked = torch.zeros_like(prev_latent)
            for idx in (up_i, ha_i, fa_i, lt_i):
                h.call("copy_rows", prev_latent, masked, B, 1, D, T, idx[-1], T, idx[0])
            prev_latent = masked

        # ---- every random draw happens here, for the whole batch, in the reference's order
        start_noise, invl = None, None
        if use_inversion:
            start_noise = tape.draw((B, T, D)).to(dev).contiguous()
            if use_insertion_guidance:
                invl = torch.zeros(S, B, T, D, device=dev)
            x = start_noise
        else:
            x = tape.draw((B, T, D)).to(dev).contiguous()
        if use_inversion and visualize_inversion and not isinstance(tape, _TorchNoise):
            # the reconstruction check of every exemplar (diffusion_architecture.py:366-379) is a 50-step DDIM loop that
            # draws randn_like(x) per step (sigma = 0: never used): keep an explicit tape aligned
            for b in range(B):
                for _ in retrieval_dict["retr_uncropped_latents"][b]:
                    for _ in range(S):
                        tape.draw((1, T, D))
        in_seq = None
        if use_prev_latent and prev_latent is not None:
            in_seq = prev_latent
        elif use_outpaint:
            in_seq = retrieval_motion_latents
        ddpm = self.inference_type == "ddpm"
        ddpm_noise = None
        if ddpm:
            # diffusion_architecture.py:424-432: p_sample_loop with a FRESH randn start (the inversion / in_seq results
            # above are computed and ignored by this branch, as in the reference); one randn_like(x) per step
            if use_inversion:   # start_noise was drawn (and spliced) for nothing; p_sample_loop draws its own start
                x = tape.draw((B, T, D)).to(dev).contiguous()
            if isinstance(tape, _TorchNoise):
                ddpm_noise = tape.draw((S, B, T, D))
            else:
                ddpm_noise = torch.empty(S, B, T, D, device=dev)
                for i in range(S - 1, -1, -1):
                    ddpm_noise[i].copy_(tape.draw((B, T, D)).to(dev))
            in_seq = None
        need_noise = (in_seq is not None or use_insertion_guidance) and not ddpm
        inseq_noise = None
        if need_noise:
            # the reference draws randn_like(in_seq) then randn_like(x) on every step (the latter is
            # multiplied by sigma = 0); keep the tape aligned
            first = S - 1
            cur_has = in_seq is not None
            if isinstance(tape, _TorchNoise):
                # generator noise: which draw lands on which step is immaterial -> one launch for all steps
                inseq_noise = tape.draw((S, B, T, D))
            else:
                inseq_noise = torch.empty(S, B, T, D, device=dev)
                for i in range(S - 1, -1, -1):
                    has = cur_has if i == first or not use_insertion_guidance else True
                    if has:
                        inseq_noise[i].copy_(tape.draw((B, T, D)).to(dev))
                    tape.draw((B, T, D))
        elif not isinstance(tape, _TorchNoise) and not ddpm:
            for _ in range(S):
                tape.draw((B, T, D))
        # everything the later phases need, so that the sampling and the tail of this batch can also run in a LATER call
        # (submit / flush: its sampling then shares launches with the next batch's inversion)
        st = types.SimpleNamespace(
            B=B, T=T, D=D, S=S, n_lat=n_lat, plan=plan, plan_s=plan_s, main=main, run_async=run_async, results=results,
            retrieval_dict=retrieval_dict, x=x, start_noise=start_noise, invl=invl, in_seq=in_seq, inseq_noise=inseq_noise,
            ddpm=ddpm, ddpm_noise=ddpm_noise, x_out=torch.empty(B, T, D, device=dev), use_inversion=use_inversion,
            use_insertion_guidance=use_insertion_guidance, guidance_iters=guidance_iters, guidance_lr=guidance_lr,
            visualize_inversion=visualize_inversion, inversion_start_time=inversion_start_time, vis_inv=[], vis_pairs=[],
            word=word, audio=audio, spk=spk, motion_mask=motion_mask, qmask=qmask, early_cond=early_cond,
            use_prev_latent=use_prev_latent, prev_latent=prev_latent, idx_groups=(up_i, ha_i, fa_i, lt_i), slot=self._slot, pid=pid, seq=self._submitted, via_submit=cob is not None,
            prev_future=prev_future if in_seq is not None else None, latent_ready=None)
        # what a later batch's pending prev_latent needs of this one (not the whole state: invl alone is 140 MB at B = 32)
        st.handle = self._last_state = types.SimpleNamespace(B=B, x_out=st.x_out, latent_ready=None)
        if cob is not None:
            # the conditions of this batch's clips are projected in the NEXT call (into the sessions it shares with that
            # batch's exemplars): private copies, the caller may reuse its input buffers meanwhile
            st.word, st.audio, st.spk = (t.to(dev).clone() for t in (word, audio, spk))
            return self._submit_chain(st)
        self._inversion_pass(st)
        self._sampling_pass(st)
        return self._tail(st)

    def pending_latent(self):
        """PendingLatent of the batch passed to the most recent forward() / submit()."""
        if getattr(self, "_last_state", None) is None:
            raise capi.RgError("pending_latent(): no batch has been submitted")
        return PendingLatent(self._last_state)

    def _fill_prev(self, st, stream):
        """Bind a pending prev_latent: on `stream`, once the producing batch's sampling is done, the first-token rows of its
        final latent go into this batch's in_seq (what forward() does at once for a tensor; diffusion_architecture.py:243-262)."""
        pf = st.prev_future
        if pf is None:
            return
        st.prev_future = None
        src = pf.state
        if src.latent_ready is None:
            raise capi.RgError("prev_latent: the batch this pending latent belongs to has not been queued for sampling yet "
                               "(submit the windows in order)")
        stream.wait_event(src.latent_ready)
        self._used_on(stream, src.x_out, st.in_seq)
        with torch.cuda.stream(stream):
            lat = self.model.post_process(src.x_out)
            if pf.rows is not None:
                lat = lat.index_select(0, torch.tensor(pf.rows, device="cpu").pin_memory().to(self.device, non_blocking=True))
            h = self.model.weights.h
            for idx in st.idx_groups:
                h.call("copy_rows", lat.contiguous(), st.in_seq, st.B, 1, st.D, st.T, idx[-1], st.T, idx[0])

    def _take_slot(self, pid, main, lanes):
        """Next set of sessions / graph buffers of pipeline `pid`; the caller's stream waits for the chains that last used
        it on any of `lanes` (sessions and graphs are keyed by (lane, slot) whatever pipeline drove them)."""
        slot = self._slots[pid] = (self._slots.get(pid, 0) + 1) % self.slots
        self._wait_slot(main, lanes, slot)
        self._slot = slot
        return slot

    def _wait_slot(self, main, lanes, slot):
        for lane in dict.fromkeys(lanes):
            for ev in self._slot_done.get((lane, slot), ()):
                main.wait_event(ev)

    @staticmethod
    def _lanes_of(*plans):
        return list(dict.fromkeys(lane for plan in plans for lane, _, _, _ in plan))

    # ------------------------------------------------------------------ co-batched pipeline
    def submit(self, **kwargs):
        """Queue a batch whose sampling is DEFERRED to a later call: its exemplars are inverted now, in the same denoiser
        launches that run the sampling loop of the batch submitted one rotation of the `batch_lanes` earlier
        (sampler.cobatched_loop: one forward per step for the 16 clips of one batch and the 48 exemplars of the other).
        Returns the results of that earlier batch (asynchronous: `done_event` / `done_stream`, see `wait_results`), or None
        while the pipeline fills; results come out in submission order; `flush()` finishes what is pending.  Same arguments as forward(); needs async_results=True.  Batches that cannot
        be co-batched (no inversion, a lane without exemplars, another batch size) are completed on their own."""
        if not self.async_results or self.slots < 2:
            raise capi.RgError("submit() needs MotionDiffusion(async_results=True, slots >= 2)")
        self._cob = dict(lane=self._submitted if (self.cobatch_lanes == "batch" and self.batch_lanes > 1) else None)
        self._submitted += 1
        try:
            self.forward(**kwargs)
        finally:
            self._cob = None
        return self._ready.popleft() if self._ready else None

    def flush(self):
        """Finish the pending batches of submit() (their sampling loops alone) and return every result not handed out yet,
        in submission order."""
        cob, self._cob = self._cob, dict(lane=0)     # (the draining chains are chains of the pipeline: their sessions publish
        try:                                          #  the workgroups they hold to the launch-form arbitration, _session_opts)
            for pid in sorted(self._pend, key=lambda p: self._pend[p].seq):
                self._ready.append(self._finish_alone(self._pend.pop(pid)))
        finally:
            self._cob = cob
        out = list(self._ready)
        self._ready.clear()
        return out

    def _finish_alone(self, st):
        """Sampling loop + tail of a batch whose exemplars are already inverted and spliced."""
        main = st.main = torch.cuda.current_stream()
        st.slot = self._take_slot(st.pid, main, self._lanes_of(st.plan_s))
        for lane, stream, b0, b1 in st.plan_s:
            self._set_conditions(b1 - b0, "sample", lane, st.word[b0:b1], st.audio[b0:b1], st.spk[b0:b1],
                                 st.motion_mask[b0:b1], {c: st.qmask[b0:b1] for c in denoiser.CONDS})
        self._sampling_pass(st)
        return self._tail(st)

    def _pending_upto(self, seq):
        """Pipeline ids of the pending batches submitted no later than submission `seq`, oldest first."""
        return sorted((p for p in self._pend if self._pend[p].seq <= seq), key=lambda p: self._pend[p].seq)

    def _set_conditions_pair(self, sess, key, own, a, b, n_a):
        """Conditions of a session that holds two batches side by side (clips [0, n_a): a, the rest: b), one graph."""
        dev = self.model.weights.dev
        ins = {}
        for tag, (word, audio, spk, mask, qm) in (("a", a), ("b", b)):
            ins["word_" + tag], ins["audio_" + tag] = word.to(dev).float(), audio.to(dev).float()
            ins["spk_" + tag], ins["mask_" + tag] = spk.to(dev).long(), mask.to(dev).float()
            for c in denoiser.CONDS:
                ins["q_%s_%s" % (c, tag)] = qm[c].to(dev).float()

        def run(st):
            for tag, off, fin in (("a", 0, False), ("b", n_a, True)):
                sess.set_conditions(st["word_" + tag], st["audio_" + tag], st["spk_" + tag], st["mask_" + tag],
                                    {c: st["q_%s_%s" % (c, tag)] for c in denoiser.CONDS}, offset=off, finalize=fin)
            return ()
        self._graph_run(key + tuple(tuple(v.shape) for v in ins.values()), ins, run, owner=own)

    def _submit_chain(self, st):
        """Second half of a submit(): [sampling of the pending batch || inversion of this one] per lane, splice, tail of
        the pending batch; this batch becomes the pending one."""
        pend, S, T, D, dev = self._pend.get(st.pid), st.S, st.T, st.D, self.device
        lanes = [(lane, stream, b0, b1, self._exemplars(st, b0, b1)) for lane, stream, b0, b1 in st.plan] if st.use_inversion else []
        so = self.session_options
        # (the engine a session RESOLVES to: without sequence streams -- L > 8, ff_size != 1024, T > 48 -- it is the chain)
        seq = so.get("engine") != "chain" and getattr(self.model.weights, "seq_streams", None) is not None
        groups_ok = self.precision == "bf16" and (seq or (
            so.get("styl_prepass", True)))
        can_defer = (groups_ok and st.use_inversion and not st.visualize_inversion and not st.ddpm and st.plan == st.plan_s
                     and all(ex for *_, ex in lanes))
        same = pend is not None and can_defer and (pend.B, pend.T) == (st.B, st.T) and \
            [(b0, b1) for _, _, b0, b1 in pend.plan] == [(b0, b1) for _, _, b0, b1 in st.plan]
        if pend is not None and not same:
            # finishes alone (in the other slot's sessions) -- behind everything submitted before it: results are handed out
            # in submission order, and this lane's pending batch is the oldest only while the rotation is undisturbed (a
            # batch without inversion picks its lane among `base_lanes`, not `batch_lanes`)
            for p in self._pending_upto(pend.seq):
                self._ready.append(self._finish_alone(self._pend.pop(p)))
            self._slot = self._slots[st.pid] = st.slot
            pend = None
        if not can_defer:                              # nothing to share launches with later: complete it now,
            for p in sorted(self._pend, key=lambda p: self._pend[p].seq):     # after everything submitted before it
                self._ready.append(self._finish_alone(self._pend.pop(p)))
            self._slot = self._slots[st.pid] = st.slot
            for lane, stream, b0, b1 in st.plan_s:
                self._set_conditions(b1 - b0, "sample", lane, st.word[b0:b1], st.audio[b0:b1], st.spk[b0:b1],
                                     st.motion_mask[b0:b1], {c: st.qmask[b0:b1] for c in denoiser.CONDS})
            self._inversion_pass(st)
            self._sampling_pass(st)
            self._ready.append(self._tail(st))
            return None
        if pend is None:                               # the pipeline fills: inversion alone
            self._inversion_pass(st)
            for lane, stream, _, _ in st.plan:                                     # (no tail marks it)
                self._slot_done[(lane, st.slot)] = [stream.record_event()]
            self._pend[st.pid] = st
            return None
        main = st.main
        self._fill_prev(pend, main)
        guided = pend.use_insertion_guidance
        gi, lr = tuple(int(v) for v in pend.guidance_iters), float(pend.guidance_lr)
        work = []
        for lane, stream, b0, b1, ex in lanes:         # front end (caller's stream): inputs of the shared sessions
            Bl = b1 - b0
            Ep, cat = self._exemplar_inputs(st, ex, main)
            okey = (Bl + Ep, "cobatch", lane, self._slot)
            sess = self._session(Bl + Ep, "cobatch", lane)
            eqm = {c: pad_rows(torch.stack([st.qmask[b] for b, _ in ex]), Ep) for c in denoiser.CONDS}
            # (on the caller's stream.  Round 6 moved these projections in front of the chain on the LANE's stream -- with
            #  device-arbitrated launch forms the caller's stream paces the pipeline, profiles/r06k_timed_region.txt -- and got
            #  32.5-33.6 instead of 30.9-31.4 ms per step AND two unverified runs in four: profiles/r06m_ab.txt.  Removed.)
            self._set_conditions_pair(
                sess, ("cond2", Bl, Ep, lane, self._slot), okey,
                (pend.word[b0:b1], pend.audio[b0:b1], pend.spk[b0:b1], pend.motion_mask[b0:b1], {c: pend.qmask[b0:b1] for c in denoiser.CONDS}),
                (cat("retr_text"), cat("retr_audio"), cat("retr_spkid"), cat("retr_motion_mask"), eqm), Bl)
            work.append((lane, stream, b0, b1, ex, Ep, sess, okey, cat("retr_motion_latent").float().contiguous()))
        for lane, stream, b0, b1, ex, Ep, sess, okey, x_e in work:
            Bl = b1 - b0
            stream.wait_stream(main)
            self._used_on(stream, pend.x, pend.in_seq, pend.inseq_noise, pend.invl, pend.x_out, x_e, st.start_noise, st.invl, st.qmask)
            with torch.cuda.stream(stream):
                sl = lambda t, dim: None if t is None else (t[b0:b1] if dim == 0 else t[:, b0:b1])
                ins = dict(xa=sl(pend.x, 0), in_seq=sl(pend.in_seq, 0), noise=sl(pend.inseq_noise, 1),
                           invl=sl(pend.invl, 1) if guided else None, xb=x_e)

                def loop(s, sess=sess, Bl=Bl, Ep=Ep):
                    x_all = torch.cat([s["xa"], s["xb"]], dim=0).contiguous()
                    out_b = torch.empty(S, Ep, T, D, device=dev)
                    sampler.cobatched_loop(sess, x_all, Bl, out_b, inverted_a=s["invl"], guidance_iters=gi, guidance_lr=lr,
                                           inseq_noise_a=s["noise"], in_seq_a=s["in_seq"], tail_glue=self.tail_glue)
                    return x_all[:Bl], out_b
                key = ("cobatch", Bl, Ep, lane, T, self._slot, guided, pend.in_seq is not None, gi, lr, self.tail_glue)
                xl, inv = self._graph_run(key, ins, loop, owner=okey)
                pend.x_out[b0:b1].copy_(xl)
                self._splice(st, ex, inv, Ep)
                if st.use_insertion_guidance and st.use_prev_latent and st.prev_latent is not None:
                    for idx in st.idx_groups:
                        st.invl[:, b0:b1, idx[0], :] = 0
        pend.main, pend.slot = main, self._slot
        self._ready.append(self._tail(pend))
        self._pend[st.pid] = st
        return None

    # ------------------------------------------------------------------ phases of forward (after the front end)
    def _exemplars(self, st, b0, b1):
        return [(b, q) for b in range(b0, b1) for q in st.retrieval_dict["retr_uncropped_latents"][b].keys()]

    def _exemplar_inputs(self, st, ex, stream):
        """(Ep, cat): cat(key) = the lane's exemplars' `key` tensors stacked and padded to the Ep bucket."""
        dev = self.device
        Ep = bucket(len(ex))
        lat = lambda b, q: st.retrieval_dict["retr_uncropped_latents"][b][q]

        def cat(key):
            parts = [lat(b, q)[key].to(dev) for b, q in ex]
            self._used_on(stream, *parts)
            return pad_rows(torch.cat(parts, dim=0), Ep)
        return Ep, cat

    def _splice(self, st, ex, inv, Ep):
        """The inverted exemplar rows into the start noise (and, level by level, into the guidance target)."""
        h, rd = self.model.weights.h, st.retrieval_dict
        lvl = st.inversion_start_time % st.S
        rows = []
        for e, (b, q_idx) in enumerate(ex):
            r0, r1 = rd["retr_startends"][b][q_idx]
            q0, q1 = rd["query_startends"][b][q_idx]
            capi.require(r1 - r0 == q1 - q0, "unsupported argument: requires r1 - r0 == q1 - q0")
            rows.append((e, b, int(r0), int(q0), int(r1 - r0)))
        # one launch per 64 exemplars (rg_splice_many) instead of two per exemplar: 96 small launches behind a lane's chain
        invl = st.invl if st.use_insertion_guidance else None
        capi.require(inv.is_contiguous() and st.start_noise.is_contiguous() and (invl is None or invl.is_contiguous()),
                     "splice: contiguous tensors expected")
        s_ = torch.cuda.current_stream().cuda_stream
        for c0 in range(0, len(rows), sampler.SPLICE_MAX):
            tab = sampler.SpliceTable()
            chunk = rows[c0:c0 + sampler.SPLICE_MAX]
            tab.n = len(chunk)
            for i, (e, b, r0, q0, n) in enumerate(chunk):
                tab.e[i], tab.b[i], tab.r0[i], tab.q0[i], tab.nrows[i] = e, b, r0, q0, n
            rc = h.lib.rg_splice_many(h._h, ctypes.byref(tab), inv.data_ptr(), st.start_noise.data_ptr(),
                                      None if invl is None else invl.data_ptr(), st.T, st.D, st.n_lat, lvl, st.S, Ep, st.B, ctypes.c_void_p(s_))
            if rc != 0:
                raise capi.RgError("rg_splice_many failed (%d): %s" % (rc, h.lib.rg_last_error(h._h).decode()))

    def _inversion_pass(self, st):
        """lanes: exemplar inversion -> splice, per clip group, concurrently."""
        dev, S, T, D = self.device, st.S, st.T, st.D
        for lane, stream, b0, b1 in st.plan:
            stream.wait_stream(st.main)
            self._used_on(stream, st.start_noise, st.invl, st.qmask)
            with torch.cuda.stream(stream):
                if st.use_inversion:
                    ex = self._exemplars(st, b0, b1)
                    if ex:
                        # the lane's E exemplars run as a batch of Ep = E rounded up (padding = copies of exemplar 0,
                        # inverted and dropped): sessions, condition graphs and inversion graphs exist per Ep only
                        E = len(ex)
                        Ep, cat = self._exemplar_inputs(st, ex, stream)
                        esess = self._session(Ep, "invert", lane)
                        with self._phase("exemplar_conditions"):
                            if st.early_cond.get(lane) != E:   # not already projected while the exemplars were encoded
                                eqm = {c: pad_rows(torch.stack([st.qmask[b] for b, _ in ex]), Ep) for c in denoiser.CONDS}
                                self._set_conditions(Ep, "invert", lane, cat("retr_text"), cat("retr_audio"), cat("retr_spkid"),
                                                     cat("retr_motion_mask"), eqm)
                            x_e = cat("retr_motion_latent").float().contiguous()
                        with self._phase("inversion"):
                            (inv,) = self._graph_run(("invert", Ep, lane, T, self._slot), dict(x=x_e), lambda s, esess=esess, Ep=Ep: (
                                sampler.ddim_reverse_sample_loop(esess, s["x"], torch.empty(S, Ep, T, D, device=dev)),),
                                owner=(Ep, "invert", lane, self._slot))
                        self._splice(st, ex, inv, Ep)
                        if st.visualize_inversion:
                            # sanity check of the reference (diffusion_architecture.py:357-382): every inversion level
                            # and the DDIM reconstruction from the last level, decoded after the sampling below
                            (rec,) = self._graph_run(("recon", Ep, lane, T, self._slot), dict(x=inv[S - 1]), lambda s, esess=esess: (
                                sampler.ddim_sample_loop(esess, s["x"]),), owner=(Ep, "invert", lane, self._slot))
                            for e in range(E):
                                st.vis_inv.append(inv[:, e])
                                st.vis_pairs.append(torch.stack([x_e[e], rec[e]]))
                    if st.use_insertion_guidance and st.use_prev_latent and st.prev_latent is not None:
                        for idx in st.idx_groups:
                            st.invl[:, b0:b1, idx[0], :] = 0

    def _sampling_pass(self, st):
        """The sampling loops.  Every lane's inversion is queued before the first sampling graph is launched (a graph
        launch costs the host ~1.5 ms: lane 1 would otherwise start 3 ms behind lane 0)."""
        self._fill_prev(st, st.main)
        T, x, in_seq, invl = st.T, st.x, st.in_seq, st.invl
        inverted = [(stream, stream.record_event()) for _, stream, _, _ in st.plan]
        for lane, stream, b0, b1 in st.plan_s:
            Bl = b1 - b0
            sess, okey = self._session(Bl, "sample", lane), (Bl, "sample", lane, self._slot)
            stream.wait_stream(st.main)
            self._used_on(stream, x, in_seq, st.inseq_noise, invl, st.ddpm_noise, st.x_out)
            if st.plan_s != st.plan:             # a sampling lane then reads rows spliced by several inversion lanes
                for other, ev in inverted:
                    if other is not stream:
                        stream.wait_event(ev)
            with torch.cuda.stream(stream):
                sl = lambda t, dim: None if t is None else (t[b0:b1] if dim == 0 else t[:, b0:b1])
                loop_in = dict(x=sl(x, 0), in_seq=sl(in_seq, 0), noise=sl(st.inseq_noise, 1), invl=sl(invl, 1))
                with self._phase("sampling"):
                    if st.ddpm:
                        (xl,) = self._graph_run(("ddpm", Bl, lane, T, self._slot), dict(x=sl(x, 0), noise=sl(st.ddpm_noise, 1)),
                                                lambda s, sess=sess: (sampler.p_sample_loop(sess, s["x"], s["noise"]),), owner=okey)
                    elif st.use_insertion_guidance:
                        gi, lr = tuple(int(v) for v in st.guidance_iters), float(st.guidance_lr)
                        key = ("guided", Bl, lane, T, self._slot, in_seq is not None, gi, lr)
                        (xl,) = self._graph_run(key, loop_in, lambda s, sess=sess, gi=gi, lr=lr: (
                            sampler.ddim_guided_sample_loop(sess, s["x"], s["invl"], gi, lr, s["noise"], in_seq=s["in_seq"]),),
                            owner=okey)
                    else:
                        key = ("sample", Bl, lane, T, self._slot, in_seq is not None)
                        (xl,) = self._graph_run(key, loop_in, lambda s, sess=sess: (sampler.ddim_sample_loop(
                            sess, s["x"], in_seq=s["in_seq"], inseq_noise=s["noise"]),), owner=okey)
                    st.x_out[b0:b1].copy_(xl)

    def _tail(self, st):
        """Latent post-processing and VAE decode once every lane is done: on the caller's stream, or, with asynchronous
        submission (where that one is already queueing the next batch), on the first sampling lane's stream."""
        gre, results, main, S, T, D, B = self.model.gesture_rep_encoder, st.results, st.main, st.S, st.T, st.D, st.B
        # (asynchronous: the lanes take turns, so that the decode does not always delay the same lane's next chain)
        tail_lane, tail = st.plan_s[self._tail_turn % len(st.plan_s)][:2] if st.run_async else (-1, main)
        if st.run_async and self.decode_on_own_stream and self._decode_stream is not None:
            tail_lane, tail = -2, self._decode_stream
        self._tail_turn += 1
        if st.run_async and st.via_submit:   # (submission order is the contract of submit() / flush(): a scheduling bug must
            #                                    not pass as a result; a plain forward() between them is the caller's own order)
            capi.require(st.seq >= self._last_out_seq, "internal: results handed out against the submission order")
            self._last_out_seq = st.seq
        if self._jitter is not None:
            self._jitter(tail, "tail")
        for _, stream, _, _ in st.plan + st.plan_s:
            if stream is not tail:
                tail.wait_stream(stream)
        st.latent_ready = st.handle.latent_ready = tail.record_event()   # (a later batch's pending prev_latent waits for this, not for the decode)
        self._used_on(tail, st.x_out, *st.vis_inv, *st.vis_pairs)
        with torch.cuda.stream(tail):
            output = self.model.post_process(st.x_out)
            results["prev_latentout"] = output
            with self._phase("vae_decode"):
                up, lo, fa, ha, tr, ex_, co = self._graph_run(("dec", B, gre.part_streams is None, tail_lane), dict(z=output),
                                                                lambda s: gre.decode(s["z"]))
            results["pred_upper"], results["pred_lower"], results["pred_facepose"] = up, lo, fa
            results["pred_hands"], results["pred_transl"], results["pred_exps"] = ha, tr, ex_
            results["pred_contact"] = co
            if st.use_inversion and st.visualize_inversion and st.vis_inv:
                # diffusion_architecture.py:488-571: decoded inversion levels [n_exemplars, S, frames, *] and decoded
                # (exemplar, reconstruction) pairs [n_exemplars, 2, frames, *]
                n_ex = len(st.vis_inv)
                keys = ("upper", "lower", "facepose", "hands", "transl", "exps")
                for name, lat, k in (("inverted_output", torch.stack(st.vis_inv).reshape(n_ex * S, T, D), S),
                                     ("reconspair_output", torch.stack(st.vis_pairs).reshape(n_ex * 2, T, D), 2)):
                    lat = self.model.post_process(lat.contiguous())
                    parts = [[] for _ in keys]
                    for c0 in range(0, lat.shape[0], 128):            # decoded in slabs of 128 like the reference (:496-528)
                        dec = gre.decode(lat[c0:c0 + 128].contiguous())
                        for j in range(6):
                            parts[j].append(dec[j])
                    for j, kname in enumerate(keys):
                        t = torch.cat(parts[j], dim=0)
                        results["%s_%s" % (name, kname)] = t.reshape(n_ex, k, t.shape[1], -1)
        done = torch.cuda.Event()
        done.record(tail)
        self._used_on(main, *[v for v in results.values() if torch.is_tensor(v)])   # the caller reads them on its stream
        if st.run_async:
            # done_stream: where the results were produced; work queued there needs no wait (ROCm maps streams onto 4
            # hardware queues: a consumer stream of its own that waits for done_event can block whichever of the
            # caller's / search / lane streams shares its queue, and with it the next batch's front end)
            results["done_event"], results["done_stream"] = done, tail
            results = AsyncResults(results)      # the first read of an entry waits for the batch on the reader's stream
            for lane in self._lanes_of(st.plan, st.plan_s):
                self._slot_done[(lane, st.slot)] = [done]
            self._inflight.append(done)
            # (whole batches in rotation: every lane's next chain is queued behind its running one before the host waits)
            #  measured: 4 lanes 38.4 ms per step with 4 in flight, 36.2 with 6; 2 lanes 38.7 with 2, 43.7 with 4)
            rot = self.base_lanes if not st.use_inversion else (2 * self.batch_lanes + 4 if st.pid is not None else 0)
            while len(self._inflight) > max(self.max_inflight, rot):
                self._inflight.popleft().synchronize()
        elif tail is not main:
            main.wait_event(done)
        return results
