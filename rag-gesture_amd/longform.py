"""Long-form synthesis driver: tools/longform_synthesis.py around `model(**chunk_data)`.

    :262-287  window split (hop = max_seq_len - frame_chunk_size = 135 frames), zero padding of the tail,
              motion_mask / speaker_ids padded by repeating their last frames
    :300-377  per-window slices of the motion tensors and of the time-stamped annotations (text segments,
              discourse relations, prominence, gesture labels: kept if fully inside the window, shifted)
    :378-398  inference_kwargs with use_prev_latent = True and prev_latent = the previous window's latent
    :431-476  15-frame linear blend with the motion so far, in 6D for the pose           -> rg_blend_aa / rg_blend_linear
    :714-741  final 15 -> 30 fps interpolation of the whole motion                        -> packing.upsample_*

Per-window conditioning features (wav2vec2 on the window's audio, BERT on its words: :320-343) are not part
of this path: the caller supplies `features(cidx, t0, t1, annotations) -> dict` with at
least `audio` [B,499,768] and `text_features`.

Throughput (BASELINE config 5: 10 clips x N windows on 8 GPUs): the windows of ONE clip are sequential (prev-latent
chain), the clips are independent.  `run_many` therefore (a) shards the clips over the ranks of the process group
(`dist.shard_range`, as the tool's DistributedSampler does, :211, 256) and (b) runs window k of all clips of a rank as
ONE forward of batch = clips still running, each clip keeping its own prev-latent chain and blend state.  The reference
runs batch 1 per clip; per clip the results are the same (clips never interact inside the model).
"""
import os

import contextlib

import torch

from . import capi, dist as rg_dist, packing

MOTION_KEYS = ("motion", "motion_upper", "motion_lower", "motion_face", "motion_hands", "contact", "trans", "facial", "beta",
               "word")
REPEAT_KEYS = ("motion_mask", "speaker_ids")


def window_bounds(sample_len, seqlen=150, overlap=15):
    """longform_synthesis.py:262-265 -> (starts, ends, frames of padding needed at the end)."""
    hop = seqlen - overlap
    starts = [0] + list(range(hop, sample_len, hop))
    ends = [s + seqlen for s in starts]
    return starts, ends, max(0, ends[-1] - sample_len)


def pad_tail(data, remainder):
    """longform_synthesis.py:267-287 (zero padding; mask / speaker ids repeat their last `remainder` frames)."""
    if remainder <= 0:
        return data
    out = dict(data)
    for k in MOTION_KEYS:
        if k in out and torch.is_tensor(out[k]):
            z = torch.zeros(out[k].shape[0], remainder, out[k].shape[2], dtype=out[k].dtype, device=out[k].device)
            out[k] = torch.cat([out[k], z], dim=1)
    for k in REPEAT_KEYS:
        if k in out and torch.is_tensor(out[k]):
            out[k] = torch.cat([out[k], out[k][:, -remainder:]], dim=1)
    return out


def window_annotations(data, t0, t1):
    """longform_synthesis.py:333-377 for a batch-of-one sample: annotations fully inside [t0, t1], window time."""
    first = lambda k: (data.get(k) or [[]])[0]
    segs = [[[s[0][0] - t0, s[0][1] - t0], s[1]] for s in first("text_segments") if s[0][0] >= t0 and s[0][1] <= t1]
    disc = [(d[0], d[1], d[2], d[3], d[4] - t0, d[5] - t0, d[6] - t0, d[7] - t0) for d in first("discourse")
            if d[4] >= t0 and d[5] <= t1]
    prom = [(p[0], p[1] - t0, p[2] - t0, p[3]) for p in first("prominence") if p[1] >= t0 and p[2] <= t1]
    labels = [dict(start=g["start"] - t0, end=g["end"] - t0, name=g["name"], word=g["word"])
              for g in first("gesture_labels") if g["start"] >= t0 and g["end"] <= t1]
    return dict(text_segments=[segs], discourse=[disc], prominence=[prom], gesture_labels=[labels])


def blend_window(prev, cur, overlap):
    """prev / cur = (motion [B,*,165] axis-angle, facial, trans) device tensors; returns the extended triple
    (longform_synthesis.py:431-476, prediction branch).  The whole new window passes through 6D like the
    reference's, the first `overlap` frames are blended with the tail of `prev`."""
    pm, pf, pt = (t.float().contiguous() for t in prev)
    cm, cf, ct = (t.float().contiguous().clone() for t in cur)
    B, n, dim = cm.shape
    h = capi.get_handle(cm.device.index)
    out_m = torch.empty_like(cm)
    h.call("blend_aa", pm[:, -overlap:].contiguous(), cm, out_m, B, n, dim // 3, overlap)
    h.call("blend_linear", pf[:, -overlap:].contiguous(), cf, B, n, cf.shape[-1], overlap)
    h.call("blend_linear", pt[:, -overlap:].contiguous(), ct, B, n, ct.shape[-1], overlap)
    return (torch.cat([pm[:, :-overlap], out_m], 1), torch.cat([pf[:, :-overlap], cf], 1),
            torch.cat([pt[:, :-overlap], ct], 1))


@contextlib.contextmanager
def pinned_lanes(model, on, rotation, form=None):
    """While a pipelined long-form pass runs, the model's submit() rotates over `rotation` lanes and chooses its launch forms
    for `form` chains side by side; both settings are the caller's again afterwards -- also when retrieval, a capi.require or
    an allocation fails in between (a shared model must not stay pinned)."""
    before = (getattr(model, "rotation_lanes", None), getattr(model, "form_lanes", None))
    if on:
        model.rotation_lanes = rotation
        if form is not None:
            model.form_lanes = form
    try:
        yield
    finally:
        if on:
            model.rotation_lanes, model.form_lanes = before


class LongformSynthesizer:
    """model: a rag-gesture_amd MotionDiffusion.  `run(data, features, **inference flags)` mirrors the
    per-sample loop of tools/longform_synthesis.py and returns 30-fps numpy arrays cut to the sample length."""

    def __init__(self, model, motion_fps=15, max_seq_len=150, overlap=None, target_fps=30):
        self.model, self.fps, self.seqlen, self.target_fps = model, motion_fps, max_seq_len, target_fps
        self.overlap = overlap if overlap is not None else model.model.cfg["frame_chunk_size"]

    def run(self, data, features, use_inversion=False, insertion_guidance=False, guidance_iters=None, guidance_lr=0.1,
            outpaint=False, inversion_start_time=-1, retrieval_method="discourse", noise_tape=None, with_gt=False,
            pipelined=None):
        """with_gt: also carry the ground-truth triple (the model's returned motion / facial / trans inputs) through the
        same overlap blend and interpolation, as the tool does for gt_motion.npz (longform_synthesis.py:480-520, 722-745);
        adds gt_poses / gt_expressions / gt_trans to the result.
        pipelined (default: model.async_results): windows through model.submit() / flush(), see run_many."""
        sample_len = data["motion"].shape[1]
        starts, ends, remainder = window_bounds(sample_len, self.seqlen, self.overlap)
        data = pad_tail(data, remainder)
        if pipelined is None:
            pipelined = bool(getattr(self.model, "async_results", False)) and not outpaint
        if pipelined:
            capi.require(not self.model._pend and not self.model._ready,
                         "long-form synthesis (pipelined): the model's submit() pipeline must be empty (call flush() first)")
        state = dict(so_far=None, gt_so_far=None, n=0)
        prev_latent, latents = None, []
        # the windows of ONE clip are a dependent chain (window k + 1 samples from window k's latent): they stay on one lane,
        # where window k + 1's exemplar inversion shares the denoiser launches of window k's sampling, instead of rotating

        def take(out):
            cidx = state["n"]
            state["n"] += 1
            latents.append(out["prev_latentout"])
            cur = (packing.scatter_parts(out["pred_upper"], out["pred_lower"], out["pred_hands"], out["pred_facepose"]),
                   out["pred_exps"].float(), out["pred_transl"].float())
            state["so_far"] = cur if cidx == 0 else blend_window(state["so_far"], cur, self.overlap)
            if with_gt:
                dev = cur[0].device
                gt = tuple(out[k].to(dev).float() for k in ("motion", "facial", "trans"))
                state["gt_so_far"] = gt if cidx == 0 else blend_window(state["gt_so_far"], gt, self.overlap)

        with pinned_lanes(self.model, pipelined, rotation=1):      # (restored whatever happens in between: ADVICE r05)
            for cidx, (c0, c1) in enumerate(zip(starts, ends)):
                t0, t1 = c0 / self.fps, c1 / self.fps
                chunk = {k: data[k][:, c0:c1] for k in MOTION_KEYS + REPEAT_KEYS if k in data and torch.is_tensor(data[k])}
                capi.require(chunk["motion"].shape[1] == self.seqlen,
                        "unsupported argument: requires chunk[\"motion\"].shape[1] == self.seqlen")
                chunk["motion_length"] = [self.seqlen] * chunk["motion"].shape[0]
                ann = window_annotations(data, t0, t1)
                chunk.update(ann)
                chunk.update(features(cidx, t0, t1, ann))
                if "sample_name" in data:
                    chunk["sample_name"] = [data["sample_name"][0].replace("/0", "/%d" % cidx)]
                chunk["retrieval_method"] = retrieval_method
                ikw = dict(use_inversion=use_inversion, outpaint=outpaint, inversion_start_time=inversion_start_time,
                           insertion_guidance=insertion_guidance, guidance_lr=guidance_lr, use_prev_latent=True,
                           prev_latent=prev_latent)
                if guidance_iters is not None:
                    ikw["guidance_iters"] = guidance_iters
                if noise_tape is not None:
                    ikw["noise_tape"] = noise_tape
                chunk["inference_kwargs"] = ikw
                if pipelined:
                    out = self.model.submit(**chunk)
                    prev_latent = self.model.pending_latent()     # bound when the next window's sampling is queued
                    if out is not None:
                        take(out)
                else:
                    out = self.model(**chunk)
                    prev_latent = out["prev_latentout"]
                    take(out)
            if pipelined:
                for out in self.model.flush():
                    take(out)
        capi.require(state["n"] == len(starts), "long-form synthesis: windows left in the pipeline")
        so_far, gt_so_far = state["so_far"], state["gt_so_far"]
        motion, facial, trans = so_far
        scale = self.target_fps // self.fps
        if scale != 1:
            motion, facial, trans = (packing.upsample_motion(motion, scale), packing.upsample_features(facial, scale),
                                     packing.upsample_features(trans, scale))
        n_out = sample_len * scale
        cut = lambda t: t[0, :n_out].detach().cpu().numpy()
        result = dict(poses=cut(motion), expressions=cut(facial), trans=cut(trans), latents=latents,
                      windows=list(zip(starts, ends)))
        if with_gt:
            gm, gf, gt_ = gt_so_far
            if scale != 1:
                gm, gf, gt_ = packing.upsample_motion(gm, scale), packing.upsample_features(gf, scale), \
                    packing.upsample_features(gt_, scale)
            result.update(gt_poses=cut(gm), gt_expressions=cut(gf), gt_trans=cut(gt_))
        return result

    def run_many(self, clips, features, use_inversion=False, insertion_guidance=False, guidance_iters=None, guidance_lr=0.1,
                 outpaint=False, inversion_start_time=-1, retrieval_method="discourse", noise_tape=None, with_gt=False,
                 shard=True, gather=False, pipelined=None):
        """clips: list of batch-of-one sample dicts (any lengths); features(clip_index, cidx, t0, t1, annotations) -> dict.
        shard: with torch.distributed initialised, this rank takes clips[shard_range(len(clips), rank, world)].
        pipelined (default: model.async_results): the windows go through model.submit() / flush() -- window k + 1 is
        submitted with window k's latent still PENDING (pipeline.PendingLatent): its retrieval and exemplar inversion do not
        need it and share their denoiser launches with window k's sampling loop; same results as the sequential loop.
        Returns {clip_index: result dict as `run`} for this rank's clips (gather=True: for all clips on every rank,
        exchanged with all_gather_object -- the final result gather of mogen/apis/test.py:129-160)."""
        import torch.distributed as td
        mine = list(range(len(clips)))
        if shard and td.is_available() and td.is_initialized() and td.get_world_size() > 1:
            lo, hi = rg_dist.shard_range(len(clips), td.get_rank(), td.get_world_size())
            mine = mine[lo:hi]
        state = {}
        for ci in mine:
            sample_len = clips[ci]["motion"].shape[1]
            starts, ends, remainder = window_bounds(sample_len, self.seqlen, self.overlap)
            state[ci] = dict(data=pad_tail(clips[ci], remainder), starts=starts, ends=ends, sample_len=sample_len,
                             prev=None, so_far=None, gt_so_far=None, latents=[])
        n_win = max((len(st["starts"]) for st in state.values()), default=0)
        if pipelined is None:
            pipelined = bool(getattr(self.model, "async_results", False)) and not outpaint
        if pipelined:
            capi.require(not self.model._pend and not self.model._ready,
                         "long-form synthesis (pipelined): the model's submit() pipeline must be empty (call flush() first)")
        queued = []          # (cidx, clips of the window) of the windows whose results are still in the pipeline
        # window k + 1 samples from window k's latents: at most two window batches are ever busy (one sampling, the next one
        # inverting its exemplars).  They rotate over up to four lanes (as in round 4), and the launch forms are chosen for the
        # two chains that really run side by side, not for the model's whole rotation (pipeline._seq_form_auto)

        def take(out):
            cidx, act = queued.pop(0)
            motion = packing.scatter_parts(out["pred_upper"], out["pred_lower"], out["pred_hands"], out["pred_facepose"])
            for j, ci in enumerate(act):
                st = state[ci]
                st["prev"] = out["prev_latentout"][j:j + 1]
                st["latents"].append(st["prev"])
                cur = (motion[j:j + 1], out["pred_exps"][j:j + 1].float(), out["pred_transl"][j:j + 1].float())
                st["so_far"] = cur if cidx == 0 else blend_window(st["so_far"], cur, self.overlap)
                if with_gt:
                    dev = cur[0].device
                    gt = tuple(out[k][j:j + 1].to(dev).float() for k in ("motion", "facial", "trans"))
                    st["gt_so_far"] = gt if cidx == 0 else blend_window(st["gt_so_far"], gt, self.overlap)

        with pinned_lanes(self.model, pipelined, rotation=min(4, getattr(self.model, "batch_lanes", 4)), form=2):
            pending, act_prev = None, None
            for cidx in range(n_win):
                act = [ci for ci in mine if cidx < len(state[ci]["starts"])]     # clips that still have a window cidx
                chunks = []
                for ci in act:
                    st = state[ci]
                    c0, c1 = st["starts"][cidx], st["ends"][cidx]
                    t0, t1 = c0 / self.fps, c1 / self.fps
                    data = st["data"]
                    chunk = {k: data[k][:, c0:c1] for k in MOTION_KEYS + REPEAT_KEYS if k in data and torch.is_tensor(data[k])}
                    capi.require(chunk["motion"].shape[0] == 1 and chunk["motion"].shape[1] == self.seqlen,
                            "unsupported argument: requires chunk[\"motion\"].shape[0] == 1 and chunk[\"motion\"].shape[1] == self.seqlen")
                    ann = window_annotations(data, t0, t1)
                    chunk.update(ann)
                    chunk.update(features(ci, cidx, t0, t1, ann))
                    chunk["sample_name"] = [data["sample_name"][0].replace("/0", "/%d" % cidx)] if "sample_name" in data \
                        else ["clip%d/%d" % (ci, cidx)]
                    chunks.append(chunk)
                # one batch: tensors concatenated along the clip dimension, per-clip lists chained
                batch = {}
                for k in chunks[0]:
                    v0 = chunks[0][k]
                    if torch.is_tensor(v0):
                        batch[k] = torch.cat([c[k] for c in chunks], dim=0)
                    elif isinstance(v0, (list, tuple)):
                        batch[k] = [x for c in chunks for x in c[k]]
                    else:
                        batch[k] = v0
                batch["motion_length"] = [self.seqlen] * len(act)
                batch["retrieval_method"] = retrieval_method
                if cidx == 0:
                    prev = None
                elif pipelined:
                    prev = pending.select([act_prev.index(ci) for ci in act])    # bound when this window's sampling is queued
                else:
                    prev = torch.cat([state[ci]["prev"] for ci in act], dim=0)
                ikw = dict(use_inversion=use_inversion, outpaint=outpaint, inversion_start_time=inversion_start_time,
                           insertion_guidance=insertion_guidance, guidance_lr=guidance_lr, use_prev_latent=True, prev_latent=prev)
                if guidance_iters is not None:
                    ikw["guidance_iters"] = guidance_iters
                if noise_tape is not None:
                    ikw["noise_tape"] = noise_tape.for_clips(act) if hasattr(noise_tape, "for_clips") else noise_tape
                batch["inference_kwargs"] = ikw
                queued.append((cidx, act))
                if pipelined:
                    out = self.model.submit(**batch)
                    pending, act_prev = self.model.pending_latent(), act
                    if out is not None:
                        take(out)
                else:
                    take(self.model(**batch))
            if pipelined:
                for out in self.model.flush():
                    take(out)
        capi.require(not queued, "long-form synthesis: windows left in the pipeline")
        results = {ci: self._finish(state[ci], with_gt) for ci in mine}
        if gather and td.is_available() and td.is_initialized() and td.get_world_size() > 1:
            parts = [None] * td.get_world_size()
            slim = {ci: {k: v for k, v in r.items() if k != "latents"} for ci, r in results.items()}
            td.all_gather_object(parts, slim)
            results = {ci: r for part in parts for ci, r in part.items()}
        return results

    def _finish(self, st, with_gt):
        motion, facial, trans = st["so_far"]
        scale = self.target_fps // self.fps
        up = lambda m, f, t: (packing.upsample_motion(m, scale), packing.upsample_features(f, scale),
                              packing.upsample_features(t, scale)) if scale != 1 else (m, f, t)
        motion, facial, trans = up(motion, facial, trans)
        n_out = st["sample_len"] * scale
        cut = lambda t: t[0, :n_out].detach().cpu().numpy()
        result = dict(poses=cut(motion), expressions=cut(facial), trans=cut(trans), latents=st["latents"],
                      windows=list(zip(st["starts"], st["ends"])))
        if with_gt:
            gm, gf, gt_ = up(*st["gt_so_far"])
            result.update(gt_poses=cut(gm), gt_expressions=cut(gf), gt_trans=cut(gt_))
        return result

    @staticmethod
    def save(result, out_dir, raw_text=None):
        """The per-sample files of the tool (longform_synthesis.py:760-788): full_pred_motion.npz, full_gt_motion.npz when
        the run carried the ground truth, gt_text.txt when the transcript is given (audio / video are the caller's)."""
        os.makedirs(out_dir, exist_ok=True)
        packing.save_npz(os.path.join(out_dir, "full_pred_motion.npz"), result["poses"], result["expressions"], result["trans"])
        if "gt_poses" in result:
            packing.save_npz(os.path.join(out_dir, "full_gt_motion.npz"), result["gt_poses"], result["gt_expressions"],
                             result["gt_trans"])
        if raw_text is not None:
            with open(os.path.join(out_dir, "gt_text.txt"), "w", encoding="utf-8") as f:
                f.write(raw_text)
