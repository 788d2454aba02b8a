#!/usr/bin/env python3
"""Benchmark of the RAG-Gesture inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload guided|base|longform] [--batch B]
                    [--no-cobatch] [--no-pipeline] [--phases]

One "step" = one pass of the hot path over one batch of synthetic clips through the drop-in model: 4x VAE encode ->
conditioning precompute -> [retrieval -> batched DDIM inversion of the retrieved exemplars] -> 50-step DDIM with CFG
[+ insertion guidance] -> 4x VAE decode (150 SMPL-X frames per clip at 15 fps).  Inputs are resident
in HBM before the timed region.  The K timed steps are K complete batches: the guided workload goes through
`model.submit()` / `model.flush()` (the sampling loop of a batch shares its denoiser launches with the exemplar
inversion of a later batch; the pipeline fills and drains INSIDE the timed region), `--no-cobatch` = asynchronous
`model(**data)` calls, `--no-pipeline` = one synchronous `model(**data)` per step (also reported under `also`).
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), clips sharded across ranks (weak scaling, B clips per
rank), the ranks' results all-gathered once at the end of the loop.

Prints ONE JSON line (rank 0) with the contract fields plus
  `roofline`            dominant kernel of the headline workload, HIP-event timed in an instrumented step,
  `roofline_retrieval`  the DB sweep against the HBM roofline,
  `also`                the other single-GPU figures in the same run: base B=32 (BASELINE config 2), the headline
                        workload as synchronous forwards and in the fp32-equivalent mode (bf16x3 operands: same precision class as the reference),
                        long-form synthesis (config 5: 10 clips x 3 windows, window k of all clips in one forward),
  `cpu_baseline`        the CPU oracle (= a faithful port of the reference) on a bounded sample, rank 0 at N = 1 only:
                        one warm-up, median of 3, in the reference-faithful and the loop-invariant-hoisted form.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

torch = None   # imported in main(), after the rank-launch decision (the launcher parent never loads it)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GI = [0] * 25 + list(range(25))  # tools/visualize.py:74-95 "decreasing_till_25"
MFMA_BF16_PEAK = 2.5e15          # dense bf16 FLOP/s, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12                # B/s (spec), MI355X_MICROARCH.md


def make_re_dict(B, seed, device):
    """Synthetic retrieval result with the schema of RetrievalDatabase.forward (raggesture.py:860-884):
    2 exemplars per clip at latent spans retr (2,5),(6,8) -> query (1,4),(7,9)."""
    import numpy as np
    g = np.random.Generator(np.random.PCG64(seed))
    T, D = 43, 512

    def n(shape):
        return torch.from_numpy(g.standard_normal(size=shape).astype(np.float32)).to(device)

    mask = torch.ones(1, T, device=device)
    mask[:, [10, 21, 32]] = 0
    rs, qs, ls = [], [], []
    for _ in range(B):
        r, q, l = {}, {}, {}
        for qi, (r0, r1, q0, q1) in enumerate(((2, 5, 1, 4), (6, 8, 7, 9))):
            r[qi], q[qi] = (r0, r1), (q0, q1)
            lat = n((1, T, D))
            lat[:, [10, 21, 32]] = 0
            l[qi] = dict(retr_motion_latent=lat, retr_text=n((1, 150, 768)), retr_audio=n((1, 499, 768)),
                         retr_spkid=torch.full((1, 150), int(g.integers(0, 25)), dtype=torch.int64, device=device),
                         retr_motion_mask=mask.clone())
        rs.append(r), qs.append(q), ls.append(l)
    return dict(retr_startends=rs, query_startends=qs, retr_uncropped_latents=ls)


def launch_ranks(n, argv, dry=False):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh rank processes (one per GPU,
    RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set as torch.distributed.run would), relay rank 0's JSON line and
    return non-zero if any rank fails.  The parent never touches the GPU (and never exec()s): every rank is a
    child process that initialises HIP itself."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("GPU_MAX_HW_QUEUES", "16")     # read by the HIP runtime at its first call: in place before the rank starts
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE, stderr=None if r == 0 else subprocess.PIPE, text=True))
    rc, outs = 0, []
    for r, p in enumerate(procs):
        out, err = p.communicate()
        outs.append(out)
        if p.returncode != 0:
            rc = rc or p.returncode or 1
            sys.stderr.write("bench.py: rank %d exited with code %s\n%s\n" % (r, p.returncode, (err or "")[-2000:]))
    if dry:   # one line per rank: what each child saw
        print(json.dumps([json.loads(o.strip().splitlines()[-1]) for o in outs if o.strip()]))
    else:
        sys.stdout.write(outs[0])
    sys.stdout.flush()
    return rc


class Workload:
    """One benchmark configuration: model + resident synthetic inputs + step()."""

    streams = None     # the measured stream set of the first model of the process; later models take it (same topology for all)

    def __init__(self, rg, kind, B, dev, rank, db_size, precision="bf16", database=None, clips=10, windows=3, pipelined=True, cobatch=True):
        import collections
        self.rg, self.kind, self.B, self.dev, self.precision = rg, kind, B, dev, precision
        self._submitted, self._latency = collections.deque(), []
        # every random draw of a batch comes from this generator, inside submit(): the state in front of a submission is all
        # it takes to repeat that batch later (verify(): the last timed batches against synchronous forwards)
        self.noise = rg.pipeline.DeviceNoise(dev, seed=4242 + rank)
        self._done = collections.deque(maxlen=max(1, int(os.environ.get("RG_BENCH_VERIFY_BATCHES", "4"))))   # (one per batch lane)
        self.cfg = rg.synth.default_model_cfg(num_layers=8)
        self.vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
        self.guided = kind in ("guided", "longform")
        # guided / longform: the retrieval DB (discourse metadata, gesture labels, token features) is replicated on every GPU
        self.database = database
        if self.guided and database is None:
            self.database = rg.synth.SyntheticDataset(db_size, seed=2025, device=dev, feat_device=dev)
        self.model = rg.build_architecture(rg.synth.reference_style_model_cfg(self.cfg, self.vae_cfgs, with_retrieval=self.guided),
                                           database=self.database if self.guided else None, device=dev, precision=precision,
                                           lane_streams=Workload.streams, **json.loads(os.environ.get("RG_BENCH_MODEL_KWARGS", "{}")))
        if Workload.streams is None:
            Workload.streams = self.model.stream_set()
        self.model.load_state_dict(rg.synth.synth_full_state(0, self.cfg, self.vae_cfgs))
        self.model.eval()
        self.model.async_results = bool(pipelined)     # (long-form: run_many then pipelines the windows through submit())
        self.model.calibrate_lanes = True              # which streams, never how many: performance only (pipeline._make_streams)
        # guided workload: the sampling loop of batch n shares its denoiser launches with the inversion of batch n + 1
        # base workload: nothing to co-batch, but submit() lets whole batches alternate between model.base_lanes lanes
        self.cobatch = self.model.async_results and kind in ("guided", "base") and cobatch and precision == "bf16"
        if kind == "longform":
            # BASELINE config 5: tools/longform_synthesis.py's loop -- overlapping 150-frame windows, every window a guided
            # forward with `use_inversion + insertion_guidance + use_prev_latent` (longform_synthesis.py:389-403) and
            # retrieval_method = "llm" (rag/llm_retrieval.py:166-466) on CACHED LLM answers: the response cache is filled
            # by a deterministic stand-in for the API call during warm-up, the timed steps only hit it
            import tempfile
            self.n_clips, self.windows = clips, windows
            self.clips = [rg.synth.synth_longform_clip(5000 + 100 * rank + 10 * c, windows, device=dev) for c in range(clips)]
            self.audio = [rg.synth.synth_batch(1, seed=7000 + w, device=dev)["audio"] for w in range(windows)]
            self.feats = [[rg.synth.synth_query(300 + 10 * c + w)["text_features"].to(dev) for w in range(windows)] for c in range(clips)]
            self.llm_cache = rg.retrieval.LLMResponseCache(os.path.join(tempfile.mkdtemp(prefix="rg_llm_"), "llm_cache.json"),
                                                           call=rg.synth.synth_llm_answer)
            self.model.model.database.llm_output = self.llm_cache.get
            self.synth = rg.longform.LongformSynthesizer(self.model, overlap=15)
            self.frames_per_step = clips * 135 * windows
            return
        self.data = rg.synth.synth_batch(B, seed=1234 + rank, device=dev)
        if self.guided:  # per-clip discourse relations / prominence / BERT token features (3 relations per query)
            qs = [rg.synth.synth_query(1000 * rank + i) for i in range(B)]
            self.data["discourse"] = [q["discourse"] for q in qs]
            self.data["prominence"] = [q["prominence"] for q in qs]
            self.data["text_features"] = [q["text_features"].to(dev) for q in qs]
            self.data["speaker_ids"] = torch.tensor([[q["speaker_id"]] * 150 for q in qs], device=dev)
        self.trans0 = self.data["trans"].clone()
        self.frames_per_step = B * 150

    def step(self):
        """One pass of the hot path; returns the packed [B,150,268] result (None for longform)."""
        if self.kind == "longform":
            def features(ci, cidx, t0, t1, ann):     # the per-window callback of longform_synthesis.py:320-343
                text = " ".join(seg[1] for seg in ann["text_segments"][0])
                return dict(audio=self.audio[cidx], raw_word=[text], text_features=[self.feats[ci][cidx]])
            self._features = features
            self._done.append((self.noise.state(), None))
            self.last = self._longform_pass()
            return None
        d = dict(self.data)
        d["trans"] = self.trans0.clone()  # forward re-zeroes trans in place like the reference
        ev = torch.cuda.Event(enable_timing=True)      # submission time of this batch (per-batch latency, see pack())
        ev.record()
        self._submitted.append((ev, self.noise.state()))
        call = self.model.submit if self.cobatch else self.model
        return self.pack(call(**dict(d, retrieval_method="discourse", inference_kwargs=self._ikw())))

    def _ikw(self):
        ikw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1) if self.kind == "guided" else {}
        return dict(ikw, noise_tape=self.noise)

    def _longform_pass(self, **kw):
        return self.synth.run_many([dict(c, trans=c["trans"].clone()) for c in self.clips], self._features, use_inversion=True,
                                   insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1, retrieval_method="llm",
                                   shard=False, noise_tape=self.noise, **kw)

    @staticmethod
    def _cat(out):
        return torch.cat([out["pred_upper"], out["pred_lower"], out["pred_facepose"], out["pred_hands"],
                          out["pred_transl"], out["pred_exps"]], dim=-1)

    def verify(self):
        """OUTSIDE the timed region: the last batches the timed loop produced (four: one per batch lane) against ONE
        SYNCHRONOUS forward each of the same inputs and the same noise (the generator state saved in front of their
        submission) -- `torch.equal`, bit for bit.  What is timed is therefore a throughput of verified results: a race
        between lanes, a stale graph buffer or a mis-ordered hand-out shows up here (every batch has its own noise, so no
        two batches have equal results).  Long-form: the last pass against the sequential window loop."""
        import numpy as np
        torch.cuda.synchronize()
        recs = list(self._done)[-1:] if self.kind == "longform" else list(self._done)   # (long-form keeps its last pass only)
        self.drain()
        torch.cuda.synchronize()
        model = self.model
        mode, keep = (self.cobatch, model.async_results), self.noise.state()
        self.cobatch, model.async_results = False, False
        res = {"verified": bool(recs), "batches": len(recs), "against": "synchronous forward, same inputs and noise (torch.equal)"}
        try:
            for n, (state, got) in enumerate(recs):
                self.noise.set_state(state)
                if self.kind == "longform":
                    ref, got = self._longform_pass(pipelined=False), self.last
                    pairs = [("clip %d %s" % (c, k), np.asarray(ref[c][k]), np.asarray(got[c][k])) for c in sorted(ref)
                             for k in ("poses", "expressions", "trans")]
                    res["against"] = "sequential window loop (pipelined=False), same inputs and noise (array_equal)"
                else:
                    d = dict(self.data)
                    d["trans"] = self.trans0.clone()
                    ref = self._cat(model(**dict(d, retrieval_method="discourse", inference_kwargs=self._ikw())))
                    torch.cuda.synchronize()
                    pairs = [("packed", ref.cpu().numpy(), got.cpu().numpy())]
                for name, a, b in pairs:
                    if a.shape != b.shape or not np.array_equal(a, b):
                        res["verified"] = False
                        res["mismatching"] = res.get("mismatching", 0) + 1
                        if os.environ.get("RG_BENCH_DUMP_MISMATCH") and a.shape == b.shape:
                            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                            np.savez_compressed(os.path.join(ROOT, "gpurun_out", "bench_mismatch_%s_%d.npz" % (self.kind, n)), ref=a, got=b)
                        res.setdefault("first_mismatch", {"batch_from_end": len(recs) - 1 - n, "key": name,
                                                          "max_abs": float(np.nanmax(np.abs(a - b))) if a.shape == b.shape else None,
                                                          "elements": int((a != b).sum()) if a.shape == b.shape else None})
        finally:
            (self.cobatch, model.async_results) = mode
            self.noise.set_state(keep)
        return res

    def pack(self, out):
        """The packed [B,150,268] result of a finished (or, with asynchronous submission, queued) batch; None while the
        co-batched pipeline fills (submit() hands out the batch submitted one call earlier)."""
        if out is None:
            return None
        # asynchronous submission (model.async_results): the batch is only queued; its packed result is assembled on
        # the stream the batch ends on, so the caller's stream is free for the next batch's front end
        with torch.cuda.stream(out.get("done_stream") or torch.cuda.current_stream()):
            packed = self._cat(out)
            done = torch.cuda.Event(enable_timing=True)
            done.record()
        if self._submitted:     # results come back in submission order: device time from a batch's submission to its packed result
            ev, state = self._submitted.popleft()
            self._latency.append((ev, done))
            self._done.append((state, packed))
        return packed

    def latency_ms(self):
        """Median / max device time from the submission of a batch to its packed result over the batches seen since the
        last call (call after a device synchronisation)."""
        ms = sorted(a.elapsed_time(b) for a, b in self._latency)
        self._latency = []
        return (round(ms[len(ms) // 2], 2), round(ms[-1], 2)) if ms else (None, None)

    def drain(self):
        """Finish whatever the co-batched pipeline still holds (the last batch's sampling loop)."""
        return [self.pack(o) for o in self.model.flush()] if self.cobatch else []

    def prime(self):
        """Graph capture for every slot of the asynchronous pipeline (setup, like building the model): the W warm-up steps
        and the K timed steps then only replay."""
        if not getattr(self, "primed", False):
            # every (phase, slot) pair of the pipeline has a graph of its own (inversion alone while the pipeline fills,
            # co-batched chain, sampling alone at the drain; slots alternate call by call): fill-and-drain sequences of odd
            # and even length until no call captures anything new
            # (whole batches rotate over the lanes: a lane sees its second batch -- the co-batched chain -- only in a
            #  sequence longer than the number of lanes)
            lanes = self.rotation() if self.kind == "guided" else 2
            short, long_ = (2, 3) if lanes <= 2 else (lanes, 2 * lanes + 1)
            seen, quiet, n = -1, 0, short
            while quiet < 2 and self.kind != "longform":
                for _ in range(n if self.model.async_results else 1):
                    self.step()
                self.drain()
                n = short + long_ - n
                quiet = quiet + 1 if len(self.model._graphs) == seen else 0
                seen = len(self.model._graphs)
            while quiet < 2 and self.kind == "longform" and self.model.async_results and n < 12:
                # long-form: the windows go through submit() / flush(); passes until no pass captures a new graph
                self.step()
                n += 1
                quiet = quiet + 1 if len(self.model._graphs) == seen else 0
                seen = len(self.model._graphs)
            self.primed = True

    def rotation(self):
        """Lanes that whole batches of this workload rotate over (submit())."""
        m = self.model
        if not self.cobatch or m.cobatch_lanes != "batch":
            return 1
        return m.batch_lanes if self.guided else m.base_lanes

    def timed(self, steps, warmup, fence):
        self.prime()
        for _ in range(warmup):
            self.step()
        self.drain()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.drain()     # the timed region holds `steps` complete batches: the pipeline fills and drains inside it
        fence()
        return time.perf_counter() - t0

    def gemm_roofline(self, local_rank):
        """HIP events around every launch of the dominant kernel in eager steps of the SAME submission mode as the timed run
        (graph replays cannot hold events; the launches are the same), taken in the steady state: the pipeline is primed with
        two eager steps per batch lane before the events start, one step per batch lane is recorded.
        bf16 mode: rg_seq_kernel (the whole denoiser forward of up to 256 sequences in one launch; variant 3), or the bf16-A
        rg_gemm kernels when the launch chain is selected (variant 1); fp32 mode: the bf16x3 GEMMs of the launch chain
        (variant 2, 3 MFMAs per product: priced at a third of the bf16 peak)."""
        rg, model = self.rg, self.model
        h = rg.capi.get_handle(local_rank)
        seq = self.precision == "bf16" and model.model.weights.seq_streams is not None and model.session_options.get("engine") != "chain"
        variant = 2 if self.precision != "bf16" else (3 if seq else 1)
        peak = MFMA_BF16_PEAK if variant != 2 else MFMA_BF16_PEAK / 3

        rot = max(2, self.rotation())

        def events(nprime=2 * rot, nprof=rot):
            self.drain()
            model.use_graphs = False
            for _ in range(nprime):
                self.step()
            torch.cuda.synchronize()
            h.lib.rg_profile_begin(h._h)
            for _ in range(nprof):
                self.step()
            model.use_graphs = True
            n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            h.lib.rg_profile_end(h._h, variant, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl))
            self.drain()
            return n.value, ms.value, fl.value

        n, ms, fl = events()
        ach = fl / (ms * 1e-3) if ms > 0 else 0.0
        duo_forms = {(int(s_.sq.duo), int(s_.sq.args.pairs)) for k, s_ in model._sessions.items() if s_.sq is not None and k[1] == "cobatch"} if seq else set()
        duo = (1, 1) in duo_forms and self.cobatch           # two clips per workgroup: conditional pair, then the classifier-free pair
        duo_wide = (1, 0) in duo_forms and self.cobatch and not duo      # the two pairs in workgroups of their own
        kernel = {3: ("rg_seq2_kernel (a whole denoiser forward per workgroup -- embedding, 8 decoder layers, head -- of TWO conditional "
                      "sequences, then of their two classifier-free twins: every streamed weight fragment feeds 6 MFMAs; bf16 MFMA, fp32 "
                      "accumulate; weights streamed by LDS-DMA)" if duo else
                      "rg_seq2_kernel (a whole denoiser forward per workgroup -- embedding, 8 decoder layers, head -- of TWO sequences "
                      "of a kind, conditional or classifier-free: every streamed weight fragment feeds 6 MFMAs; bf16 MFMA, fp32 "
                      "accumulate; weights streamed by LDS-DMA)" if duo_wide else
                      "rg_seq_kernel (a whole denoiser forward per workgroup -- embedding, 8 decoder layers, head -- of one sequence, or of "
                      "a clip's conditional sequence and then its classifier-free twin; bf16 MFMA, fp32 accumulate; weights streamed by "
                      "LDS-DMA)"),
                  1: "rg_gemm bf16-A kernels (gemm_dma_kernel<true,..>, gemm_bf16_big_kernel; bf16 MFMA, fp32 accumulate)",
                  2: "rg_gemm bf16x3 kernels (fp32-equivalent products: hi*hi + hi*lo + lo*hi)"}[variant]
        r = {"bound": "mfma", "kernel": kernel,
             "achieved": round(ach / 1e12, 3), "peak": round(peak / 1e12, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 5),
             "launches": n, "avg_launch_us": round(ms * 1e3 / max(1, n), 2), "flops_per_launch_avg": round(fl / max(1, n)),
             "lanes": self.rotation(), "flops_per_step": round(fl / rot)}      # (`rot` recorded steps)
        if seq:
            # a launch holds one workgroup per sequence and a workgroup owns a CU (155 KiB of LDS): a lane of 2 x (16 + 48)
            # sequences runs on 128 of the 256 CUs and the other lane's launch on the rest, at the same time
            per_clip = 2.0 * 43 * 512 * 512 * (130 + 82) + 8 * 7 * 2.0 * 43 * 32 * 32 * 16   # one conditional + one classifier-free sequence
            pairs = [s.sq.args.pairs for k, s in model._sessions.items() if s.sq is not None and k[1] == "cobatch"]
            paired = bool(pairs) and all(pairs) and self.cobatch
            seqs = 2 * round(fl / max(1, n) / per_clip)
            cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
            r["sequences_per_launch"] = seqs
            r["workgroups_per_launch"] = seqs // 4 if duo else (seqs // 2 if (paired or duo_wide) else seqs)
            r["launch_form"] = "duo_pairs" if duo else ("duo" if duo_wide else ("pairs" if paired else "one_per_sequence"))
            # a workgroup owns a compute unit for the whole launch (155 KiB of LDS): the share of the chip a launch can use
            r["cu_share"] = round(min(1.0, r["workgroups_per_launch"] / cus), 4)
            r["frac_of_occupied_cus"] = round(ach / (peak * r["cu_share"]), 5) if r["cu_share"] else None
            if duo_wide:
                # half of the workgroups run classifier-free pairs and leave early: the ratio of the two kinds' pass times comes from
                # the in-kernel stamps of the diagnostic build (profiles/dbg/seq2_stamps.py writes the JSON) -- NOT measured in
                # this run, and only used while the JSON belongs to the kernel source that is running; the CU time a launch
                # really holds is (1 + ratio) / 2 of workgroups x launch time
                try:
                    with open(os.path.join(ROOT, "profiles", STAMPS_JSON)) as f:
                        sj = json.load(f)
                    if sj.get("kernel_source_sha256") == _sha256(os.path.join(ROOT, "rag-gesture_amd", "csrc", "rg_seq2.hip")):
                        ratio = sj["classifier_free_pass_us"] / sj["conditional_pass_us"]
                        r["cu_time_held_share"] = round(r["cu_share"] * (1 + ratio) / 2, 4)
                        r["frac_of_held_cu_time"] = round(ach / (peak * r["cu_time_held_share"]), 5)
                        r["stamps_source"] = "profiles/" + STAMPS_JSON
                    else:
                        r["stamps_source"] = "profiles/%s is of another build of rg_seq2.hip: held-CU-time figures omitted" % STAMPS_JSON
                except (OSError, ValueError, KeyError, ZeroDivisionError):
                    pass
            r["note"] = ("per launch; a launch occupies one CU per workgroup (%s), so %d concurrent lanes share the chip: `frac` prices "
                         "one lane's launch against the WHOLE chip's peak, `frac_of_occupied_cus` against the peak of the CUs it "
                         "holds; the chip-level rate is the sum over the lanes' concurrent launches (`whole_step`)"
                         % ("one workgroup per two clips: their conditional sequences, then their classifier-free twins" if duo else
                            "one workgroup per two sequences of a kind; the classifier-free half of the workgroups leaves after 0.63 of "
                            "the launch, which `frac_of_occupied_cus` does not credit" if duo_wide else
                            "one workgroup per clip: conditional sequence, then its classifier-free twin" if paired
                            else "one workgroup per sequence", self.rotation()))
        return r


STAMPS_JSON = "r06_seq2_stamps.json"
PMC_JSON = {"duo": "r06_pmc_seq2_wide.json", "duo_pairs": "r06_pmc_seq2_pairs.json", "pairs": "r04s_pmc_seq_pairs.json", None: "r04_pmc_seq.json"}


def _sha256(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def engines(wl):
    """Which implementation of each stack the timed model resolved to (a checkpoint whose hyper-parameters the fused kernels are
    not specialised for falls to the per-op launch chains; the line says which one ran)."""
    m = wl.model
    sess = [s_ for s_ in m._sessions.values()]
    den = sorted({("seq2" if s_.sq.duo else "seq") + ("+pairs" if s_.sq.args.pairs else "") if s_.sq is not None else "chain" for s_ in sess})
    vaes = list(m.model.gesture_rep_encoder.vaes.values())
    return {"denoiser": "|".join(den) if den else None,
            "vae_encoder": "|".join(sorted({"fused" if getattr(v, "venc", None) is not None else "chain" for v in vaes})),
            "vae_decoder": "|".join(sorted({"fused" if getattr(v, "vdec", None) is not None else "chain" for v in vaes}))}


def run_steps(wl, n, dist=None, world=1, sync=None):
    """n batches through the pipeline; then, like the reference's multi-GPU test loop (mogen/apis/test.py:129-160: every
    rank runs its shard of the clips, `collect_results_gpu` gathers ONCE at the end), one all-gather of this rank's
    packed results -- the only collective on the path (RCCL over xGMI).  Inside the timed region.
    wl: anything with step() -> packed [B, 150, 268] result or None and drain() -> list of the same (the co-batched
    pipeline hands results out late); sync(): device synchronisation before the collective (the results sit on the lanes'
    streams); dist: torch.distributed or None."""
    import torch
    packed = [p for p in (wl.step() for _ in range(n)) if p is not None]
    packed += [p for p in wl.drain() if p is not None]   # co-batched pipeline: the last batches' sampling
    if dist is None or not packed:
        return packed
    if sync is not None:
        sync()
    mine = torch.cat(packed, dim=0)
    gathered = torch.empty(world * mine.shape[0], mine.shape[1], mine.shape[2], device=mine.device, dtype=mine.dtype)
    dist.all_gather_into_tensor(gathered, mine)
    return gathered


def retrieval_roofline(rg, wl):
    """The DB sweep (rg_discourse_scores_batched: every query relation of the batch against every DB entry) against the
    HBM roofline: algorithmic bytes = what one pass must read (the integer-coded CSR, once per query relation) and
    write (a float64 score and an int32 relation index per entry and query), over its HIP-event time."""
    idx = wl.database_index()
    queries = []
    spks = [int(v) for v in wl.data["speaker_ids"][:, 0].tolist()]
    for b in range(wl.B):
        queries += rg.retrieval.discourse_queries(wl.data["discourse"][b], wl.data["prominence"][b], spks[b])
    for _ in range(2):
        idx.collect(idx.sweep_async(queries))
    # the launches alone, buffers and query parameters resident (as everything else in the timed region)
    bufs = idx.sweep_buffers(queries)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        pend = idx.sweep_launch(bufs)
    e1.record()
    torch.cuda.synchronize()
    idx.collect(pend)
    ms = e0.elapsed_time(e1) / reps
    db_bytes = sum(t.numel() * t.element_size() for t in (idx.spk, idx.rel_off, idx.rel_sense, idx.rel_conn, idx.rel_prom))
    Q, n = len(queries), idx.n
    alg = Q * (db_bytes + n * 12)
    ach = alg / (ms * 1e-3)
    return {"bound": "hbm", "kernel": "rg_discourse_select_fused: sweep + top-score selection of one batch of query relations in two launches",
            "achieved": round(ach / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(ach / HBM_PEAK, 5),
            "bytes_per_sweep": alg, "queries": Q, "db_entries": n, "sweep_ms": round(ms, 4),
            "note": "algorithmic bytes as SURVEY 8d prices the sweep (CSR once per query relation + 12 B per (query, entry)); the fused "
                    "form keeps the scores in registers and reads the %.1f MB CSR twice per relation from L2" % (db_bytes / 1e6)}


def cpu_baseline(rg, wl, guided):
    """the fp32 CPU port on the host cores, one warm-up, median of 3, in the reference-faithful
    form (K/V projections recomputed on every denoiser call, as the reference does) and with that loop-invariant part
    hoisted.  Sample: ONE clip of the headline workload (150 frames)."""
    from oracle import pipeline as opipe, diffusion as odf, retrieval as oret, denoiser as od
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    avail = cores
    cores = max(1, min(cores, 32))  # torch's CPU matmuls on [43, 512] operands stop scaling well below that (one run with
    # every core of a 200+ core host did not finish one clip in 40 minutes: oversubscribed OpenMP teams on tiny GEMMs)
    torch.set_num_threads(cores)
    cfg, vae_cfgs = wl.cfg, wl.vae_cfgs
    P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
    cdata0 = rg.synth.synth_batch(1, seed=1234)
    ckw = {}
    if guided:
        cpu_db = oret.build_db_dicts([dict(r, text_feature=r["text_feature"].cpu()) for r in wl.database.retrieval_samples])
        cpu_ds = rg.synth.SyntheticDataset(0)
        q0 = rg.synth.synth_query(0)
        ccond = dict(text_features=[q0["text_features"]], discourse=[q0["discourse"]], prominence=[q0["prominence"]],
                     speaker_ids=torch.tensor([[q0["speaker_id"]] * 150]))
        ckw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)

    def one(full=True):
        cdata = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in cdata0.items()}
        tape = rg.synth.NoiseTape(1)
        od.hoist_clear()
        t = time.perf_counter()
        with torch.no_grad():
            if guided and full:  # sweep over the same DB + exemplar encode + placement, as the reference does per clip
                cre = lambda tp: oret.database_forward(P, vae_cfgs, cpu_db, cpu_ds, ccond, ["bench_query"], tp)
                opipe.motion_diffusion_forward(P, cfg, vae_cfgs, odf.SpacedSchedule(), cdata, tape, re_dict=cre, **ckw)
            else:
                opipe.motion_diffusion_forward(P, cfg, vae_cfgs, odf.SpacedSchedule(), cdata, tape, re_dict=None)
        return time.perf_counter() - t

    out = {}
    one(full=False)   # warm-up: one base clip (thread pool, allocator, first-touch of the weights)

    for mode in ("faithful", "hoisted"):
        od.OPTS["hoist"] = mode == "hoisted"
        ts = sorted(one() for _ in range(3))
        out[mode] = ts[1]
    od.OPTS["hoist"] = False
    od.hoist_clear()
    return {"value": round(150.0 / out["faithful"], 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "host_cores_available": avail,
            "hoisted": {"value": round(150.0 / out["hoisted"], 2), "seconds_per_clip": round(out["hoisted"], 2)},
            "seconds_per_clip": round(out["faithful"], 2),
            "sample": "1 clip (150 frames) of the headline workload (%s), torch fp32 on %d of the host's %d hardware threads "
                      "(BASELINE.md section 3 says all physical cores; capped at 32 because torch's CPU matmuls on [43, 512] operands "
                      "stop scaling below that -- a run on all 200+ threads of a box did not finish one clip in 40 minutes); 1 warm-up, "
                      "median of 3 per mode; value = reference-faithful (recomputes the cross-attention K/V side every denoiser call), "
                      "hoisted = that loop-invariant part computed once" % ("guided" if guided else "base", cores, avail)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher check (no GPU): every rank prints the environment it was started with and exits")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["guided", "base", "longform"], default="guided")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: 16 guided, 32 base, 10 longform)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="synchronous forwards (results valid on the caller's stream) instead of asynchronous submission")
    ap.add_argument("--no-cobatch", action="store_true",
                    help="asynchronous submission without sharing launches between consecutive batches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the additional single-GPU records (base, fp32 mode, longform)")
    ap.add_argument("--db-size", type=int, default=32768, help="retrieval DB entries (guided workload)")
    ap.add_argument("--phases", action="store_true",
                    help="after the timed run, one extra step with device syncs at phase boundaries; prints the breakdown to stderr")
    args = ap.parse_args()

    # ---- rank launch: decided before anything touches the GPU
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], dry=args.dry_launch))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or unset WORLD_SIZE "
                         "to let bench.py start the ranks itself)" % (args.gpus, world, args.gpus))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # (torchrun-started ranks: still before torch is imported)
    if args.dry_launch:
        print(json.dumps(dict(rank=rank, local_rank=local_rank, world=world, master_addr=os.environ.get("MASTER_ADDR"),
                              master_port=os.environ.get("MASTER_PORT"), pid=os.getpid(),
                              hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"), torch_imported="torch" in sys.modules,
                              ipc_legacy=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))))
        return
    global torch
    import torch
    # (before the first HIP call: importing the package sets GPU_MAX_HW_QUEUES for the runtime unless the user has)
    rg = importlib.import_module("rag-gesture_amd")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("RG_BENCH_FORCE_DIST") == "1":   # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:      # (the switch without a launcher: a world of one)
            for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", str(local_rank)), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k_, v_)
        dist.init_process_group(backend="nccl", device_id=dev)

    Workload.database_index = lambda self: self.model.model.database.index
    kind = args.workload
    B = args.batch or {"guided": 16, "base": 32, "longform": 10}[kind]
    wl = Workload(rg, kind, B, dev, rank, args.db_size, clips=B, pipelined=not args.no_pipeline, cobatch=not args.no_cobatch)
    guided = kind == "guided"

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    wl.prime()
    run_steps(wl, args.warmup, dist, world, torch.cuda.synchronize)
    fence()
    wl.latency_ms()          # (forget the warm-up batches)
    t0 = time.perf_counter()
    run_steps(wl, args.steps, dist, world, torch.cuda.synchronize)     # the timed region holds exactly `steps` complete batches
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * wl.frames_per_step * args.steps / dt
    lat_med, lat_max = wl.latency_ms()
    # ---- what was timed is checked (outside the clock): the last batches of the timed loop against synchronous forwards
    verified = wl.verify()
    all_ok = verified["verified"]
    if dist is not None:
        t = torch.tensor([1.0 if all_ok else 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        all_ok = bool(t.item() == 1.0)
        verified = dict(verified, all_ranks=all_ok)

    if args.phases and rank == 0 and kind != "longform":
        # phase walls need device syncs at phase boundaries, which serialise concurrent lanes: the breakdown is taken
        # on a single lane (one warm-up step captures its graphs), the timed run above used model.lanes lanes
        model = wl.model
        wl.drain()
        lanes_prod, model.lanes = model.lanes, 1
        mode_prod, wl.cobatch, model.async_results = (wl.cobatch, model.async_results), False, False   # synchronous forwards
        wl.step()
        model.profile_phases, model.phase_ms = True, {}
        if guided:
            model.model.database.phase_ms = model.phase_ms
        wl.step()
        torch.cuda.synchronize()
        model.profile_phases = False
        if guided:
            model.model.database.phase_ms = None
        model.lanes = lanes_prod
        wl.cobatch, model.async_results = mode_prod
        print("phase breakdown (ms, one synchronised single-lane step; the timed run used %d lanes): " % lanes_prod +
              ", ".join("%s %.1f" % kv for kv in model.phase_ms.items()), file=sys.stderr)

    # ---- rank-0-only records (instrumented / additional steps never enter a collective)
    roofline = roof_retr = None
    also = {}
    steady = None
    if rank == 0 and world == 1 and wl.cobatch:
        # Informational, never `value`: the same K submissions with the pipeline already full when the clock starts and
        # still full when it stops (K results come out, the two pending batches are drained after the clock) -- `value`
        # above pays the pipeline's fill and drain inside its K steps.
        for _ in range(2 * wl.rotation()):
            wl.step()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        torch.cuda.synchronize()
        steady = round((time.perf_counter() - ts) / args.steps * 1e3, 2)
        wl.drain()
        torch.cuda.synchronize()
        wl.latency_ms()
    if rank == 0:
        if kind != "longform":
            roofline = wl.gemm_roofline(local_rank)
            # HBM-side bytes per launch from the committed PMC passes over the same kernels
            # (profiles/r01e_pmc_gemm_traffic.txt explains how they were collected and corrected); null if absent
            roofline["traffic"] = None
            # the same kernel over the whole timed region: its algorithmic FLOPs per step / the step time (all idle time,
            # the front end, the decode and the pipeline's fill and drain included) -- what the chip did end to end
            roofline["whole_step"] = {"achieved": round(roofline["flops_per_step"] / (ms_per_step * 1e-3) / 1e12, 1), "unit": "TFLOP/s",
                                      "frac": round(roofline["flops_per_step"] / (ms_per_step * 1e-3) / (roofline["peak"] * 1e12), 4)}
            try:
                # HBM-side bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE x2 gfx950
                # correction + WRITE_SIZE; profiles/pmc_seq.py, pmc_seq_summarize.py; round 1-2: the rg_gemm shapes)
                if roofline.get("workgroups_per_launch") is not None:
                    # (r04_pmc_seq: 128 workgroups, one per sequence; r04s_pmc_seq_pairs: 64 workgroups, one per clip -- the twin's
                    #  pass streams the shared weights through every L2 a second time)
                    # (r05B_pmc_seq2_pairs: 32 workgroups, two clips each -- rg_seq2_kernel with eight batch lanes;
                    #  r05D_pmc_seq2_wide: 64 workgroups, two sequences of a kind each -- what the default four lanes launch, the
                    #  last build; r05B_*: the build with the register path, before the conditions' A fragments lost their
                    #  unused low-order halves (48 MB more traffic and algorithmic bytes per launch).  The one-sequence forms' files are
                    #  round-4 passes of rg_seq_kernel BEFORE it got that path: only their traffic figures still apply)
                    form = roofline.get("launch_form")
                    src = PMC_JSON.get(form, PMC_JSON[None])
                    with open(os.path.join(ROOT, "profiles", src)) as f:
                        pm = json.load(f)
                    # these are NOT measured in this run: they are read from the committed rocprofv3 --pmc passes over the same
                    # launch form, one launch alone on the chip (profiles/pmc_seq.py) -- and only while those passes belong to the
                    # kernel source that is running (the JSON records the source's hash; a stale file says so and
                    # yields no utilisation figure)
                    ksrc = "rg_seq2.hip" if form in ("duo", "duo_pairs") else "rg_seq.hip"
                    fresh = pm.get("kernel_source_sha256") == _sha256(os.path.join(ROOT, "rag-gesture_amd", "csrc", ksrc))
                    roofline["pmc_source"] = "profiles/" + src
                    roofline["pmc_stale"] = not fresh
                    roofline["traffic_algorithmic"] = round(pm["algorithmic_hbm_bytes"])
                    if fresh:
                        roofline["traffic"] = round(pm["fetch_bytes"] + pm["write_bytes"])
                        roofline["mfma_utilisation_pmc"] = round(pm["mfma_utilisation"], 4)      # (of the whole chip's MFMA cycles, one launch alone)
                        if roofline.get("cu_share"):
                            roofline["mfma_utilisation_pmc_of_occupied_cus"] = round(pm["mfma_utilisation"] / roofline["cu_share"], 4)
                        if roofline.get("cu_time_held_share"):
                            roofline["mfma_utilisation_pmc_of_held_cu_time"] = round(pm["mfma_utilisation"] / roofline["cu_time_held_share"], 4)
                else:
                    with open(os.path.join(ROOT, "profiles", "r02m_pmc_gemm_traffic.json")) as f:
                        rows = json.load(f)
                    roofline["traffic"] = round(sum(r["fetch_bytes"] + r["write_bytes"] for r in rows) / len(rows))
            except (OSError, ValueError, KeyError, ZeroDivisionError, IndexError):
                pass
        if guided:
            roof_retr = retrieval_roofline(rg, wl)
        if world == 1 and not args.no_also and guided:
            def record(w, steps=6, warmup=1, roof=True):
                d = w.timed(steps, warmup, torch.cuda.synchronize)
                r = {"value": round(w.frames_per_step * steps / d, 1), "unit": "frames/s", "ms_per_step": round(d / steps * 1e3, 2),
                     "steps": steps, "warmup": warmup, "dtype": "bf16" if w.precision == "bf16" else "bf16x3 (fp32-equivalent)"}
                r["verified"] = w.verify()
                r["stream_topology"] = w.model.lane_report
                if w.kind != "longform" and roof:
                    r["roofline"] = w.gemm_roofline(local_rank)
                return r
            also["base_B32"] = dict(record(Workload(rg, "base", 32, dev, rank, args.db_size), steps=16, warmup=4),
                                    workload="base diffusion len150 DDIM-50 (no guidance), 32 clips per batch (BASELINE config 2), "
                                             "whole batches alternating between %d lanes through submit()" % wl.model.base_lanes)
            if wl.model.async_results:
                # the same model (same calibrated lane streams) switched to one synchronous forward per batch
                wl.drain()
                mode, wl.cobatch, wl.model.async_results, wl.primed = (wl.cobatch, wl.model.async_results), False, False, False
                r = record(wl, roof=False)
                wl.drain()
                (wl.cobatch, wl.model.async_results), wl.primed = mode, False
                also["guided_B16_synchronous"] = dict(r, workload="the headline workload as one synchronous forward per batch "
                                                      "(results valid on the caller's stream at return, like the reference's tools)")
            if B == 16:
                # the same pipeline with 32 clips per step on ONE lane: every rg_seq launch then holds 2 x (32 + 96) = 256
                # sequences = one per CU, which is what `roofline.frac` means for a launch that fills the chip
                w32 = Workload(rg, "guided", 32, dev, rank, args.db_size, database=wl.database)
                w32.model.lanes = w32.model.batch_lanes = 1
                also["guided_B32_one_lane"] = dict(record(w32, steps=8, warmup=2),
                                                   workload="the headline workload with 32 clips per step on one lane: 256 sequences per launch")
                del w32
            also["guided_B16_fp32mode"] = dict(
                record(Workload(rg, "guided", B, dev, rank, args.db_size, precision="fp32", database=wl.database)),
                workload="the headline workload with bf16x3 split operands (~fp32 products): same-precision figure")
            lw = Workload(rg, "longform", 10, dev, rank, args.db_size, database=wl.database, clips=10)
            also["longform_10clips"] = dict(
                record(lw, steps=4, roof=False),
                workload="long-form synthesis as tools/longform_synthesis.py runs it (BASELINE config 5 on one GPU): 10 clips x %d "
                         "overlapping 150-frame windows, every window with retrieval_method='llm' over the replicated DB on "
                         "cached LLM answers (LLMResponseCache), use_inversion + insertion_guidance + use_prev_latent, window k of "
                         "all clips in one forward, 6D overlap blend, 30 fps; frames = model frames at 15 fps" % lw.windows,
                llm_cache={"hits": lw.llm_cache.hits, "misses": lw.llm_cache.misses})
            del lw
            # the same loop over 32 clips at once: a pass costs about the same (a launch takes ~1 ms for any number of
            # sequences up to one per CU), so the frames per second follow the clip count
            lw = Workload(rg, "longform", 32, dev, rank, args.db_size, database=wl.database, clips=32)
            also["longform_32clips"] = dict(record(lw, steps=3, roof=False), workload="the same with 32 clips per pass",
                                            llm_cache={"hits": lw.llm_cache.hits, "misses": lw.llm_cache.misses})
            del lw
    if dist is not None:
        dist.barrier()

    # ---- CPU baseline: the oracle (faithful port of the reference) on a bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and kind != "longform":
        cpu = cpu_baseline(rg, wl, guided)

    if rank == 0:
        names = {"guided": ("guided discourse config: discourse retrieval over a replicated %d-entry DB, use_inversion + "
                            "insertion_guidance decreasing_till_25, <=3 exemplars/clip, len150 DDIM-50" % args.db_size),
                 "base": "base diffusion len150 DDIM-50 (no guidance)",
                 "longform": "long-form synthesis (tools/longform_synthesis.py loop): %d clips x 3 overlapping 150-frame windows per GPU, llm retrieval on cached answers + inversion + insertion guidance + prev-latent per window" % B}
        line = {
            "metric": "SMPL-X frames/sec, len150 DDIM-50 + insertion guidance; 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if dist is not None else None, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "batch_latency_ms": {"median": lat_med, "max": lat_max,
                                 "note": "device time from a batch's submission to its packed result; the co-batched pipeline "
                                         "hands a batch out one rotation of the batch lanes later (its sampling shares launches with "
                                         "the next batch's inversion on its lane), so latency > lanes x ms_per_step while "
                                         "throughput = 1 / ms_per_step"},
            "steady_state_ms_per_step": steady,   # informational (pipeline full at both ends of the clock); never `value`
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": names[kind], "engines": engines(wl),
                       "clips_per_gpu": B, "global_batch": world * B, "frames_per_clip": 150, "ddim_steps": 50,
                       "denoiser": "8 layers x 512, CFG x2 rows", "vae": "all_encoder, 8 layers, synthetic hparams",
                       "weights": "random-init at config shapes", "parallelism": "clip-sharded x%d" % world,
                       "submission": (("asynchronous, submit(): whole batches alternate between %d lanes (one 50-launch chain "
                                       "each); every batch completes inside the timed region" % wl.model.base_lanes)
                                      if wl.cobatch and kind == "base" else
                                      ("asynchronous, co-batched: whole batches rotate over %d lanes; the sampling loop of batch n and the "
                                       "exemplar inversion of batch n+%d advance in the same denoiser launches (sampler.cobatched_loop), "
                                       "the front ends of the batches between them run beside the lanes' chains; the pipeline fills and "
                                       "drains inside the timed region, which holds exactly `steps` complete batches"
                                       % (wl.rotation(), wl.rotation())) if wl.cobatch else
                                      ("asynchronous, %d slots: the front end (conditions, VAE encodes, retrieval) of batch n+1 and "
                                       "the decode of batch n run beside the inversion -> sampling chain; every batch completes "
                                       "inside the timed region" % wl.model.slots)) if wl.model.async_results
                       else "synchronous forwards"},
            "verified": all_ok, "verification": verified,
            "stream_topology": wl.model.lane_report,
            "roofline": roofline, "roofline_retrieval": roof_retr, "also": also or None, "cpu_baseline": cpu,
        }
        all_ok = all_ok and all(v.get("verified", {}).get("verified", True) for v in (also or {}).values())
        line["verified"] = all_ok
        # (once more inside the objects a harness that keeps only the contract's keys still carries: VERDICT r05)
        line["config"]["verified"], line["config"]["steady_state_ms_per_step"] = all_ok, steady
        if isinstance(roofline, dict):
            roofline["verified"], roofline["steady_state_ms_per_step"] = all_ok, steady
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio on stdout: everything it has to say is flushed out first, so that the
        # JSON is the LAST line of stdout
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)
    if not all_ok:
        sys.stderr.write("bench.py: a timed batch does not equal its synchronous forward -- the figure above is not a result\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
