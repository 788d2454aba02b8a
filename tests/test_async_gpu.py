"""Asynchronous submission (MotionDiffusion(async_results=True)): batches queued back to back, the host never waiting, must
give the same bits as one synchronous forward per batch -- the front end of batch n+1 runs beside the chain of batch n on
other sessions / graph buffers (slots), and every tensor that crosses streams is protected from the caching allocator."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
KEYS = ("pred_upper", "pred_lower", "pred_facepose", "pred_hands", "pred_transl", "pred_exps", "prev_latentout")
GI = [2] * 25 + [0] * 25


@pytest.fixture(scope="module")
def rg():
    return importlib.import_module("rag-gesture_amd")


def _where(model):
    """The schedule a comparison ran under (part of every failure message: a mismatch must be sizeable from the log)."""
    return dict(lane_streams=len(model._lane_streams), search_stream=model._search_stream is not None,
                topology=model.lane_report, graphs=len(model._graphs), cross_stream_waits=model.graph_cross_stream_waits,
                use_graphs=model.use_graphs)


def _same(got, want, model, tag):
    """Bit identity of two result lists (dicts of tensors or lists of numpy arrays), reporting the first mismatch in full."""
    import numpy as np
    assert len(got) == len(want), (tag, len(got), len(want))
    for i, (a, b) in enumerate(zip(got, want)):
        items = [(k, a[k], b[k]) for k in b] if isinstance(b, dict) else [(j, x, y) for j, (x, y) in enumerate(zip(a, b))]
        for k, x, y in items:
            x, y = (t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t) for t in (x, y))
            assert x.shape == y.shape, (tag, i, k, x.shape, y.shape)
            if not np.array_equal(x, y):
                d = np.abs(x.astype(np.float64) - y.astype(np.float64))
                try:       # keep the evidence: where a mismatch sits tells which kernel produced it
                    import os
                    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
                    os.makedirs(out, exist_ok=True)
                    np.savez_compressed(os.path.join(out, "mismatch_%s_b%d_%s.npz" % ("".join(c if c.isalnum() else "_" for c in tag)[:40], i, k)),
                                        got=x, want=y)
                except OSError:
                    pass
                raise AssertionError("%s: batch %d key %s differs: max abs %.3e, %d of %d elements, first at %s; schedule %s"
                                     % (tag, i, k, np.nanmax(d), int((x != y).sum()), x.size,
                                        np.argwhere(x != y)[0].tolist(), _where(model)))


class Jitter:
    """Random device-side delays (torch.cuda._sleep, 0-3 ms) in front of every graph use and every tail: moves the lanes
    against each other so that a missing ordering between streams shows up as a bit mismatch instead of once a week."""

    def __init__(self, seed, max_cycles=6_000_000):
        import random
        self.rng, self.max_cycles, self.calls = random.Random(seed), max_cycles, 0

    def __call__(self, stream, tag):
        self.calls += 1
        if self.rng.random() < 0.5:
            with torch.cuda.stream(stream):
                torch.cuda._sleep(self.rng.randrange(1, self.max_cycles))


def _batches(rg, B, n, dev):
    out = []
    for i in range(n):
        d = rg.synth.synth_batch(B, seed=900 + i, device=dev)
        qs = [rg.synth.synth_query(50 * i + j) for j in range(B)]
        d["discourse"] = [q["discourse"] for q in qs]
        d["prominence"] = [q["prominence"] for q in qs]
        d["text_features"] = [q["text_features"].to(dev) for q in qs]
        d["speaker_ids"] = torch.tensor([[q["speaker_id"]] * 150 for q in qs], device=dev)
        out.append(d)
    return out


@pytest.mark.parametrize("guided", [True, False])
def test_async_pipeline_equals_synchronous_forwards(rg, guided):
    dev = torch.device("cuda", 0)
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    db = rg.synth.SyntheticDataset(512, seed=11, device=dev, feat_device=dev) if guided else None
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=guided), database=db, device=dev)
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    B, N = 4, 5
    batches = _batches(rg, B, N, dev)
    ikw = lambda i: dict(dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1) if guided else {},
                         noise_tape=rg.synth.NoiseTape(3000 + i))

    def run(i):
        d = dict(batches[i])
        d["trans"] = batches[i]["trans"].clone()
        return model(**dict(d, retrieval_method="discourse", inference_kwargs=ikw(i)))

    ref = []
    for i in range(N):
        out = run(i)
        torch.cuda.synchronize()
        ref.append({k: out[k].clone() for k in KEYS})
    model.async_results = True
    consumer = torch.cuda.Stream()
    got = []
    for rep in range(2):                    # second pass: every slot's graphs exist, nothing synchronises the device
        got = []
        for i in range(N):
            out = run(i)
            assert "done_event" in out
            if i % 2:
                with torch.cuda.stream(consumer):           # a consumer stream of its own: waits for the completion event
                    model.wait_results(out)
                    got.append({k: out[k].clone() for k in KEYS})
            else:
                with torch.cuda.stream(out["done_stream"]):  # or the stream the batch ends on: ordered, no wait
                    got.append({k: out[k].clone() for k in KEYS})
            del out                          # results and temporaries die on the host while the lanes are a batch behind
            junk = [torch.full((B, 43, 512), float(i), device=dev) for _ in range(8)]   # reuse freed blocks on the caller's stream
            del junk
    torch.cuda.synchronize()
    _same(got, ref, model, "async forward()")
    # synchronous mode again: no event, tensors valid on the caller's stream
    model.async_results = False
    out = run(0)
    assert "done_event" not in out
    assert torch.equal(out["pred_upper"], ref[0]["pred_upper"])


@pytest.mark.parametrize("mode", ["batch", "batch4-pairs", "batch4-small", "split", "one-lane"])
def test_cobatched_pipeline_equals_synchronous_forwards(rg, mode):
    """submit() / flush(): the sampling loop of batch n advances in the same denoiser launches as the exemplar inversion
    of batch n + 1 (shared sessions, two step groups per forward); a batch without exemplars in a lane or of another size
    completes on its own.  Every batch must come out as from its own synchronous forward."""
    dev = torch.device("cuda", 0)
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    db = rg.synth.SyntheticDataset(512, seed=11, device=dev, feat_device=dev)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=db, device=dev)
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    B, N = 4, 6
    batches = _batches(rg, B, N, dev) + _batches(rg, 2, 1, dev)      # the last one has another batch size
    kinds = ["guided", "guided", "guided", "inv", "guided", "base", "guided"]
    if mode == "batch4-pairs":
        # the default rotation over four batch lanes, one workgroup per clip forced (the pipeline picks it by itself only for
        # launches that would not fit the chip side by side: 16 clips + exemplars): batch n shares its lane -- and, where
        # both invert exemplars at the same size, its launches -- with batch n + 4
        batches = _batches(rg, B, 12, dev) + _batches(rg, 2, 1, dev)
        kinds = ["guided"] * 5 + ["inv", "guided", "base", "guided", "guided", "inv", "guided", "guided"]

    if mode == "batch4-small":
        # batches of FEWER clips than lanes behind full ones (4, 4, 4, 4, 4, 2, 1, 1, 3, 4): the lane of a batch must not depend
        # on its size, or a small batch samples a lane's pending batch while an older one is still pending elsewhere
        batches = _batches(rg, B, 5, dev) + _batches(rg, 2, 1, dev) + _batches(rg, 1, 2, dev) + _batches(rg, 3, 1, dev) + _batches(rg, B, 1, dev)
        kinds = ["guided"] * 10

    def ikw(i):
        k = dict(guided=dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1),
                 inv=dict(use_inversion=True), base={})[kinds[i]]
        return dict(k, noise_tape=rg.synth.NoiseTape(4000 + i))

    def args(i):
        d = dict(batches[i])
        d["trans"] = batches[i]["trans"].clone()
        return dict(d, retrieval_method="discourse", inference_kwargs=ikw(i))

    ref = []
    for i in range(len(batches)):
        out = model(**args(i))
        torch.cuda.synchronize()
        ref.append({k: out[k].clone() for k in KEYS})
    model.async_results = True
    if mode == "one-lane":
        model.lanes = model.batch_lanes = 1    # a single pipeline on a single lane stream
    elif mode == "batch4-small":
        model.batch_lanes = 4
    elif mode == "batch4-pairs":
        assert model.batch_lanes == 4 and model.cobatch_lanes == "batch"
        torch.cuda.synchronize()     # (the references above ran one workgroup per sequence: new sessions from here on)
        model._sessions.clear(), model._graphs.clear(), model._graph_owner.clear()
        model.session_options["seq_pairs"] = True
    else:
        model.cobatch_lanes = mode   # "batch": whole batches alternate between the lanes (results two calls later); "split":
        model.batch_lanes = 2        # every batch is cut over the lanes (results one call later)
    for rep in range(2):
        got = []
        for i in range(len(batches)):
            out = model.submit(**args(i))
            if out is not None:
                with torch.cuda.stream(out["done_stream"]):
                    got.append({k: out[k].clone() for k in KEYS})
            del out
        for out in model.flush():
            with torch.cuda.stream(out["done_stream"]):
                got.append({k: out[k].clone() for k in KEYS})
        torch.cuda.synchronize()
        _same(got, ref, model, "submit()/flush() mode %s pass %d kinds %s" % (mode, rep, kinds))
    assert model.flush() == []
    if mode == "batch4-pairs":
        assert any(k[0] == "cobatch" for k in model._graphs), "no launch was shared: the test lost its subject"
        assert all(s.sq.args.pairs == 1 for s in model._sessions.values() if s.sq is not None)
    # the pipeline is an asynchronous-mode feature
    model.async_results = False
    with pytest.raises(rg.capi.RgError):
        model.submit(**args(0))
    assert model.flush() == []


def _tool_body(rg, output):
    """tools/visualize.py:201-291 restated: index the result dict right after the call (no explicit wait: the first read of
    an asynchronous result makes the reader's stream wait, pipeline.AsyncResults), scatter the body parts into the 55-joint
    pose, interpolate 15 -> 30 fps through 6D, move to the host."""
    pred_motion = rg.packing.scatter_parts(output["pred_upper"], output["pred_lower"], output["pred_hands"], output["pred_facepose"])
    pred_motion = rg.packing.upsample_motion(pred_motion, 2)
    pred_facial = rg.packing.upsample_features(output["pred_exps"].float(), 2)
    pred_trans = rg.packing.upsample_features(output["pred_transl"].float(), 2)
    return [t.cpu().numpy() for t in (pred_motion, pred_facial, pred_trans)]


def test_unchanged_tool_loop_and_three_line_pipelined_loop(rg):
    """(i) the reference tool's loop as written -- `output = model(**data)` then its body -- on a model with
    async_results=True: correct without any change (lazy results), throughput of the synchronous path;
    (ii) the three-line change of INTEGRATION.md -- `output = model.submit(**data)`, skip while None, `model.flush()` behind
    the loop -- gives every batch the same bits from the co-batched pipeline."""
    import numpy as np
    dev = torch.device("cuda", 0)
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    db = rg.synth.SyntheticDataset(512, seed=11, device=dev, feat_device=dev)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=db, device=dev)
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    batches = _batches(rg, 4, 5, dev)

    def args(i):
        d = dict(batches[i])
        d["trans"] = batches[i]["trans"].clone()
        return dict(d, retrieval_method="discourse",
                    inference_kwargs=dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1,
                                          noise_tape=rg.synth.NoiseTape(4100 + i)))

    want = []
    for i in range(len(batches)):           # synchronous model: the reference's semantics
        with torch.no_grad():
            want.append(_tool_body(rg, model(**args(i))))
    model.async_results = True
    got = []
    for i in range(len(batches)):           # (i) the same loop, unchanged
        with torch.no_grad():
            output = model(**args(i))
        assert isinstance(output, rg.pipeline.AsyncResults)
        got.append(_tool_body(rg, output))
    _same(got, want, model, "unchanged loop on async_results=True")
    got = []
    for i in range(len(batches)):           # (ii) three changed lines
        with torch.no_grad():
            output = model.submit(**args(i))
        if output is None:
            continue
        got.append(_tool_body(rg, output))
    got += [_tool_body(rg, output) for output in model.flush()]
    _same(got, want, model, "submit()/flush() loop")
    # the schedule is fixed by the constructor arguments alone: eight batch lanes (two of them the lanes of a synchronous forward,
    # three the base lanes) + search + decode
    assert (len(model._lane_streams), model._search_stream is not None, model._decode_stream is not None) == (8, True, True), _where(model)


def test_base_batches_alternate_between_base_lanes(rg):
    """submit() of batches without exemplar inversion: whole batches alternate between `base_lanes` lanes (a launch holds one
    compute unit per sequence, so several such chains fit the chip side by side); every batch complete and bit-identical
    to its own synchronous forward, results in submission order."""
    dev = torch.device("cuda", 0)
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs), database=None, device=dev)
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    batches = _batches(rg, 3, model.base_lanes + 2, dev)      # (more batches than lanes: two lanes see a second batch)

    def args(i):
        d = dict(batches[i])
        d["trans"] = batches[i]["trans"].clone()
        return dict(d, retrieval_method="discourse", inference_kwargs=dict(noise_tape=rg.synth.NoiseTape(4300 + i)))

    ref = []
    for i in range(len(batches)):
        out = model(**args(i))
        torch.cuda.synchronize()
        ref.append({k: out[k].clone() for k in KEYS})
    model.async_results = True
    got = []
    for i in range(len(batches)):
        out = model.submit(**args(i))
        assert out is not None, "a batch without inversion has nothing to wait for: its results are handed out at once"
        got.append({k: out[k].clone() for k in KEYS})     # (first read waits on the reading stream)
    assert model.flush() == []
    torch.cuda.synchronize()
    lanes_used = sorted(p for p in model._slots if p is not None)
    assert lanes_used == list(range(min(model.base_lanes, len(model._lane_streams)))), lanes_used
    _same(got, ref, model, "base lanes")


@pytest.mark.parametrize("use_graphs,calibrate,batch_lanes", [(True, False, 4), (True, True, 4), (False, False, 4), (True, False, 2)])
def test_pipelines_are_bit_stable_under_stream_jitter(rg, use_graphs, calibrate, batch_lanes):
    """The regression test of the round-3 race (two tails on two lanes shared ONE decode graph): guided batches through
    submit() / flush() without any host synchronisation between them -- so tails, chains and front ends of neighbouring
    batches really overlap -- with random delays injected on every stream, repeatedly; then base batches over the base lanes.
    Every pass must reproduce the synchronous forwards bit for bit.  Also with graphs off (eager launches: bisects graph
    buffers against everything else) and with the measured stream choice."""
    dev = torch.device("cuda", 0)
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    db = rg.synth.SyntheticDataset(512, seed=11, device=dev, feat_device=dev)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=db, device=dev,
                                  calibrate_lanes=calibrate, batch_lanes=batch_lanes)
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    model.use_graphs = use_graphs
    batches = _batches(rg, 4, 2 * batch_lanes + 2, dev)
    guided = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)

    def args(i, flags):
        d = dict(batches[i])
        d["trans"] = batches[i]["trans"].clone()
        return dict(d, retrieval_method="discourse", inference_kwargs=dict(flags, noise_tape=rg.synth.NoiseTape(4500 + i)))

    want = {}
    for name, flags in (("guided", guided), ("base", {})):
        want[name] = []
        for i in range(len(batches)):
            out = model(**args(i, flags))
            torch.cuda.synchronize()
            want[name].append({k: out[k].clone() for k in KEYS})
    model.async_results = True
    for rep in range(6 if use_graphs else 2):
        model._jitter = Jitter(100 + rep)
        for name, flags in (("guided", guided), ("base", {})):
            outs = []
            for i in range(len(batches)):
                out = model.submit(**args(i, flags))
                if out is not None:
                    outs.append(out)                 # NOT read here: nothing makes the host or the caller's stream wait
            outs += model.flush()
            got = [{k: o[k].clone() for k in KEYS} for o in outs]
            torch.cuda.synchronize()
            _same(got, want[name], model, "jitter pass %d %s" % (rep, name))
        assert model._jitter.calls > 0
    n_lanes = max(2, model.base_lanes, batch_lanes)      # lanes of a synchronous forward, base lanes, batch lanes
    assert model.lane_report["streams"] == n_lanes + 2 and len(model._lane_streams) == n_lanes and model._search_stream is not None
    if use_graphs:
        assert any(k[0] == "dec" and k[-1] >= 0 for k in model._graphs), "decode graphs are per tail lane"


@pytest.mark.parametrize("dynamic_forms", [False, True])
def test_full_depth_pipeline_is_bit_stable_under_load(rg, dynamic_forms):
    """The benchmarked size (8 layers, 16 clips per batch, 48 exemplars, four batch lanes) through submit() / flush(), three
    passes of twelve batches, every clip of every batch against its synchronous forward -- the fill and drain batches too,
    which bench.py's own check (the last four batches) never sees.  This is the test that caught round 6's failure: with
    248 of 256 vector registers per wave rg_seq2_kernel left room on its SIMDs for waves of the pipeline's small kernels,
    and beside them one workgroup in ~10^5 (two clips of a batch) came out wrong -- only at full depth, only under load
    (csrc/rg_common.h RG_OWN_THE_SIMD).  Also with the launch forms arbitrated on the device (rg_seqx_forward)."""
    dev = torch.device("cuda", 0)
    cfg = rg.synth.default_model_cfg(num_layers=8)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    db = rg.synth.SyntheticDataset(4096, seed=11, device=dev, feat_device=dev)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=db, device=dev,
                                  calibrate_lanes=False, dynamic_forms=dynamic_forms)
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    batches = _batches(rg, 16, 12, dev)
    guided = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)

    def args(i):
        d = dict(batches[i])
        d["trans"] = batches[i]["trans"].clone()
        return dict(d, retrieval_method="discourse", inference_kwargs=dict(guided, noise_tape=rg.synth.NoiseTape(4700 + i)))

    want = []
    for i in range(len(batches)):
        out = model(**args(i))
        torch.cuda.synchronize()
        want.append({k: out[k].clone() for k in KEYS})
    model.async_results = True
    for rep in range(3):
        outs = []
        for i in range(len(batches)):
            out = model.submit(**args(i))
            if out is not None:
                outs.append(out)
        outs += model.flush()
        got = [{k: o[k].clone() for k in KEYS} for o in outs]
        torch.cuda.synchronize()
        _same(got, want, model, "full depth, pass %d, dynamic forms %s" % (rep, dynamic_forms))
    forms = {k[1]: (s_.sq.duo, s_.sq.lane_dyn is not None) for k, s_ in model._sessions.items() if s_.sq is not None and k[1] == "cobatch"}
    assert forms == {"cobatch": (True, dynamic_forms)}, forms

