"""Diagnostic: host and device timeline of the graph replays of one guided bench step (lanes, inversion / sampling).
For every _graph_run call: host time at call and at return (graph launch cost), device start/end from events on the
stream the graph is replayed on, all relative to the start of the step."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "16"))
GI = [2] * 25 + [0] * 25
cfg = rg.synth.default_model_cfg(num_layers=8)
vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
database = rg.synth.SyntheticDataset(int(os.environ.get("DB", "32768")), seed=2025, device=dev, feat_device=dev)
model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=database, device=dev)
model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
model.eval()
if "SEQ_PAIRS" in os.environ:
    model.session_options["seq_pairs"] = bool(int(os.environ["SEQ_PAIRS"]))
if "MAX_INFLIGHT" in os.environ:
    model.max_inflight = int(os.environ["MAX_INFLIGHT"])
if "SLOTS" in os.environ:
    model.slots = int(os.environ["SLOTS"])
if "LANES" in os.environ:
    model.lanes = int(os.environ["LANES"])
if "BATCH_LANES" in os.environ:
    model.batch_lanes = int(os.environ["BATCH_LANES"])
model.async_results = bool(int(os.environ.get("PIPELINED", "0")))
model.calibrate_lanes = os.environ.get("CALIBRATE", "1") == "1"
if "SAMPLE_LANES" in os.environ:
    model.sample_lanes = int(os.environ["SAMPLE_LANES"])
data = rg.synth.synth_batch(B, seed=1234, device=dev)
qs = [rg.synth.synth_query(i) for i in range(B)]
data["discourse"] = [q["discourse"] for q in qs]
data["prominence"] = [q["prominence"] for q in qs]
data["text_features"] = [q["text_features"].to(dev) for q in qs]
data["speaker_ids"] = torch.tensor([[q["speaker_id"]] * 150 for q in qs], device=dev)
trans0 = data["trans"].clone()
log = []
orig = model._graph_run
t_step = [0.0]
ev_step = [None]


def traced(key, inputs, fn, owner=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter()
    e0.record()
    out = orig(key, inputs, fn, owner=owner)
    e1.record()
    log.append((key, (h0 - t_step[0]) * 1e3, (time.perf_counter() - t_step[0]) * 1e3, e0, e1))
    return out


def one_step(trace=False, sync=True):
    d = dict(data)
    d["trans"] = trans0.clone()
    model.model.database.test_indexes.clear()
    ikw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
    if trace:
        log.clear()
        torch.cuda.synchronize()
        ev_step[0] = torch.cuda.Event(enable_timing=True)
        ev_step[0].record()
        t_step[0] = time.perf_counter()
    call = model.submit if os.environ.get("COBATCH") else model
    out = call(**dict(d, retrieval_method="discourse", inference_kwargs=ikw))
    t_host = (time.perf_counter() - t_step[0]) * 1e3
    if not sync:
        return t_host, None
    torch.cuda.synchronize()
    return t_host, (time.perf_counter() - t_step[0]) * 1e3


for _ in range(int(os.environ.get("WARM", "3" if not os.environ.get("COBATCH") else "9"))):     # (the co-batched pipeline has graphs per phase, lane and slot)
    one_step()
if os.environ.get("COBATCH"):
    model.flush()
    for _ in range(int(os.environ.get("WARM2", "5"))):
        one_step()
model._graph_run = traced
model.model.gesture_rep_encoder.graph_runner = traced
for rep in range(2):
    t_host, t_all = one_step(trace=True)
    print("step: host returns from forward at %.1f ms, device done at %.1f ms (lanes=%d)" % (t_host, t_all, model.lanes))
    for key, h0, h1, e0, e1 in log:
        print("   %-44s host call %6.1f -> %6.1f ms | device %6.1f -> %6.1f ms (%.1f)" % (
            str(key)[:44], h0, h1, ev_step[0].elapsed_time(e0), ev_step[0].elapsed_time(e1), e0.elapsed_time(e1)))

blocks = []
if os.environ.get("HOSTBLOCK"):
    # every host call that can wait for the device, with its duration and the frames that made it
    import traceback

    def wrap(owner, name):
        fn = getattr(owner, name)

        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            dt = (time.perf_counter() - t0) * 1e3
            if dt > 0.5:
                fr = [f for f in traceback.extract_stack()[:-1] if "rag-gesture_amd" in f.filename][-3:]
                blocks.append(((t0 - t_step[0]) * 1e3, dt, "%s.%s" % (getattr(owner, "__name__", owner), name),
                               " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr))))
            return r
        setattr(owner, name, w)
    for owner, names in ((torch.Tensor, ("cpu", "item", "tolist", "numpy", "to", "nonzero", "__bool__", "__int__", "__float__", "copy_", "__setitem__", "__getitem__", "pin_memory", "index_copy_", "index_select", "clone", "record_stream")),
                         (torch.cuda.Event, ("synchronize",)), (torch.cuda.Stream, ("synchronize",)), (torch.cuda, ("synchronize",)),
                         (torch, ("tensor", "as_tensor", "stack", "cat", "zeros", "empty", "randn"))):
        for nm in names:
            wrap(owner, nm)

if os.environ.get("TIMED_REGION"):
    # what bench.py times: K submissions into an EMPTY pipeline + flush(), no synchronisation in between
    K = int(os.environ["TIMED_REGION"])
    model.flush()
    torch.cuda.synchronize()
    for _ in range(K):            # (a first, untraced region: its fill and drain capture the graphs they need)
        one_step(sync=False)
    model.flush()
    torch.cuda.synchronize()
    one_step(trace=True, sync=False)
    for _ in range(K - 1):
        one_step(sync=False)
    model.flush()
    torch.cuda.synchronize()
    print("timed region: %d batches + flush done at %.1f ms" % (K, (time.perf_counter() - t_step[0]) * 1e3))
    for i, (key, h0, h1, e0, e1) in enumerate(log):
        if key[0] in ("cobatch", "dec", "invert", "sample", "inv", "loop") or "cob" in str(key[0]) or True:
            print("   %-44s host call %6.1f -> %6.1f ms | device %6.1f -> %6.1f ms (%.1f)" % (
                str(key)[:44], h0, h1, ev_step[0].elapsed_time(e0), ev_step[0].elapsed_time(e1), e0.elapsed_time(e1)))
    sys.exit(0)

if os.environ.get("BACK_TO_BACK"):
    # three steps without a synchronisation in between: where does step n + 1 start relative to step n's end?
    one_step(trace=True, sync=False)
    marks = [len(log)]
    for _ in range(int(os.environ.get("BACK_TO_BACK", "3")) - 1):
        one_step(sync=False)
        marks.append(len(log))
    torch.cuda.synchronize()
    print("three steps back to back, all done at %.1f ms" % ((time.perf_counter() - t_step[0]) * 1e3))
    for i, (key, h0, h1, e0, e1) in enumerate(log):
        if i in marks:
            print("   ---- next step")
        print("   %-44s host call %6.1f -> %6.1f ms | device %6.1f -> %6.1f ms (%.1f)" % (
            str(key)[:44], h0, h1, ev_step[0].elapsed_time(e0), ev_step[0].elapsed_time(e1), e0.elapsed_time(e1)))
    for t0, dt, what, where in blocks:
        print("   host blocked %6.1f -> %6.1f ms (%5.1f) in %s  %s" % (t0, t0 + dt, dt, what, where))
