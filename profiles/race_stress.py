"""Round-4 race hunt: the three-line submit()/flush() loop of tests/test_async_gpu.py, repeated under stream jitter.

    python profiles/race_stress.py --reps 50                 # the product as it is
    python profiles/race_stress.py --reps 50 --old           # round 3's _graph_run: ONE decode graph for every tail stream,
                                                             #   no ordering between its uses (reproduces GPUTEST_r03's failure)
    ... --no-graphs      eager launches (bisect: graph buffers vs everything else)
    ... --no-jitter      no injected delays
    ... --read           read every result as it is handed out (the failing test's loop; default: nothing waits)

One JSON line per repetition on stdout, a summary line at the end, everything also in gpurun_out/race_stress_<tag>.json.
"""
import argparse
import importlib
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KEYS = ("pred_upper", "pred_lower", "pred_facepose", "pred_hands", "pred_transl", "pred_exps", "prev_latentout")
GI = [2] * 25 + [0] * 25


def old_graph_run(self, key, inputs, fn, owner=None):
    """Round 3's behaviour, for reproduction only: the decode key carries no lane and nothing orders two uses of a graph
    that happen on different streams."""
    if key[0] == "dec":
        key = key[:3]
    cur = torch.cuda.current_stream()
    self._used_on(cur, *inputs.values())
    if not self.use_graphs:
        return fn(inputs)
    ent = self._graphs.get(key)
    if ent is None:
        torch.cuda.synchronize()
        static = {k: (None if v is None else torch.empty(v.shape, dtype=v.dtype, device=v.device).copy_(v)) for k, v in inputs.items()}
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            fn(static)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with rg.capi.capture(graph):
            outs = fn(static)
        ent = self._graphs[key] = (graph, static, outs, [None, None])
    graph, static, outs, _ = ent
    if self._jitter is not None:
        self._jitter(cur, key)
    for k, v in inputs.items():
        if v is not None:
            static[k].copy_(v)
    graph.replay()
    return tuple(o.clone() for o in outs)


def main():
    global rg
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--old", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-jitter", action="store_true")
    ap.add_argument("--read", action="store_true")
    ap.add_argument("--calibrate", action="store_true")
    ap.add_argument("--batches", type=int, default=5)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--tag", default=None)
    ap.add_argument("--pairs", action="store_true", help="one workgroup per clip in every rg_seq launch of the asynchronous passes")
    ap.add_argument("--batch-lanes", type=int, default=None)
    ap.add_argument("--layers", type=int, default=2, help="denoiser depth (8: the benchmarked size)")
    ap.add_argument("--db", type=int, default=512)
    ap.add_argument("--model-kwargs", default="{}", help="JSON: more constructor arguments (dynamic_forms, ...)")
    a = ap.parse_args()
    rg = importlib.import_module("rag-gesture_amd")
    if a.old:
        rg.pipeline.MotionDiffusion._graph_run = old_graph_run
    dev = torch.device("cuda", 0)
    cfg = rg.synth.default_model_cfg(num_layers=a.layers)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    db = rg.synth.SyntheticDataset(a.db, seed=11, device=dev, feat_device=dev)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=db, device=dev,
                                  calibrate_lanes=a.calibrate, **({} if a.batch_lanes is None else dict(batch_lanes=a.batch_lanes)),
                                  **json.loads(a.model_kwargs))
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    model.use_graphs = not a.no_graphs
    batches = []
    for i in range(a.batches):
        d = rg.synth.synth_batch(a.B, seed=900 + i, device=dev)
        qs = [rg.synth.synth_query(50 * i + j) for j in range(a.B)]
        d["discourse"] = [q["discourse"] for q in qs]
        d["prominence"] = [q["prominence"] for q in qs]
        d["text_features"] = [q["text_features"].to(dev) for q in qs]
        d["speaker_ids"] = torch.tensor([[q["speaker_id"]] * 150 for q in qs], device=dev)
        batches.append(d)

    def args(i):
        d = dict(batches[i])
        d["trans"] = batches[i]["trans"].clone()
        return dict(d, retrieval_method="discourse",
                    inference_kwargs=dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1,
                                          noise_tape=rg.synth.NoiseTape(4100 + i)))

    want = []
    for i in range(len(batches)):
        out = model(**args(i))
        torch.cuda.synchronize()
        want.append({k: out[k].cpu().numpy() for k in KEYS})
    model.async_results = True
    if a.pairs:              # (the synchronous references above ran one workgroup per sequence)
        torch.cuda.synchronize()
        model._sessions.clear(), model._graphs.clear(), model._graph_owner.clear()
        model.session_options["seq_pairs"] = True
    rng = random.Random(7)

    def jitter(stream, tag):
        if rng.random() < 0.5:
            with torch.cuda.stream(stream):
                torch.cuda._sleep(rng.randrange(1, 6_000_000))

    rows, bad = [], 0
    for rep in range(a.reps):
        model._jitter = None if a.no_jitter else jitter
        outs = []
        for i in range(len(batches)):
            out = model.submit(**args(i))
            if out is None:
                continue
            outs.append({k: out[k].cpu().numpy() for k in KEYS} if a.read else out)
        outs += model.flush()
        got = [o if not isinstance(o, rg.pipeline.AsyncResults) else {k: o[k].cpu().numpy() for k in KEYS} for o in outs]
        torch.cuda.synchronize()
        miss = []
        for i, (g, w) in enumerate(zip(got, want)):
            for k in KEYS:
                if not np.array_equal(g[k], w[k]):
                    clips = [int(c) for c in np.nonzero((g[k] != w[k]).reshape(g[k].shape[0], -1).any(axis=1))[0]]
                    miss.append(dict(batch=i, key=k, max_abs=float(np.nanmax(np.abs(g[k].astype(np.float64) - w[k]))),
                                     n=int((g[k] != w[k]).sum()), of=int(w[k].size), clips=clips))
        row = dict(rep=rep, ok=not miss and len(got) == len(want), mismatches=miss[:6])
        bad += not row["ok"]
        rows.append(row)
        print(json.dumps(row), flush=True)
    summary = dict(reps=a.reps, failed=bad, batches=a.batches, pairs=a.pairs, batch_lanes=model.batch_lanes,
                   cobatch_graphs=sum(1 for k in model._graphs if k[0] == "cobatch"), old_graph_run=a.old, graphs=model.use_graphs, jitter=not a.no_jitter, read=a.read,
                   topology=model.lane_report, lane_streams=len(model._lane_streams), search_stream=model._search_stream is not None,
                   cross_stream_waits=model.graph_cross_stream_waits, hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"))
    print(json.dumps(dict(summary=summary)), flush=True)
    tag = a.tag or ("old" if a.old else "new") + ("_nographs" if a.no_graphs else "") + ("_nojitter" if a.no_jitter else "") + ("_read" if a.read else "")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "race_stress_%s.json" % tag), "w") as f:
        json.dump(dict(summary=summary, rows=rows), f, indent=1)
    return 1 if (bad and not a.old) else 0


if __name__ == "__main__":
    sys.exit(main())
