"""Experiment: `seq_pairs` (one workgroup per clip: conditional sequence, then its classifier-free twin) -- bits, launch time,
and the guided workload (config 3) with whole batches rotating over N lanes of such narrow launches.
    python profiles/dbg/pairs_lanes.py [bits|step] ..."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
what = sys.argv[1:] or ["bits", "step"]

if "bits" in what:
    cfg = rg.synth.default_model_cfg(num_layers=8)
    P = rg.synth.synth_denoiser_state(0, cfg)
    W = rg.denoiser.DenoiserWeights(P, cfg, rg.schedule.Schedule(), "cuda")
    for B in (3, 16, 64):
        data = rg.synth.synth_batch(B, seed=1234)
        x = torch.from_numpy(np.random.Generator(np.random.PCG64(99)).standard_normal((B, 43, 512)).astype(np.float32)).cuda()
        mm = torch.ones(B, 43); mm[:, [10, 21, 32]] = 0; mm[0, 30:] = 0
        outs, times = [], []
        for pairs in (False, True):
            s = rg.denoiser.DenoiserSession(W, B, engine="seq", seq_pairs=pairs)
            s.set_conditions(data["word"], data["audio"], data["speaker_ids"], mm, None)
            o = [s.forward(x, st, step_b=max(0, st - 7), split=B // 2).clone() for st in (49, 34, 0)]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): s.forward(x, 34)
            e1.record(); torch.cuda.synchronize()
            outs.append(o); times.append(e0.elapsed_time(e1) / 20 * 1e3)
        same = all(torch.equal(a, b) for a, b in zip(*outs))
        print("B=%2d: pairs == one-per-sequence bit for bit: %s; forward alone %.0f us (2B workgroups) vs %.0f us (B workgroups, pairs)"
              % (B, same, times[0], times[1]), flush=True)

if "step" in what:
    db = None
    # (batch lanes, max_inflight, seq_pairs: "auto" = the pipeline's own choice)
    cases = [(4, 2, "auto"), (2, 2, "auto"), (4, 2, False), (3, 2, "auto")]
    if len(what) > 1 and what[-1].isdigit():
        cases = cases[:int(what[-1])]
    for lanes, inflight, pairs in cases:
        bench.Workload.streams = None
        wl = bench.Workload(rg, "guided", 16, dev, 0, 32768, database=db)
        db = wl.database
        wl.model.batch_lanes, wl.model.max_inflight = lanes, inflight
        wl.model.session_options = dict(wl.model.session_options, seq_pairs=pairs)
        t0 = time.perf_counter()
        dt = wl.timed(20, 5, torch.cuda.synchronize)
        lat = wl.latency_ms()
        dt2 = wl.timed(48, 0, torch.cuda.synchronize)
        ok = wl.verify()
        paired = sorted({(k[0], k[1], s_.sq.args.pairs) for k, s_ in wl.model._sessions.items() if s_.sq is not None})
        print("batch lanes %d max_inflight %d seq_pairs %s: %.2f ms per batch of 16 over 20, %.2f over 48 (%.0f frames/s), latency %s, verified %s (%d batches), wall %.0f s; sessions (B, role, pairs): %s"
              % (lanes, inflight, pairs, dt / 20 * 1e3, dt2 / 48 * 1e3, 16 * 150 * 48 / dt2, lat, ok.get("verified"), ok.get("batches"),
                 time.perf_counter() - t0, paired), flush=True)
        del wl
        torch.cuda.empty_cache()
