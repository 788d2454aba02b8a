"""Two denoiser sessions of B clips each replaying their forward-step graphs CONCURRENTLY on two streams (the inversion's two
lanes): microseconds per forward step of the pair, per B.  Shows what workgroup-count quantisation costs (B = 24: 2 x 132
workgroups per N = 512 GEMM on 256 CUs; B = 23: 2 x 124)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rg = importlib.import_module("rag-gesture_amd")
cfg = rg.synth.default_model_cfg(num_layers=8)
sch = rg.schedule.Schedule()
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, sch, "cuda")
STEPS = 10
WAVES = [int(v[8:]) for v in sys.argv[1:] if v.startswith("--waves=")] or [0]   # rg_set_gemm_waves: 0 auto, 16 = two 8-wave workgroups per CU
BS = [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [20, 22, 23, 24, 26, 8]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for B, waves in [(b, w) for b in BS for w in WAVES]:
    W.h.lib.rg_set_gemm_waves(W.h._h, waves)
    W.h.lib.rg_set_gemm_path(W.h._h, int(os.environ.get("GEMM_PATH", "0")))
    graphs = []
    for st in streams:
        sess = rg.denoiser.DenoiserSession(W, B, ln_mode="folded", engine=os.environ.get("ENGINE") or None)
        d = rg.synth.synth_batch(B, seed=1)
        mask = torch.ones(B, 43); mask[:, [10, 21, 32]] = 0
        sess.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, {c: torch.ones(B, 43) for c in rg.denoiser.CONDS})
        x = torch.randn(B, 43, 512, device="cuda")
        with torch.cuda.stream(st):
            for s in range(2):
                sess.forward(x, 40 - s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                for s in range(STEPS):
                    sess.forward(x, 40 - s)
        graphs.append((g, sess, x))
    res = {}
    for name, use in (("one lane alone", graphs[:1]), ("two lanes together", graphs)):
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for st in streams[:len(use)]:
                st.wait_event(e0)
            for (g, _, _), st in zip(use, streams):
                with torch.cuda.stream(st):
                    g.replay()
            for st in streams[:len(use)]:
                torch.cuda.current_stream().wait_stream(st)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / STEPS)
        res[name] = best
    print("waves=%2d B=%2d per lane (M=%4d): one lane alone %.1f us per step, two lanes together %.1f us per step of the pair = %.2f us per clip"
          % (waves, B, 2 * B * 43, res["one lane alone"], res["two lanes together"], res["two lanes together"] / (2 * B)), flush=True)
