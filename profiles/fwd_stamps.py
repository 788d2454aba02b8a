"""Per-tile wall-clock stamps of the persistent denoiser forward (rg_fwd_args.stamps): where a forward step's time
goes, per stage type -- dependency wait, A panel + K loop, epilogue + publish -- and how busy the workgroups are.
    python profiles/fwd_stamps.py [B ...]"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rg = importlib.import_module("rag-gesture_amd")
F = rg.fwd
cfg = rg.synth.default_model_cfg(num_layers=8)
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, rg.schedule.Schedule(), "cuda")
NAMES = ["EMBED", "QKV_SA", "SAOUT", "Q3_CA", "MIX", "FF1", "FF2", "FFOUT", "HEAD"]
for B in [int(a) for a in sys.argv[1:]] or [16, 48]:
    sess = rg.denoiser.DenoiserSession(W, B, persistent=True)
    d = rg.synth.synth_batch(B, seed=1)
    mask = torch.ones(B, 43); mask[:, [10, 21, 32]] = 0
    sess.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, {c: torch.ones(B, 43) for c in rg.denoiser.CONDS})
    x = torch.randn(B, 43, 512, device="cuda")
    for _ in range(3):
        sess.forward(x, 20)
    torch.cuda.synchronize()
    stamps = torch.zeros(sess.pf.n_tiles, 4, dtype=torch.int64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sess.pf.run(x, 20, stamps=stamps); e1.record()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().astype(np.float64) / 100.0   # us (100 MHz counter)
    t0 = st[:, 0].min()
    st -= t0
    tiles = sess.pf.sched_host[F.SCHED_HEADER:].reshape(-1, 4)
    typ = tiles[:, 0] & 0xff
    print("B=%d: %d tiles, kernel %.1f us by events, %.1f us first stamp -> last publish; aborted=%s" %
          (B, len(tiles), e0.elapsed_time(e1) * 1e3, st[:, 3].max(), sess.pf.aborted()))
    busy = (st[:, 3] - st[:, 0]).sum()
    print("  sum of tile spans %.0f us = %.1f%% of 256 CUs x kernel; waiting %.0f us (%.1f%%), panel+K loop %.0f, epilogue %.0f"
          % (busy, 100 * busy / (256 * st[:, 3].max()), (st[:, 1] - st[:, 0]).sum(), 100 * (st[:, 1] - st[:, 0]).sum() / busy,
             (st[:, 2] - st[:, 1]).sum(), (st[:, 3] - st[:, 2]).sum()))
    print("  %-7s %6s %9s %9s %9s %9s" % ("stage", "tiles", "wait us", "K-loop us", "epi us", "span us"))
    for t in range(9):
        m = typ == t
        if m.any():
            print("  %-7s %6d %9.2f %9.2f %9.2f %9.2f" % (NAMES[t], m.sum(), (st[m, 1] - st[m, 0]).mean(),
                  (st[m, 2] - st[m, 1]).mean(), (st[m, 3] - st[m, 2]).mean(), (st[m, 3] - st[m, 0]).mean()))
    # progress of sequence 0 through the layers: publish time of its last tile per stage
    s0 = tiles[:, 1] == 0
    lay = tiles[:, 0] >> 8
    line = []
    for l in range(W.L):
        m = s0 & (lay == l) & (typ == F.FFOUT)
        line.append("%.0f" % st[m, 3].max())
    print("  sequence 0 finishes layer l at us:", " ".join(line))
