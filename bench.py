#!/usr/bin/env python3
"""Benchmark of the RAG-Gesture inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload guided|base] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic clips through the drop-in
`model(**data)`: 4x VAE encode -> conditioning precompute -> [batched DDIM inversion of the
retrieved exemplars] -> 50-step DDIM with CFG [+ insertion guidance] -> 4x VAE decode
(150 SMPL-X frames per clip at 15 fps).  Inputs are resident in HBM before
the timed region.  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), clips
sharded across ranks (weak scaling, B clips per rank), results all-gathered once per step.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, HIP-event
timed in a separate instrumented step of the same workload) and `cpu_baseline` (the CPU oracle
= a faithful port of the reference, timed on a bounded sample, rank 0 at N = 1 only).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher check (no GPU): every rank prints the environment it was started with and exits")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["guided", "base"], default="guided")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: 16 guided, 32 base)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    if args.dry_launch:
        print(json.dumps(dict(rank=rank, local_rank=local_rank, world=world, master_addr=os.environ.get("MASTER_ADDR"),
                              master_port=os.environ.get("MASTER_PORT"), pid=os.getpid())))
        return
    global torch
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("RG_BENCH_FORCE_DIST") == "1":   # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    rg = importlib.import_module("rag-gesture_amd")
    B = args.batch or (16 if args.workload == "guided" else 32)
    cfg = rg.synth.default_model_cfg(num_layers=8)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    guided = args.workload == "guided"
    # guided: the retrieval DB (discourse metadata + token features) is replicated on every GPU
    database = rg.synth.SyntheticDataset(args.db_size, seed=2025, device=dev, feat_device=dev) if guided else None
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=guided),
                                  database=database, device=dev)
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()

    data = rg.synth.synth_batch(B, seed=1234 + rank, device=dev)
    if guided:  # per-clip discourse relations / prominence / BERT token features (3 relations per query)
        qs = [rg.synth.synth_query(1000 * rank + i) for i in range(B)]
        data["discourse"] = [q["discourse"] for q in qs]
        data["prominence"] = [q["prominence"] for q in qs]
        data["text_features"] = [q["text_features"].to(dev) for q in qs]
        data["speaker_ids"] = torch.tensor([[q["speaker_id"]] * 150 for q in qs], device=dev)
    trans0 = data["trans"].clone()

    def one_step(gather=True):
        """gather=False: the rank-local part only (the instrumented extra steps below run on rank 0 alone and must
        not enter a collective the other ranks are not in)."""
        d = dict(data)
        d["trans"] = trans0.clone()  # forward re-zeroes trans in place like the reference
        ikw = {}
        if guided:
            ikw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
        out = model(**dict(d, retrieval_method="discourse", inference_kwargs=ikw))
        packed = torch.cat([out["pred_upper"], out["pred_lower"], out["pred_facepose"], out["pred_hands"],
                            out["pred_transl"], out["pred_exps"]], dim=-1)
        if dist is not None and gather:  # the only collective on the path: final result gather (RCCL over xGMI)
            gathered = torch.empty(world * B, packed.shape[1], packed.shape[2], device=dev)
            dist.all_gather_into_tensor(gathered, packed.contiguous())
            return gathered
        return packed

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * 150 * args.steps / dt

    if args.phases and rank == 0:
        # phase walls need device syncs at phase boundaries, which serialise concurrent lanes: the breakdown is taken
        # on a single lane (one warm-up step captures its graphs), the timed run above used model.lanes lanes
        lanes_prod, model.lanes = model.lanes, 1
        one_step(gather=False)
        model.profile_phases, model.phase_ms = True, {}
        if guided:
            model.model.database.phase_ms = model.phase_ms
        one_step(gather=False)
        torch.cuda.synchronize()
        model.profile_phases = False
        if guided:
            model.model.database.phase_ms = None
        model.lanes = lanes_prod
        print("phase breakdown (ms, one synchronised single-lane step; the timed run used %d lanes): " % lanes_prod +
              ", ".join("%s %.1f" % kv for kv in model.phase_ms.items()), file=sys.stderr)

    # ---- roofline of the dominant kernel (HIP events around every rg_gemm launch, one extra step)
    roofline = None
    if rank == 0:
        h = rg.capi.get_handle(local_rank)

        def gemm_events():
            """HIP events around every rg_gemm launch of one eager step (graph replays cannot hold events; the
            launches are the same).  Returns (launches, total ms, total flops) of the bf16-A GEMMs = variant 1
            (gemm_dma_kernel<true,...> / gemm_bf16_big_kernel: every per-step denoiser GEMM, ~2/3 of the GPU time)."""
            model.use_graphs = False
            one_step(gather=False)
            torch.cuda.synchronize()
            h.lib.rg_profile_begin(h._h)
            one_step(gather=False)
            model.use_graphs = True
            n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            h.lib.rg_profile_end(h._h, 1, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl))
            return n.value, ms.value, fl.value

        n, ms, fl = gemm_events()
        ach = fl / (ms * 1e-3) if ms > 0 else 0.0
        # HBM-side bytes per launch from the committed PMC passes over the same kernels
        # (profiles/r01e_pmc_gemm_traffic.txt explains how they were collected and corrected); null if absent
        traffic = None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01e_pmc_gemm_traffic.json")) as f:
                rows = json.load(f)
            traffic = round(sum(r["fetch_bytes"] + r["write_bytes"] for r in rows) / len(rows))
        except (OSError, ValueError, KeyError, ZeroDivisionError):
            pass
        roofline = {"bound": "mfma", "kernel": "rg_gemm bf16-A kernels (gemm_dma_kernel<true,..>, gemm_bf16_big_kernel; bf16 MFMA, fp32 accumulate)",
                    "achieved": round(ach / 1e12, 3), "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_PEAK, 5), "traffic": traffic, "launches": n,
                    "avg_launch_us": round(ms * 1e3 / max(1, n), 2), "flops_per_launch_avg": round(fl / max(1, n)),
                    "lanes": model.lanes}
        if model.lanes > 1:
            # The timed run cuts the batch into concurrent lanes: its launches are 1/lanes of the batch each and
            # overlap in time, so the per-launch figure above understates what the chip does.  The same kernels on
            # one lane (whole batch per launch; the shapes of the PMC passes and of profiles/r01f_*):
            lanes_prod, model.lanes = model.lanes, 1
            n1, ms1, fl1 = gemm_events()
            model.lanes = lanes_prod
            a1 = fl1 / (ms1 * 1e-3) if ms1 > 0 else 0.0
            roofline["single_lane"] = {"achieved": round(a1 / 1e12, 3), "frac": round(a1 / MFMA_BF16_PEAK, 5), "launches": n1,
                                       "avg_launch_us": round(ms1 * 1e3 / max(1, n1), 2),
                                       "flops_per_launch_avg": round(fl1 / max(1, n1))}
    if dist is not None:
        dist.barrier()

    # ---- CPU baseline: the oracle (faithful port of the reference) on a bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pipeline as opipe, diffusion as odf, retrieval as oret
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = max(1, min(cores, 32))  # the port is bandwidth/latency bound well below that
        torch.set_num_threads(cores)
        P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
        cdata = rg.synth.synth_batch(1, seed=1234)
        ckw, cre = {}, None
        tape = rg.synth.NoiseTape(1)
        if guided:
            cpu_db = oret.build_db_dicts([dict(r, text_feature=r["text_feature"].cpu()) for r in database.retrieval_samples])
            cpu_ds = rg.synth.SyntheticDataset(0)
            q0 = rg.synth.synth_query(0)
            ccond = dict(text_features=[q0["text_features"]], discourse=[q0["discourse"]], prominence=[q0["prominence"]],
                         speaker_ids=torch.tensor([[q0["speaker_id"]] * 150]))
            ckw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
        tc = time.perf_counter()
        with torch.no_grad():
            if guided:  # sweep over the same DB + exemplar encode + placement, as the reference does per clip
                cre = lambda tp: oret.database_forward(P, vae_cfgs, cpu_db, cpu_ds, ccond, ["bench_query"], tp)
                opipe.motion_diffusion_forward(P, cfg, vae_cfgs, odf.SpacedSchedule(), cdata, tape, re_dict=cre, **ckw)
            else:
                opipe.motion_diffusion_forward(P, cfg, vae_cfgs, odf.SpacedSchedule(), cdata, tape, re_dict=None)
        tc = time.perf_counter() - tc
        cpu = {"value": round(150.0 / tc, 2), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": "1 clip (150 frames), same workload (%s), torch fp32 on %d host threads, %.1f s"
                         % (args.workload, cores, tc)}

    if rank == 0:
        line = {
            "metric": "SMPL-X frames/sec, len150 DDIM-50 + insertion guidance; 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if dist is not None else None, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("guided discourse config: discourse retrieval over a replicated %d-entry DB, "
                                    "use_inversion + insertion_guidance decreasing_till_25, <=3 exemplars/clip, "
                                    "len150 DDIM-50" % args.db_size
                                    if args.workload == "guided" else "base diffusion len150 DDIM-50 (no guidance)"),
                       "clips_per_gpu": B, "global_batch": world * B, "frames_per_clip": 150, "ddim_steps": 50,
                       "denoiser": "8 layers x 512, CFG x2 rows", "vae": "all_encoder, 8 layers, synthetic hparams",
                       "weights": "random-init at config shapes", "parallelism": "clip-sharded x%d" % world},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
