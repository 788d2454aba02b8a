"""bf16-A rg_gemm launches of the denoiser at given rows, graph-replayed (20 launches on rotating operands), alone and with
the same chain on a second stream: microseconds per launch per kernel choice (rg_set_gemm_path: 0 auto, 5 never the
big-tile kernel, 6 big-tile 128x128, 4 big-tile 128x256)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rg = importlib.import_module("rag-gesture_amd")
G = rg.gemm
h = rg.capi.get_handle(0)
D = 512
M0 = int(sys.argv[1]) if len(sys.argv) > 1 else 2752
SHAPES = [("qkv", M0, 3 * D, D), ("sa_out", M0, D, D), ("q3", M0 // 2, 3 * D, D), ("ca_mix", M0, D, 4 * D), ("ff1", M0, 2 * D, D),
          ("ff2", M0, D, 2 * D)]
REPS, SETS = 20, 3
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def chain(M, N, K):
    ops = []
    for _ in range(SETS):
        ops.append((torch.randn(M, K, device="cuda").bfloat16(), G.pack_weight(torch.randn(N, K) * 0.05, "cuda"),
                    torch.empty(M, N, device="cuda"), torch.zeros(N, device="cuda")))
    def run():
        for r in range(REPS):
            a, W, out, b = ops[r % SETS]
            G.gemm(h, M=M, N=N, K=K, W=W, out=out, A=a, bias=b)
    return run


for name, M, N, K in SHAPES:
    line = "%-7s M=%5d N=%5d K=%5d:" % (name, M, N, K)
    for path in [int(v) for v in os.environ.get("PATHS", "0,5,6,4").split(",")]:
        h.lib.rg_set_gemm_path(h._h, path)
        graphs, keep = [], []
        for st in streams:
            run = chain(M, N, K)
            with torch.cuda.stream(st):
                run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                with torch.cuda.graph(g, stream=st):
                    run()
            graphs.append(g)
            keep.append(run)       # the closure owns the operands the graph replays on
        res = []
        for use in (graphs[:1], graphs):
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for st in streams[:len(use)]:
                    st.wait_event(e0)
                for g, st in zip(use, streams):
                    with torch.cuda.stream(st):
                        g.replay()
                for st in streams[:len(use)]:
                    torch.cuda.current_stream().wait_stream(st)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
            res.append(best)
        line += "  path %d: %5.1f / %5.1f" % (path, res[0], res[1])
        print("   ... %s path %d done" % (name, path), file=sys.stderr, flush=True)
    print(line + "   (us per launch: one stream / two streams together)", flush=True)
h.lib.rg_set_gemm_path(h._h, 0)
