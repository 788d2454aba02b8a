"""PMC target: the bf16-A rg_gemm launches of ONE denoiser layer at the two row counts of the guided
workload (sampling M = 1376, exemplar inversion M = 4128), each shape launched REPS times on rotating
operands.  Run under rocprofv3 with one counter per pass (the full bench has ~66k launches and does not
finish a PMC pass inside the box limit):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 profiles/pmc_gemm_shapes.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 profiles/pmc_gemm_shapes.py

then `python profiles/pmc_summarize.py <fetch counter csv> <write counter csv>`.
Prints the launch order (shape per dispatch) so the counter rows can be attributed."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rg = importlib.import_module("rag-gesture_amd")
G = rg.gemm
h = rg.capi.get_handle(0)
REPS = 6
D = 512
# (name, rows relative to M, N, K, fp32 out?, bf16 copy?)  -- one layer of DenoiserSession.forward (bf16 path)
LAYER = [("qkv", 1.0, 3 * D, D, True, False), ("sa_out", 1.0, D, D, True, True), ("q3", 0.5, 3 * D, D, True, False),
         ("ca_mix", 1.0, D, 4 * D, True, True), ("ff1", 1.0, 2 * D, D, False, False), ("ff2", 1.0, D, 2 * D, True, False),
         ("ffn_out", 1.0, D, D, True, True)]
order = []
keep = []
for M0 in [int(v) for v in os.environ.get("PMC_ROWS", "1376,4128").split(",")]:   # round 2 (co-batched lane): PMC_ROWS=2752
    for name, frac, N, K, f32out, copy in LAYER:
        M = int(M0 * frac)
        for r in range(REPS):
            a = torch.randn(M, K, device="cuda").bfloat16()
            W = G.pack_weight(torch.randn(N, K) * 0.05, "cuda")
            out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32out else torch.bfloat16)
            kw = dict(out2=torch.empty(M, N, device="cuda", dtype=torch.bfloat16)) if copy else {}
            keep.append((a, W, out, kw))
            G.gemm(h, M=M, N=N, K=K, W=W, out=out, A=a, bias=torch.zeros(N, device="cuda"), **kw)
            alg = M * K * 2 + N * K * 2 + M * N * (4 if f32out else 2) + (M * N * 2 if copy else 0)
            order.append(dict(name=name, M=M, N=N, K=K, algorithmic_bytes=alg, flops=2 * M * N * K))
torch.cuda.synchronize()
with open(os.environ.get("PMC_ORDER", "gpurun_out/pmc_order.json"), "w") as f:
    json.dump(order, f)
print("launched", len(order), "GEMMs")
