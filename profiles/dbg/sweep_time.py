"""Device time of the retrieval sweep + selection launches alone (buffers resident): fused (two launches) vs the round-3 form."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rg = importlib.import_module("rag-gesture_amd")
smp = rg.synth.synth_retrieval_samples(32768, seed=2025)
index = rg.retrieval.DiscourseIndex(rg.retrieval.build_db_dicts(smp), "cuda")
queries = []
for i in range(16):
    q = rg.synth.synth_query(1000 + i)
    queries += rg.retrieval.discourse_queries(q["discourse"], q["prominence"], q["speaker_id"])
for fused in (True, False):
    index.fused_sweep = fused
    bufs = index.sweep_buffers(queries)
    for _ in range(3): index.sweep_launch(bufs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): index.sweep_launch(bufs)
    e1.record(); torch.cuda.synchronize()
    print("fused=%s: %d query relations x 32768 entries: %.1f us per sweep + selection" % (fused, len(queries), e0.elapsed_time(e1) / 50 * 1e3))
