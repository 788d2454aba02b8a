"""PMC / kernel-trace target for the sequence-stationary denoiser forward: REPS launches of rg_seq_forward on one lane of the
benchmarked pipeline (B = 64 clips = 16 sampling + 48 inverting -> 128 sequences, 8 layers) and, for the co-running case,
the same on two streams.  Run under rocprofv3, one counter set per pass:

    rocprofv3 --kernel-trace --stats -d gpurun_out/seq_stats --output-format csv -- python3 profiles/pmc_seq.py
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/seq_fetch --output-format csv -- python3 profiles/pmc_seq.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/seq_write --output-format csv -- python3 profiles/pmc_seq.py
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/seq_mfma --output-format csv -- python3 profiles/pmc_seq.py

then `python profiles/pmc_seq_summarize.py gpurun_out`.  Prints the algorithmic numbers of one launch (JSON) for the join."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rg = importlib.import_module("rag-gesture_amd")
B, REPS, L, T = int(os.environ.get("SEQ_B", "64")), int(os.environ.get("SEQ_REPS", "10")), 8, 43
cfg = rg.synth.default_model_cfg(num_layers=L)
W = rg.denoiser.DenoiserWeights(rg.synth.synth_denoiser_state(0, cfg), cfg, rg.schedule.Schedule(), "cuda")
PAIRS = os.environ.get("SEQ_PAIRS", "0") == "1"     # one workgroup per clip (conditional sequence, then its twin): B workgroups
DUO = os.environ.get("SEQ_DUO", "0") == "1"         # rg_seq2_forward: two sequences of a kind per workgroup (with PAIRS: B / 2 workgroups)
sess = rg.denoiser.DenoiserSession(W, B, engine="seq", seq_pairs=PAIRS, seq_duo=DUO)
d = rg.synth.synth_batch(B, seed=1)
mask = torch.ones(B, 43); mask[:, [10, 21, 32]] = 0
sess.set_conditions(d["word"], d["audio"], d["speaker_ids"], mask, {c: mask.clone() for c in rg.denoiser.CONDS})
x = torch.randn(B, 43, 512, device="cuda")
for _ in range(2):
    sess.forward(x, 30, 19, 16)
torch.cuda.synchronize()
for i in range(REPS):
    sess.forward(x, 49 - i, i, 16)        # 16 clips sampling at step 49 - i, 48 exemplars inverting at step i
torch.cuda.synchronize()
st = W.seq_streams
unit = 2.0 * T * 512 * 512
att = 2.0 * T * 32 * 32 * 16
flops = B * ((16 * L + 2) * unit + L * 5 * att + (10 * L + 2) * unit + L * 2 * att)
# bytes one launch must move at least once (HBM-side algorithmic traffic): the weight stream + this step's parameter /
# table fragments (shared by all workgroups), every clip's cross-attention fragments, the latent in and the head out
alg = (st.wstream.numel() * 2 + 2 * (st.pstream[0].numel() * 4 + st.ustream[0].numel() * 4) + sess.sq.afrag.numel() * 2
       + B * T * 512 * 4 + 2 * B * T * 512 * 4)
# bytes the workgroups pull through their LDS rings (what the per-CU intake sees): every sequence streams its own copy
per_cond = (16 * L + 2) * 520 * 1024 + L * 3 * 32 * 1024
per_unc = (10 * L + 2) * 520 * 1024 + L * 16 * 1024
if DUO:   # + the fp32 tile round trips (3 per layer) and bf16 panel images (4 per layer for conditional pairs) through xbuf / gbuf: L2-resident scratch
    # (round 6: two xbuf round trips per layer and sequence pair of either kind, three gbuf panel images per conditional pair)
    scratch = (B // 2) * L * (2 * (2 * 2 * 192 * 1024) + 3 * 2 * 96 * 1024)
else:
    scratch = 0
wgs = (B // 2 if PAIRS else B) if DUO else (B if PAIRS else 2 * B)
import hashlib
_src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rag-gesture_amd", "csrc", "rg_seq2.hip" if DUO else "rg_seq.hip")
print(json.dumps(dict(kernel_source_sha256=hashlib.sha256(open(_src, "rb").read()).hexdigest(), kernel="rg_seq2_kernel" if DUO else "rg_seq_kernel", sequences=2 * B, workgroups=wgs, launches=REPS, flops_per_launch=flops, algorithmic_hbm_bytes=alg,
                      lds_ring_bytes=B * (per_cond + per_unc) // (2 if DUO else 1), scratch_round_trip_bytes=scratch)))
