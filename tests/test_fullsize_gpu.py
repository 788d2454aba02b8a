"""GPU: the two single-GPU BASELINE configurations at FULL size, checked per clip against the CPU oracle.

  config 3: B = 16, use_inversion + insertion_guidance (decreasing_till_25), discourse retrieval over the replicated
            32768-entry DB, bf16, concurrent lanes, 8 layers  -- clips {0, 7, 15}
  config 2: B = 32 base diffusion, bf16                          -- clips {0, 31}

Clips are independent, so clip i of the batched HIP run must equal a batch-of-one oracle run that is fed clip i's
inputs and clip i's share of the noise.  The batched run's explicit noise tape is recorded; ClipTape replays, in the
oracle's draw order, the rows that belong to one clip (VAE noise rows [10 i, 10 i + 10), the exemplar-encode draws of
the clip's own exemplars, row i of every [B, 43, 512] draw).
Bars: retrieved samples, bounds and placement exact; final latent <= 1e-2 relative over all kept rows and <= 3e-2 on the
worst token row (bf16 operands vs fp32, rows 10/20/30 excluded as everywhere); decoded translation
<= 3e-2.  Every measured value goes through the `parity` recorder (tests/conftest.py) and is printed with the run.

  config 3 as BENCHMARKED: three different B = 16 batches through model.submit() / flush() (asynchronous results, the
            sampling loop of batch n sharing its denoiser launches with the exemplar inversion of batch n + 1 / n + 2, whole
            batches alternating between two lanes): clips {0, 7, 15} of the MIDDLE batch against the per-clip oracle.
"""
import pytest
import torch

from conftest import relerr, rowerr
from oracle import diffusion as odf, pipeline as opipe, retrieval as oret

pytestmark = pytest.mark.gpu
KEEP = [r for r in range(43) if r not in (10, 20, 30)]
GI = [0] * 25 + list(range(25))


class RecordingTape:
    def __init__(self, tape):
        self.tape, self.record = tape, []

    def draw(self, shape, device=None):
        t = self.tape.draw(shape)
        self.record.append(t)
        return t.to(device) if device is not None else t


class ClipTape:
    """Clip i's share of a recorded batched tape, in the order a batch-of-one run draws it."""

    def __init__(self, record, i, B, exemplar_clips):
        E = len(exemplar_clips)
        seq = [record[k][10 * i:10 * i + 10] for k in range(4)]                      # VAE encode of the clip batch
        assert all(record[k].shape == (B * 10, 1, 512) for k in range(4))
        for e, b in enumerate(exemplar_clips):                                        # 4 draws per visited exemplar
            chunk = record[4 + 4 * e:8 + 4 * e]
            assert all(c.shape == (10, 1, 512) for c in chunk)
            if b == i:
                seq += chunk
        for t in record[4 + 4 * E:]:
            assert t.shape == (B, 43, 512), t.shape
            seq.append(t[i:i + 1])
        self.seq, self.pos = seq, 0

    def draw(self, shape, device=None):
        t = self.seq[self.pos]
        self.pos += 1
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t


def _clip(data, i):
    out = {}
    for k, v in data.items():
        if torch.is_tensor(v):
            out[k] = v[i:i + 1].cpu().clone()
        elif isinstance(v, (list, tuple)):
            out[k] = [v[i].cpu() if torch.is_tensor(v[i]) else v[i]]
    return out


def _guided_setup(rg, dev, B, n_db):
    cfg = rg.synth.default_model_cfg(num_layers=8)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    database = rg.synth.SyntheticDataset(n_db, seed=2025, device=dev, feat_device=dev)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=database,
                                  device=dev)
    P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
    model.load_state_dict(P)
    model.eval()
    assert model.precision == "bf16" and model.lanes >= 2
    cpu_db = oret.build_db_dicts([dict(r, text_feature=r["text_feature"].cpu()) for r in database.retrieval_samples])
    return cfg, vae_cfgs, database, model, P, cpu_db


def _guided_batch(rg, dev, B, seed, q0):
    data = rg.synth.synth_batch(B, seed=seed, device=dev)
    qs = [rg.synth.synth_query(q0 + i) for i in range(B)]
    data["discourse"] = [q["discourse"] for q in qs]
    data["prominence"] = [q["prominence"] for q in qs]
    data["text_features"] = [q["text_features"].to(dev) for q in qs]
    data["speaker_ids"] = torch.tensor([[q["speaker_id"]] * 150 for q in qs], device=dev)
    return data


def _check_clips(rg, parity, tag, clips, B, P, cfg, vae_cfgs, cpu_db, keep, tape, ex, out, ikw):
    rd = out["retrieval_dict"]
    cpu_ds = rg.synth.SyntheticDataset(0)
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    for i in clips:
        cdata = _clip(keep, i)
        ccond = dict(text_features=cdata["text_features"], discourse=cdata["discourse"], prominence=cdata["prominence"],
                     speaker_ids=cdata["speaker_ids"])
        got = {}
        ct = ClipTape(tape.record, i, B, [b for b, _, _, _ in ex])
        with torch.no_grad():
            ref = opipe.motion_diffusion_forward(
                P, cfg, vae_cfgs, odf.SpacedSchedule(), cdata, ct,
                re_dict=lambda tp: got.setdefault("re", oret.database_forward(P, vae_cfgs, cpu_db, cpu_ds, ccond,
                                                                               cdata["sample_name"], tp)), **ikw)
        assert ct.pos == len(ct.seq), "the oracle consumed exactly the clip's share of the noise"
        # retrieval: same exemplars, same spans (bit-exact bar)
        assert rd["retr_startends"][i] == got["re"]["retr_startends"][0]
        assert rd["query_startends"][i] == got["re"]["query_startends"][0]
        assert list(rd["retr_uncropped_latents"][i].keys()) == list(got["re"]["retr_uncropped_latents"][0].keys())
        assert len(rd["retr_startends"][i]) >= 1
        lat, rlat = out["prev_latentout"][i:i + 1].cpu()[:, KEEP], ref["prev_latentout"][:, KEEP]
        parity.check("%s clip %d (%d exemplars): final latent, norm ratio" % (tag, i, len(rd["retr_startends"][i])), relerr(lat, rlat), 1e-2)
        parity.check("%s clip %d: final latent, worst token row" % (tag, i), rowerr(lat, rlat), 3e-2)
        parity.check("%s clip %d: decoded translation, norm ratio" % (tag, i), relerr(out["pred_transl"][i:i + 1].cpu(), ref["pred_transl"]), 3e-2)


def test_config3_guided_b16_full_db_vs_oracle(rg, parity):
    dev = torch.device("cuda", 0)
    B, N_DB = 16, 32768
    cfg, vae_cfgs, database, model, P, cpu_db = _guided_setup(rg, dev, B, N_DB)
    data = _guided_batch(rg, dev, B, 1234, 0)
    ikw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
    tape = RecordingTape(rg.synth.NoiseTape(9))
    keep = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}   # forward re-zeroes trans in place
    out = model(**dict(data, retrieval_method="discourse", inference_kwargs=dict(ikw, noise_tape=tape)))
    torch.cuda.synchronize()
    ex = model.model.database.last_exemplars
    assert len(ex) >= B, "the synthetic queries should retrieve exemplars"
    _check_clips(rg, parity, "config 3 (synchronous forward)", (0, 7, 15), B, P, cfg, vae_cfgs, cpu_db, keep, tape, ex, out, ikw)


@pytest.mark.parametrize("lanes", [None, 8])
def test_config3_as_benchmarked_submit_flush_vs_oracle(rg, parity, lanes):
    """What bench.py times: asynchronous submission, co-batched pipeline, whole batches rotating over the batch lanes -- the
    default four (DenoiserSession seq_duo, picked by the pipeline for launches of this width: 64 workgroups of rg_seq2_kernel,
    two sequences of a kind each) and eight (seq_pairs + seq_duo: a workgroup runs two conditional sequences, then their
    classifier-free twins).  Batch n shares a lane with batches n - L and n + L: of 2 L + 1 different batches batch L has its
    exemplar inversion co-batched with an earlier batch's sampling loop and its own sampling with a later batch's inversion:
    every step-group code path and the real exemplar counts meet the independent reference here."""
    dev = torch.device("cuda", 0)
    B, N_DB = 16, 32768
    cfg, vae_cfgs, database, model, P, cpu_db = _guided_setup(rg, dev, B, N_DB)
    model.async_results = True
    assert model.batch_lanes == 4
    if lanes is not None:
        model.batch_lanes = lanes
    L = model.batch_lanes
    ikw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
    outs, keeps, tapes, exs = [], [], [], []
    for n in range(2 * L + 1):
        data = _guided_batch(rg, dev, B, 1234 + 17 * n, 100 * n)
        keeps.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()})
        tapes.append(RecordingTape(rg.synth.NoiseTape(90 + n)))
        r = model.submit(**dict(data, retrieval_method="discourse", inference_kwargs=dict(ikw, noise_tape=tapes[-1])))
        exs.append(list(model.model.database.last_exemplars))
        if r is not None:
            outs.append(r)
    outs += model.flush()
    assert len(outs) == 2 * L + 1
    for r in outs:
        model.wait_results(r)
    torch.cuda.synchronize()
    paired = {k[:2]: (s.sq.args.pairs, int(s.sq.duo)) for k, s in model._sessions.items() if s.sq is not None}
    form = (1, 1) if L == 8 else (0, 1)
    assert [v for k, v in paired.items() if k[1] == "cobatch"] and all(v == form for k, v in paired.items() if k[1] == "cobatch"), paired
    # (16 clips + their exemplars in a launch: four lanes x 64 workgroups of two sequences, or eight x 32 of two clips each)
    assert any(k[0] == "cobatch" for k in model._graphs)
    assert len(exs[L]) >= B
    assert not torch.equal(outs[0]["prev_latentout"], outs[L]["prev_latentout"])
    _check_clips(rg, parity, "config 3 as benchmarked (submit / flush, batch %d of %d)" % (L, 2 * L + 1), (0, 7, 15), B, P, cfg,
                 vae_cfgs, cpu_db, keeps[L], tapes[L], exs[L], outs[L], ikw)


def test_config2_base_b32_vs_oracle(rg, parity):
    dev = torch.device("cuda", 0)
    B = 32
    cfg = rg.synth.default_model_cfg(num_layers=8)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=False), database=None,
                                  device=dev)
    P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
    model.load_state_dict(P)
    model.eval()
    data = rg.synth.synth_batch(B, seed=1234, device=dev)
    keep = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
    tape = RecordingTape(rg.synth.NoiseTape(10))
    out = model(**dict(data, retrieval_method="discourse", inference_kwargs=dict(noise_tape=tape)))
    torch.cuda.synchronize()
    for i in (0, 31):
        ct = ClipTape(tape.record, i, B, [])
        with torch.no_grad():
            ref = opipe.motion_diffusion_forward(P, cfg, vae_cfgs, odf.SpacedSchedule(), _clip(keep, i), ct)
        lat, rlat = out["prev_latentout"][i:i + 1].cpu()[:, KEEP], ref["prev_latentout"][:, KEEP]
        parity.check("config 2 clip %d: final latent, norm ratio" % i, relerr(lat, rlat), 1e-2)
        parity.check("config 2 clip %d: final latent, worst token row" % i, rowerr(lat, rlat), 3e-2)
        parity.check("config 2 clip %d: decoded translation, norm ratio" % i, relerr(out["pred_transl"][i:i + 1].cpu(), ref["pred_transl"]), 3e-2)
