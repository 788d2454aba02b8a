"""Exemplar-count bucketing (pipeline.E_BUCKET): a lane's E exemplars run as a batch of E rounded up to a multiple of 4
(padding = copies of exemplar 0, inverted and dropped), so that sessions and graphs exist per bucket, not per count.  The
padding must not change any result, and batches whose counts fall into the same bucket must share their graphs."""
import importlib

import numpy as np
import pytest
import torch

import oracle.pipeline as opipe

pytestmark = pytest.mark.gpu
KEYS = ("pred_upper", "pred_hands", "pred_transl", "prev_latentout")
SPANS = ((2, 5, 1, 4), (6, 8, 7, 9), (0, 2, 5, 7))


@pytest.fixture(scope="module")
def rg():
    return importlib.import_module("rag-gesture_amd")


def _re_dict(B, seed, counts):
    """synthetic retrieval result with counts[b] exemplars for clip b"""
    full = opipe.synthetic_re_dict(B, seed=seed, exemplars=SPANS)
    for b in range(B):
        for key in ("retr_startends", "query_startends", "retr_uncropped_latents"):
            full[key][b] = {q: v for q, v in full[key][b].items() if q < counts[b]}
    return full


def test_padding_is_invisible_and_buckets_share_graphs(rg):
    pl = rg.pipeline
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs), device="cuda")
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    B = 4                                    # two lanes of two clips
    cases = [(1, 2, 2, 1), (2, 1, 3, 0), (1, 1, 1, 3), (3, 3, 3, 3)]      # lane totals: 3|3, 3|3, 2|4, 6|6

    def run(i, counts):
        data = rg.synth.synth_batch(B, seed=70 + i)
        data["re_dict"] = _re_dict(B, 500 + i, counts)
        ikw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=[2] * 25 + [0] * 25, guidance_lr=0.1,
                   noise_tape=rg.synth.NoiseTape(90 + i))
        out = model(**dict(data, retrieval_method="discourse", inference_kwargs=ikw))
        torch.cuda.synchronize()
        return {k: out[k].clone() for k in KEYS}

    got = [run(i, c) for i, c in enumerate(cases)]
    inv_graphs = sorted(k[1] for k in model._graphs if k[0] == "invert")
    assert inv_graphs == [4, 4, 8, 8], inv_graphs          # buckets 4 and 8 on each of the two lanes, whatever the counts
    # the same batches without padding (one graph per exact count): bit-identical results
    saved = pl.E_BUCKET
    pl.E_BUCKET = 1
    try:
        model._graphs.clear(); model._sessions.clear(); model._graph_owner.clear()
        ref = [run(i, c) for i, c in enumerate(cases)]
    finally:
        pl.E_BUCKET = saved
    assert sorted(set(k[1] for k in model._graphs if k[0] == "invert")) == [2, 3, 4, 6]
    for i in range(len(cases)):
        for k in KEYS:
            assert torch.equal(got[i][k], ref[i][k]), (i, k, (got[i][k] - ref[i][k]).abs().max().item())
