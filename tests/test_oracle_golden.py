"""CPU: the oracle (torch fp32 restatement) against golden vectors produced by the real
reference (tests/golden/make_goldens.py).  The denoiser/sampler restatement reproduces the
reference bit for bit on the machine that generated the goldens; tolerances below leave room
for a different CPU's BLAS/vector width."""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser as od, diffusion as odf, rotation as orot, vae as ovae, pipeline as opipe

KEEP = [r for r in range(43) if r not in (10, 20, 30)]


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def test_schedule_tables(golden_dir, rg):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    for sch in (odf.SpacedSchedule(), rg.schedule.Schedule()):
        assert list(g["timestep_map"]) == list(sch.timestep_map)
        assert np.array_equal(g["betas"], sch.betas)
        assert np.array_equal(g["alphas_cumprod"], sch.alphas_cumprod)
    s = rg.schedule.Schedule()
    assert np.array_equal(g["sqrt_recip_alphas_cumprod"].astype(np.float32), s.c_recip)
    assert np.array_equal(g["sqrt_recipm1_alphas_cumprod"].astype(np.float32), s.c_recipm1)
    assert len(s.timestep_map) == 50 and s.timestep_map[0] == 0 and s.timestep_map[-1] == 999


def test_rotation(golden_dir):
    g = _load(golden_dir, "rotation.npz")
    d6 = orot.matrix_to_rotation_6d(orot.axis_angle_to_matrix(g["aa_in"]))
    assert (d6 - g["d6_out"]).abs().max() <= 1e-6
    aa = orot.matrix_to_axis_angle(orot.rotation_6d_to_matrix(g["d6_in"]))
    assert (aa - g["aa_out"]).abs().max() <= 1e-5
    rt = orot.matrix_to_axis_angle(orot.rotation_6d_to_matrix(g["d6_out"]))
    assert (rt - g["aa_roundtrip"]).abs().max() <= 1e-4


@pytest.mark.parametrize("tag,L", [("L2_allenc", 2), ("L8_encdec", 8)])
def test_denoiser_forward(golden_dir, rg, tag, L):
    torch.set_num_threads(8)
    synth = rg.synth
    cfg = synth.default_model_cfg(num_layers=L)
    P = synth.synth_denoiser_state(0, cfg)
    g = _load(golden_dir, "denoiser_%s.npz" % tag)
    B = 2
    data = synth.synth_batch(B, seed=1234)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(99)).standard_normal((B, 43, 512)).astype(np.float32))
    mm = torch.ones(B, 43)
    mm[:, [10, 21, 32]] = 0
    xf = od.encode_conditions(P, data["word"], data["audio"], data["speaker_ids"])
    qm_real = od.make_query_masks(mm)
    qm_ones = {k: torch.ones_like(v) for k, v in qm_real.items()}
    for t in (999, 99) if L == 8 else (999, 514, 99, 0):
        ts = torch.full((B,), t, dtype=torch.long)
        out = od.denoiser_forward(P, cfg, x, ts, mm, xf, qm_ones)
        assert (out - g["den_ones_t%d" % t]).abs().max() <= 2e-4
        out = od.denoiser_forward(P, cfg, x, ts, mm, xf, qm_real)
        # rows 10/20/30 carry LayerNorm of a -1e6-offset row: platform-dependent rounding
        assert (out - g["den_real_t%d" % t])[:, KEEP].abs().max() <= 5e-2


def test_vae_encode_decode(golden_dir, rg):
    torch.set_num_threads(8)
    synth = rg.synth
    tag, arch, vkw = "L8_encdec", "encoder_decoder", dict(num_layers=4, ff_size=512)
    cfg = synth.default_model_cfg(num_layers=8)
    vae_cfgs = synth.synth_vae_cfgs(decoder_arch=arch, **vkw)
    P = {}
    for i, part in enumerate(synth.PARTS):
        P.update(synth.synth_vae_state(101 + i, vae_cfgs[part], prefix="gesture_rep_encoder.%s_vae." % part))
    g = _load(golden_dir, "vae_%s.npz" % tag)
    B = 2
    data = synth.synth_batch(B, seed=1234)
    tape = synth.NoiseTape(555)
    lat, mask = ovae.gesture_encode(P, vae_cfgs, data, [tape.draw((B * 10, 1, 512)) for _ in range(4)])
    assert (lat - g["enc_latent"]).abs().max() <= 1e-3
    assert mask[:, [10, 21, 32]].sum() == 0 and mask.sum() == B * 40
    gg = np.random.Generator(np.random.PCG64(99))
    gg.standard_normal((B, 43, 512))
    zl = torch.from_numpy(gg.standard_normal((B, 43, 512)).astype(np.float32))
    dec = ovae.gesture_decode(P, vae_cfgs, zl)
    for nm, a in zip(("upper", "lower", "face", "hands", "transl", "exps", "contact"), dec):
        assert (a - g["dec_" + nm]).abs().max() <= 5e-3, nm


def test_vae_allenc(golden_dir, rg):
    torch.set_num_threads(8)
    synth = rg.synth
    vae_cfgs = synth.synth_vae_cfgs(decoder_arch="all_encoder")
    P = {}
    for i, part in enumerate(synth.PARTS):
        P.update(synth.synth_vae_state(101 + i, vae_cfgs[part], prefix="gesture_rep_encoder.%s_vae." % part))
    g = _load(golden_dir, "vae_L2_allenc.npz")
    B = 2
    data = synth.synth_batch(B, seed=1234)
    tape = synth.NoiseTape(555)
    lat, _ = ovae.gesture_encode(P, vae_cfgs, data, [tape.draw((B * 10, 1, 512)) for _ in range(4)])
    assert (lat - g["enc_latent"]).abs().max() <= 1e-3
    gg = np.random.Generator(np.random.PCG64(99))
    gg.standard_normal((B, 43, 512))
    zl = torch.from_numpy(gg.standard_normal((B, 43, 512)).astype(np.float32))
    dec = ovae.gesture_decode(P, vae_cfgs, zl)
    for nm, a in zip(("upper", "lower", "face", "hands", "transl", "exps", "contact"), dec):
        assert (a - g["dec_" + nm]).abs().max() <= 5e-3, nm


@pytest.mark.parametrize("rtag,ikw,need_re", [
    ("base", dict(), False),
    ("guided", dict(use_inversion=True, insertion_guidance=True, guidance_iters=[0] * 25 + list(range(25)),
                    guidance_lr=0.1), True),
    ("prevonly", dict(use_prev_latent=True), False),
])
def test_end_to_end_L2(golden_dir, rg, rtag, ikw, need_re):
    torch.set_num_threads(8)
    synth = rg.synth
    cfg = synth.default_model_cfg(num_layers=2)
    vae_cfgs = synth.synth_vae_cfgs(decoder_arch="all_encoder")
    P = synth.synth_full_state(0, cfg, vae_cfgs)
    g = _load(golden_dir, "e2e_L2_allenc.npz")
    B = 2
    data = synth.synth_batch(B, seed=4321)
    re_dict = opipe.synthetic_re_dict(B, seed=77) if need_re else None
    prev = None
    if ikw.get("use_prev_latent"):
        prev = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal((B, 43, 512)).astype(np.float32))
    with torch.no_grad():
        out = opipe.motion_diffusion_forward(P, cfg, vae_cfgs, odf.SpacedSchedule(), data, synth.NoiseTape(2024),
                                             re_dict=re_dict, prev_latent=prev, **ikw)
    lat, ref = out["prev_latentout"], g["%s_prev_latentout" % rtag]
    assert ((lat - ref)[:, KEEP].norm() / ref[:, KEEP].norm()) <= 2e-2
    for k in ("pred_transl", "pred_exps"):
        assert ((out[k] - g["%s_%s" % (rtag, k)]).norm() / g["%s_%s" % (rtag, k)].norm()) <= 2e-2


def test_ddpm_and_visualize_inversion_vs_reference(golden_dir, rg):
    """inference_type="ddpm" (p_sample_loop, fixed_large variance) and inference_kwargs["visualize_inversion"]
    (decoded inversion levels + DDIM reconstructions): oracle against the real reference's outputs
    (tests/golden/make_ddpm_golden.py)."""
    torch.set_num_threads(8)
    g = _load(golden_dir, "e2e_ddpm_L2.npz")
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
    sch = odf.SpacedSchedule()
    with torch.no_grad():
        out = opipe.motion_diffusion_forward(P, cfg, vae_cfgs, sch, rg.synth.synth_batch(2, seed=4321), rg.synth.NoiseTape(2024),
                                             inference_type="ddpm")
    for k in ("prev_latentout", "pred_upper", "pred_hands", "pred_transl", "pred_exps"):
        assert (out[k] - g["ddpm_" + k]).abs().max() <= 5e-4, k
    with torch.no_grad():
        out = opipe.motion_diffusion_forward(P, cfg, vae_cfgs, sch, rg.synth.synth_batch(2, seed=4321), rg.synth.NoiseTape(2024),
                                             re_dict=opipe.synthetic_re_dict(2, seed=77), use_inversion=True, visualize_inversion=True)
    assert (out["prev_latentout"] - g["visinv_prev_latentout"]).abs().max() <= 5e-4
    assert out["inverted_output_upper"].shape == (4, 50, 150, 39) and out["reconspair_output_exps"].shape == (4, 2, 150, 100)
    assert (out["inverted_output_transl"][:, [0, 24, 49]] - g["visinv_inverted_output_transl_lv"]).abs().max() <= 2e-3
    assert (out["reconspair_output_transl"] - g["visinv_reconspair_output_transl"]).abs().max() <= 2e-3
