"""Caller-side packing and the long-form driver.

CPU: the oracle restatement against golden vectors produced with the real reference's functions
(tests/golden/make_goldens.py: run_packing_goldens), and the host logic of the product modules.
GPU: the HIP kernels against the same goldens, and a three-window long-form run against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import packing as opk, rotation as orot


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "packing.npz"))


def _parts(gold):
    pm = torch.from_numpy(gold["pred_motion"])
    m = {k: gold["mask_" + k].astype(bool) for k in ("upper", "lower", "hands", "face")}
    return pm, {k: pm[..., m[k]] for k in m}, m


def rot_err(a, b):
    """axis-angle is discontinuous at |angle| = pi: compare as rotation matrices"""
    ma, mb = orot.axis_angle_to_matrix(a.reshape(-1, 3)), orot.axis_angle_to_matrix(b.reshape(-1, 3))
    return (ma - mb).abs().max().item()


def rot_close(a, b, rel=2e-5, worst=2e-3):
    """HIP vs CPU: the reference's matrix -> quaternion step takes sqrt(max(0, 1 +- m00 +- m11 +- m22)); where
    that argument is ~1e-7 a 1-ulp difference upstream (FMA contraction) becomes ~3e-4 in one component, so
    the check is a tight Frobenius-relative bound plus a loose worst-element bound."""
    ma, mb = orot.axis_angle_to_matrix(a.reshape(-1, 3)), orot.axis_angle_to_matrix(b.reshape(-1, 3))
    return ((ma - mb).norm() / mb.norm()).item() <= rel and (ma - mb).abs().max().item() <= worst


# ------------------------------------------------------------------ CPU
def test_oracle_packing_matches_reference(gold):
    pm, parts, masks = _parts(gold)
    om = opk.part_masks()
    for k in masks:
        assert np.array_equal(om[k], masks[k])
    assert torch.equal(opk.scatter_parts(parts["upper"], parts["lower"], parts["hands"], parts["face"]), pm)
    # (bit-equal on the machine that generated the goldens; sin/cos/atan2 differ in the last bit across CPUs)
    assert rot_err(opk.interp_motion(pm, 2), torch.from_numpy(gold["poses30"])) <= 1e-6
    long_m = torch.from_numpy(gold["long_motion"])
    assert rot_err(opk.interp_motion(long_m, 2), torch.from_numpy(gold["long30"])) <= 1e-6
    st, en, rem = opk.window_bounds(700)
    assert st == gold["starts_700"].tolist() and en == gold["ends_700"].tolist() and rem == en[-1] - 700
    f = opk.npz_fields(np.zeros((300, 165)), np.zeros((300, 100)), np.zeros((300, 3)))
    assert set(f) == {"betas", "poses", "expressions", "trans", "model", "gender", "mocap_frame_rate"}
    assert f["betas"].shape == (300,) and f["mocap_frame_rate"] == 30 and f["model"] == "smplx2020" and f["gender"] == "neutral"


def test_host_logic_matches_oracle(rg):
    P, L = rg.packing, rg.longform
    om, pm = opk.part_masks(), P.part_masks()
    for k in om:
        assert np.array_equal(om[k], pm[k])
    for n in (150, 151, 285, 286, 700, 1350):
        assert L.window_bounds(n) == opk.window_bounds(n)
    data = dict(text_segments=[[[[0.5, 1.0], "a"], [[9.5, 10.5], "b"], [[10.0, 12.0], "c"]]],
                discourse=[[("but", "Comparison", "x", "y", 9.2, 12.5, 10.1, 10.4), ("so", "Cause", "x", "y", 1.0, 3.0, 2.0, 2.1)]],
                prominence=[[("w", 9.1, 9.4, 0.7), ("v", 18.9, 19.2, 0.1)]],
                gesture_labels=[[dict(start=9.5, end=11.0, name="beat", word="w")]])
    got = L.window_annotations(data, 9.0, 19.0)
    want = opk.window_annotations({k: v[0] for k, v in data.items()}, 9.0, 19.0)
    assert got["text_segments"][0] == want[0] and got["discourse"][0] == want[1]
    assert got["prominence"][0] == want[2] and got["gesture_labels"][0] == want[3]
    assert len(got["text_segments"][0]) == 2 and len(got["discourse"][0]) == 1 and len(got["prominence"][0]) == 1
    d = dict(motion=torch.ones(1, 160, 165), motion_mask=torch.arange(160.0).view(1, 160), speaker_ids=torch.full((1, 160), 3),
             word=torch.ones(1, 160, 768))
    p = L.pad_tail(d, 125)
    assert p["motion"].shape == (1, 285, 165) and p["motion"][:, 160:].abs().sum() == 0 and p["word"].shape[1] == 285
    assert torch.equal(p["motion_mask"][0, 160:], d["motion_mask"][0, -125:]) and p["speaker_ids"].shape == (1, 285)
    assert L.pad_tail(d, 0) is d


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_packing_matches_reference(rg, gold):
    P = rg.packing
    pm, parts, _ = _parts(gold)
    c = lambda t: t.cuda()
    got = P.scatter_parts(c(parts["upper"]), c(parts["lower"]), c(parts["hands"]), c(parts["face"]))
    assert torch.equal(got.cpu(), pm)                                   # pure data movement: bit exact
    up = P.upsample_motion(c(pm), 2).cpu()
    assert up.shape == (1, 300, 165) and rot_close(up, torch.from_numpy(gold["poses30"]))
    for key, dim in (("facial30", 100), ("trans30", 3)):
        src = opk.interp_features(torch.from_numpy(gold[key]), 1)  # identity, keeps the tensor type
        assert src.shape[-1] == dim
    facial = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).standard_normal((2, 150, 100)).astype(np.float32))
    assert (P.upsample_features(c(facial), 2).cpu() - opk.interp_features(facial, 2)).abs().max() <= 1e-6
    # scale 3: (t + 0.5) / 3 is not exact in fp32, the source coordinate differs by 1 ulp between FMA / non-FMA builds
    assert (P.upsample_features(c(facial), 3).cpu() - opk.interp_features(facial, 3)).abs().max() <= 5e-5
    out = dict(pred_upper=c(parts["upper"]), pred_lower=c(parts["lower"]), pred_hands=c(parts["hands"]),
               pred_facepose=c(parts["face"]), pred_exps=c(facial[:1]), pred_transl=c(facial[:1, :, :3]))
    poses, expr, trans = P.pack_outputs(out)
    assert poses.shape == (1, 300, 165) and expr.shape == (1, 300, 100) and trans.shape == (1, 300, 3)
    f = P.npz_fields(poses[0], expr[0], trans[0])
    assert f["poses"].shape == (300, 165) and f["betas"].shape == (300,) and f["mocap_frame_rate"] == 30


@pytest.mark.gpu
def test_hip_overlap_blend_matches_reference(rg, gold):
    L = rg.longform
    pm, _, _ = _parts(gold)
    long_m, long_f, long_t = (torch.from_numpy(gold[k]) for k in ("long_motion", "long_facial", "long_trans"))
    # window 1 = pred_motion / first 150 frames of the long features; window 2 is recovered from the oracle inputs:
    # regenerate the generator's second window with the same seeded stream
    g = np.random.Generator(np.random.PCG64(4242))
    u = lambda *sh: torch.from_numpy(g.uniform(-0.6, 0.6, size=sh).astype(np.float32))
    up, lo, ha, fa = u(1, 150, 39), u(1, 150, 27), u(1, 150, 90), u(1, 150, 3)
    facial, trans = u(1, 150, 100), u(1, 150, 3)
    up2, lo2, ha2, fa2 = u(1, 150, 39), u(1, 150, 27), u(1, 150, 90), u(1, 150, 3)
    facial2, trans2 = u(1, 150, 100), u(1, 150, 3)
    assert torch.equal(opk.scatter_parts(up, lo, ha, fa), pm)
    pred2 = opk.scatter_parts(up2, lo2, ha2, fa2)
    c = lambda t: t.cuda()
    m, f, t = L.blend_window((c(pm), c(facial), c(trans)), (c(pred2), c(facial2), c(trans2)), 15)
    assert m.shape == (1, 285, 165)
    assert rot_close(m.cpu(), long_m)
    assert (f.cpu() - long_f).abs().max() <= 1e-6 and (t.cpu() - long_t).abs().max() <= 1e-6
    assert rot_close(rg.packing.upsample_motion(m, 2).cpu(), torch.from_numpy(gold["long30"]))


@pytest.mark.gpu
def test_longform_three_windows_vs_oracle(rg, parity):
    """300-frame sample -> windows [0,150), [135,285), [270,420) (120 frames of padding), prev-latent chaining,
    blend, 30 fps; product (fp32 mode) against the oracle pipeline driven by the same loop."""
    from oracle import pipeline as opipe, diffusion as odf
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder", num_layers=2)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs), database=None, precision="fp32")
    P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
    model.load_state_dict({"model." + k: v for k, v in P.items()})
    a, b = rg.synth.synth_batch(1, seed=11), rg.synth.synth_batch(1, seed=12)
    data = {k: torch.cat([a[k], b[k]], dim=1) for k in rg.longform.MOTION_KEYS + rg.longform.REPEAT_KEYS
            if k in a and torch.is_tensor(a[k]) and a[k].dim() >= 2 and a[k].shape[1] == 150}
    assert data["motion"].shape[1] == 300
    feats = [rg.synth.synth_batch(1, seed=100 + i) for i in range(3)]
    features = lambda cidx, t0, t1, ann: dict(audio=feats[cidx]["audio"], text_features=None)
    synth = rg.longform.LongformSynthesizer(model, overlap=15)
    tape = rg.synth.NoiseTape(31)
    got = synth.run({k: v.clone() for k, v in data.items()}, features, noise_tape=tape, with_gt=True)
    assert got["windows"] == [(0, 150), (135, 285), (270, 420)] and got["poses"].shape == (600, 165)
    # ground truth through the same blend + interpolation (longform_synthesis.py:480-520, 722-745): overlapping windows
    # of ONE sequence carry identical frames in the overlap, so the result is the 30-fps interpolation of the sample
    # (all but the last output frame, which interpolates towards the zero padding behind the sample)
    gt_want = rg.packing.upsample_motion(data["motion"].cuda().float(), 2)[0, :599].cpu()
    assert got["gt_poses"].shape == (600, 165) and rot_close(torch.from_numpy(got["gt_poses"][:599]), gt_want)
    want_f = rg.packing.upsample_features(data["facial"].cuda().float(), 2)[0, :599].cpu()
    assert (torch.from_numpy(got["gt_expressions"][:599]) - want_f).abs().max() <= 1e-5
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        rg.longform.LongformSynthesizer.save(got, tmp, raw_text="hello there")
        z = np.load(os.path.join(tmp, "full_gt_motion.npz"))
        assert sorted(z.files) == sorted(["betas", "poses", "expressions", "trans", "model", "gender", "mocap_frame_rate"])
        assert np.array_equal(np.load(os.path.join(tmp, "full_pred_motion.npz"))["poses"], got["poses"])
        assert open(os.path.join(tmp, "gt_text.txt")).read() == "hello there"
    # ---- oracle loop
    otape = rg.synth.NoiseTape(31)
    sch = odf.SpacedSchedule()
    starts, ends, rem = opk.window_bounds(300)
    od = rg.longform.pad_tail({k: v.clone() for k, v in data.items()}, rem)
    prev, so_far = None, None
    for cidx, (c0, c1) in enumerate(zip(starts, ends)):
        chunk = {k: od[k][:, c0:c1] for k in od}   # views: the in-place trans re-zeroing reaches the next window, as in the reference
        chunk["audio"] = feats[cidx]["audio"]
        o = opipe.motion_diffusion_forward(P, cfg, vae_cfgs, sch, chunk, otape, use_prev_latent=True, prev_latent=prev)
        prev = o["prev_latentout"]
        cur = (opk.scatter_parts(o["pred_upper"], o["pred_lower"], o["pred_hands"], o["pred_facepose"]), o["pred_exps"],
               o["pred_transl"])
        so_far = cur if cidx == 0 else opk.blend_window(*so_far, *cur, 15)
    want_m = opk.interp_motion(so_far[0], 2)[0, :600]
    want_f, want_t = opk.interp_features(so_far[1], 2)[0, :600], opk.interp_features(so_far[2], 2)[0, :600]
    rel = lambda x, y: ((x - y).norm() / y.norm()).item()
    gm = torch.from_numpy(got["poses"])
    ma, mb = orot.axis_angle_to_matrix(gm.reshape(-1, 3)), orot.axis_angle_to_matrix(want_m.reshape(-1, 3))
    e_m, e_f, e_t = rel(ma, mb), rel(torch.from_numpy(got["expressions"]), want_f), rel(torch.from_numpy(got["trans"]), want_t)
    # (what remains in fp32 mode is the reference's -1e6 LayerNorm quirk on rows 10/20/30)
    parity.check("longform 3 windows fp32 vs oracle loop: poses (rotation matrices)", e_m, 3e-3)
    parity.check("longform 3 windows fp32 vs oracle loop: expressions", e_f, 3e-3)
    parity.check("longform 3 windows fp32 vs oracle loop: trans", e_t, 3e-3)


def test_guidance_iters_presets(rg):
    """tools/visualize.py:74-95: the --guidance_iters names (50 respaced steps, index 49 = noisiest)."""
    p = rg.pipeline.guidance_iters_preset
    assert p("all_one") == [1] * 50 and p("all_zero") == [0] * 50 and p("all_10") == [10] * 50
    assert p("decreasing") == list(range(50)) and p("increasing") == list(range(49, -1, -1))
    assert p("drop_decreasing_till_25") == [0] * 25 + list(range(25, 50))
    assert p("step_increasing_from_25") == list(range(49, 24, -1)) + [0] * 25
    assert p("decreasing_till_25") == [0] * 25 + list(range(25))
    assert p("increasing_from_25") == list(range(24, -1, -1)) + [0] * 25
    with pytest.raises(ValueError):
        p("sometimes")


def test_save_sample_files_layout(rg, tmp_path):
    """visualize.py:449-492: directory per clip, file names, npz keyword set, the zeroed translation of the
    `_notrans` copy (host-only: numpy inputs)."""
    g = np.random.Generator(np.random.PCG64(3))
    pred = (g.standard_normal((2, 300, 165)).astype(np.float32), g.standard_normal((2, 300, 100)).astype(np.float32),
            g.standard_normal((2, 300, 3)).astype(np.float32))
    gt = tuple(x + 1 for x in pred)
    names = ["2_scott_0_1_1/0", "2_scott_0_1_1/15"]
    rg.packing.save_sample_files(str(tmp_path), names, pred, gt=gt, use_inversion=True, texts=["a b", "c"])
    for j, n in enumerate(names):
        d = tmp_path / n
        assert sorted(p.name for p in d.iterdir()) == ["gt_motion.npz", "gt_text.txt", "pred_motion.npz", "pred_motion_notrans.npz"]
        z = np.load(d / "pred_motion.npz")
        assert sorted(z.files) == sorted(["betas", "poses", "expressions", "trans", "model", "gender", "mocap_frame_rate"])
        assert z["betas"].shape == (300,) and str(z["model"]) == "smplx2020" and int(z["mocap_frame_rate"]) == 30
        assert np.array_equal(z["poses"], pred[0][j]) and np.array_equal(np.load(d / "gt_motion.npz")["trans"], gt[2][j])
        assert not np.load(d / "pred_motion_notrans.npz")["trans"].any()
        assert (d / "gt_text.txt").read_text() == ["a b", "c"][j]


@pytest.mark.gpu
def test_pack_ground_truth_is_the_interpolated_input(rg):
    d = rg.synth.synth_batch(2, seed=5)
    out = dict(motion=d["motion"], facial=d["facial"], trans=d["trans"])
    poses, expr, trans = rg.packing.pack_ground_truth(out)
    assert poses.shape == (2, 300, 165) and expr.shape == (2, 300, 100) and trans.shape == (2, 300, 3)
    assert torch.equal(poses, rg.packing.upsample_motion(d["motion"].cuda().float(), 2))
    assert torch.equal(expr, rg.packing.upsample_features(d["facial"].cuda().float(), 2))
