"""Caller-side output packing: what tools/visualize.py does with the dict that
`model(**data)` returns before it writes the `.npz` files `tools/evaluate.py` and the renderer consume.

    visualize.py:204-213   body-part scatter     pred_motion[..., part_mask] = pred_part     -> scatter_parts
    visualize.py:262-291   15 -> 30 fps          aa -> 6D -> F.interpolate(linear) -> aa      -> upsample_motion
                                                 F.interpolate(linear) on expressions / trans -> upsample_features
    visualize.py:458-466   np.savez schema       betas[300] = 0, poses, expressions, trans,
                                                 model, gender, mocap_frame_rate = 30         -> npz_fields / save_npz

All arithmetic runs in the HIP extension (rg_scatter_joints, rg_interp_aa, rg_interp_linear); no CPU fallback.
"""
import numpy as np
import torch

from . import capi

# mogen/datasets/utils/beatx_utils.py joints_list["beat_smplx_{upper,lower,hands,face}"] as SMPL-X joint
# indices (beatx_dataset.py:82-109 turns them into the 165-column boolean masks); checked against the
# reference's table when the goldens are generated (tests/golden/make_goldens.py: run_packing_goldens)
UPPER = (3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21)
LOWER = (0, 1, 2, 4, 5, 7, 8, 10, 11)
HANDS = tuple(range(25, 55))
FACE = (22,)
N_JOINTS = 55
PARTS = (UPPER, LOWER, HANDS, FACE)   # order = rg_scatter_joints part codes 0..3


def part_masks():
    """{'upper','lower','hands','face'} -> bool[165], the reference's test_dataset.*_mask."""
    out = {}
    for name, joints in zip(("upper", "lower", "hands", "face"), PARTS):
        m = np.zeros(N_JOINTS * 3, dtype=bool)
        for j in joints:
            m[3 * j:3 * j + 3] = True
        out[name] = m
    return out


_maps = {}


def _joint_maps(dev):
    if dev not in _maps:
        part = np.full(N_JOINTS, -1, dtype=np.int32)
        idx = np.zeros(N_JOINTS, dtype=np.int32)
        for code, joints in enumerate(PARTS):
            for k, j in enumerate(joints):
                part[j], idx[j] = code, k
        _maps[dev] = (torch.from_numpy(part).to(dev), torch.from_numpy(idx).to(dev))
    return _maps[dev]


def _dev(t):
    if not t.is_cuda:
        raise capi.RgError("device tensor expected (the packing kernels have no CPU path)")
    return t.float().contiguous()


def scatter_parts(pred_upper, pred_lower, pred_hands, pred_face):
    """[B,n,39], [B,n,27], [B,n,90], [B,n,3] -> pred_motion [B,n,165] (joints of no part stay zero)."""
    up, lo, ha, fa = _dev(pred_upper), _dev(pred_lower), _dev(pred_hands), _dev(pred_face)
    B, n = up.shape[:2]
    capi.require(up.shape[-1] == 3 * len(UPPER) and lo.shape[-1] == 3 * len(LOWER) and ha.shape[-1] == 3 * len(HANDS)
                 and fa.shape[-1] == 3 * len(FACE), "body-part widths must be 39 / 27 / 90 / 3 (axis-angle, 3 per joint)")
    h = capi.get_handle(up.device.index)
    part, idx = _joint_maps(up.device)
    out = torch.empty(B, n, N_JOINTS * 3, device=up.device)
    h.call("scatter_joints", up, up.shape[-1], lo, lo.shape[-1], ha, ha.shape[-1], fa, fa.shape[-1], part, idx, out,
           B * n, N_JOINTS)
    return out


def upsample_motion(motion_aa, scale=2):
    """axis-angle [B,n,J*3] -> [B,n*scale,J*3]: 6D, linear interpolation along time, back (visualize.py:266-291)."""
    x = _dev(motion_aa)
    B, n, dim = x.shape
    out = torch.empty(B, n * scale, dim, device=x.device)
    capi.get_handle(x.device.index).call("interp_aa", x, out, B, n, dim // 3, int(scale))
    return out


def upsample_features(x, scale=2):
    """F.interpolate(x.permute(0,2,1), scale_factor=scale, mode='linear').permute(0,2,1) for [B,n,dim]."""
    x = _dev(x)
    B, n, dim = x.shape
    out = torch.empty(B, n * scale, dim, device=x.device)
    capi.get_handle(x.device.index).call("interp_linear", x, out, B, n, dim, int(scale))
    return out


def pack_outputs(output, motion_fps=15, target_fps=30):
    """The dict returned by MotionDiffusion.forward -> (poses [B,N,165], expressions [B,N,100], trans [B,N,3])
    at target_fps, device tensors (visualize.py:204-291, prediction branch)."""
    capi.require(target_fps % motion_fps == 0, "unsupported argument: requires target_fps % motion_fps == 0")
    scale = target_fps // motion_fps
    poses = scatter_parts(output["pred_upper"], output["pred_lower"], output["pred_hands"], output["pred_facepose"])
    expr, trans = _dev(output["pred_exps"]), _dev(output["pred_transl"])
    if scale != 1:
        poses, expr, trans = upsample_motion(poses, scale), upsample_features(expr, scale), upsample_features(trans, scale)
    return poses, expr, trans


def npz_fields(poses, expressions, trans, fps=30):
    """Keyword set of the reference's np.savez (visualize.py:458-466) for ONE clip."""
    a = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    return dict(betas=np.zeros(300,), poses=a(poses), expressions=a(expressions), trans=a(trans), model="smplx2020",
                gender="neutral", mocap_frame_rate=fps)


def save_npz(path, poses, expressions, trans, fps=30):
    np.savez(path, **npz_fields(poses, expressions, trans, fps))


def pack_ground_truth(output, motion_fps=15, target_fps=30):
    """The ground-truth triple of the same batch at the output rate (visualize.py:214-216, 266-291): the model returns
    its inputs, `motion` [B,n,165] axis-angle, `facial`, `trans` (x/z made relative by the forward)."""
    scale = target_fps // motion_fps
    up = lambda t: _dev(t if t.is_cuda else t.cuda())     # the inputs may still live on the host
    poses, expr, trans = up(output["motion"]), up(output["facial"]), up(output["trans"])
    if scale != 1:
        poses, expr, trans = upsample_motion(poses, scale), upsample_features(expr, scale), upsample_features(trans, scale)
    return poses, expr, trans


def save_sample_files(exp_dir, sample_names, pred, gt=None, use_inversion=False, texts=None):
    """visualize.py:449-492, one directory per clip: pred_motion.npz, pred_motion_notrans.npz (with --use_inversion:
    translation minus itself), gt_motion.npz and gt_text.txt when given.  pred / gt = (poses, expressions, trans)
    batches as returned by pack_outputs / pack_ground_truth.  Audio and rendering stay with the caller."""
    import os
    a = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    pred = tuple(a(t) for t in pred)
    gt = None if gt is None else tuple(a(t) for t in gt)
    for j, name in enumerate(sample_names):
        d = os.path.join(exp_dir, name)
        os.makedirs(d, exist_ok=True)
        save_npz(os.path.join(d, "pred_motion.npz"), pred[0][j], pred[1][j], pred[2][j])
        if use_inversion:
            save_npz(os.path.join(d, "pred_motion_notrans.npz"), pred[0][j], pred[1][j], pred[2][j] - pred[2][j])
        if gt is not None:
            save_npz(os.path.join(d, "gt_motion.npz"), gt[0][j], gt[1][j], gt[2][j])
        if texts is not None:
            with open(os.path.join(d, "gt_text.txt"), "w", encoding="utf-8") as f:
                f.write(texts[j])
