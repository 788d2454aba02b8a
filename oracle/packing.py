"""Oracle: caller-side output packing and the long-form window driver (test infrastructure, see
oracle/__init__.py).

Restates the inline code of tools/visualize.py:204-291, 458-466 (body-part scatter, 15 -> 30 fps
interpolation in 6D, .npz schema) and tools/longform_synthesis.py:262-287 (window split / tail padding),
:300-377 (per-window slicing of the annotations), :431-476 (overlap blend), :714-741 (final interpolation).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import rotation as rot

# mogen/datasets/utils/beatx_utils.py joints_list["beat_smplx_{upper,lower,hands,face}"] as SMPL-X joint indices
UPPER = [3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21]
LOWER = [0, 1, 2, 4, 5, 7, 8, 10, 11]
HANDS = list(range(25, 55))
FACE = [22]
N_JOINTS = 55


def part_masks():
    """The four boolean masks over the 165 axis-angle columns (beatx_dataset.py:82-109)."""
    out = {}
    for name, joints in (("upper", UPPER), ("lower", LOWER), ("hands", HANDS), ("face", FACE)):
        m = np.zeros(N_JOINTS * 3, dtype=bool)
        for j in joints:
            m[3 * j:3 * j + 3] = True
        out[name] = m
    return out


def scatter_parts(pred_upper, pred_lower, pred_hands, pred_face):
    """visualize.py:208-213"""
    m = part_masks()
    out = torch.zeros(*pred_upper.shape[:-1], N_JOINTS * 3)
    out[..., m["upper"]] = pred_upper
    out[..., m["lower"]] = pred_lower
    out[..., m["hands"]] = pred_hands
    out[..., m["face"]] = pred_face
    return out


def interp_features(x, scale):
    """F.interpolate(x.permute(0,2,1), scale_factor=scale, mode='linear').permute(0,2,1)"""
    return F.interpolate(x.permute(0, 2, 1), scale_factor=float(scale), mode="linear").permute(0, 2, 1)


def interp_motion(aa, scale):
    """visualize.py:266-291: aa [B,n,165] -> [B,n*scale,165] through 6D"""
    bs, n, dim = aa.shape
    nj = dim // 3
    d6 = rot.matrix_to_rotation_6d(rot.axis_angle_to_matrix(aa.reshape(bs, n, nj, 3))).reshape(bs, n, nj * 6)
    d6 = interp_features(d6, scale)
    return rot.matrix_to_axis_angle(rot.rotation_6d_to_matrix(d6.reshape(bs, n * scale, nj, 6))).reshape(bs, n * scale, nj * 3)


def npz_fields(poses, expressions, trans):
    """visualize.py:458-466 (np.savez keyword set)"""
    return dict(betas=np.zeros(300,), poses=np.asarray(poses), expressions=np.asarray(expressions), trans=np.asarray(trans),
                model="smplx2020", gender="neutral", mocap_frame_rate=30)


# ------------------------------------------------------------------ long-form driver
def window_bounds(sample_len, seqlen=150, overlap=15):
    """longform_synthesis.py:262-265: (starts, ends, remainder of zero padding)"""
    hop = seqlen - overlap
    starts = [0] + list(range(hop, sample_len, hop))
    ends = [s + seqlen for s in starts]
    return starts, ends, max(0, ends[-1] - sample_len)


def window_annotations(data, t0, t1):
    """longform_synthesis.py:333-377: annotations fully inside [t0, t1], shifted to window time."""
    segs = [[[s[0][0] - t0, s[0][1] - t0], s[1]] for s in data.get("text_segments", []) if s[0][0] >= t0 and s[0][1] <= t1]
    disc = [(d[0], d[1], d[2], d[3], d[4] - t0, d[5] - t0, d[6] - t0, d[7] - t0) for d in data.get("discourse", [])
            if d[4] >= t0 and d[5] <= t1]
    prom = [(p[0], p[1] - t0, p[2] - t0, p[3]) for p in data.get("prominence", []) if p[1] >= t0 and p[2] <= t1]
    labels = [dict(start=g["start"] - t0, end=g["end"] - t0, name=g["name"], word=g["word"])
              for g in data.get("gesture_labels", []) if g["start"] >= t0 and g["end"] <= t1]
    return segs, disc, prom, labels


def blend_window(prev_motion, prev_facial, prev_trans, motion, facial, trans, overlap):
    """longform_synthesis.py:431-476 (prediction branch): returns the extended (motion, facial, trans)."""
    bs, n, dim = motion.shape
    nj = dim // 3
    keep_m, keep_f, keep_t = prev_motion[:, :-overlap], prev_facial[:, :-overlap], prev_trans[:, :-overlap]
    tail_m, tail_f, tail_t = prev_motion[:, -overlap:], prev_facial[:, -overlap:], prev_trans[:, -overlap:]
    m6 = rot.matrix_to_rotation_6d(rot.axis_angle_to_matrix(motion.reshape(bs, n, nj, 3))).reshape(bs, n, nj * 6)
    t6 = rot.matrix_to_rotation_6d(rot.axis_angle_to_matrix(tail_m.reshape(bs, overlap, nj, 3))).reshape(bs, overlap, nj * 6)
    wn = torch.linspace(0, 1, overlap).unsqueeze(0).unsqueeze(-1)
    wp = 1 - wn
    m6 = m6.clone()
    m6[:, :overlap] = t6 * wp + m6[:, :overlap] * wn
    facial, trans = facial.clone(), trans.clone()
    facial[:, :overlap] = tail_f * wp + facial[:, :overlap] * wn
    trans[:, :overlap] = tail_t * wp + trans[:, :overlap] * wn
    motion = rot.matrix_to_axis_angle(rot.rotation_6d_to_matrix(m6.reshape(bs, n, nj, 6))).reshape(bs, n, nj * 3)
    return torch.cat([keep_m, motion], 1), torch.cat([keep_f, facial], 1), torch.cat([keep_t, trans], 1)
