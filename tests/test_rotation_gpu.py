"""GPU: the HIP rotation kernels (rg_aa_to_6d / rg_6d_to_aa) on the reference's golden vectors
(tests/golden/rotation.npz, made by rotation_conversions.py:416-550 itself): random angles, angles of ~1e-8
(small-angle branch at 1e-6), angles within 1e-3 of pi, and +-2.5 rad -- every input family, not just the
U(-0.6, 0.6) inputs the VAE tests feed them."""
import os

import numpy as np
import pytest
import torch

from oracle import rotation as orot

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(golden_dir):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "rotation.npz")).items()}


def _aa_to_6d(rg, aa, joints):
    h = rg.capi.get_handle()
    rows = aa.shape[0] // joints
    x = aa.reshape(rows, joints * 3).cuda().contiguous()
    out = torch.full((rows, joints * 6 + 5), 7.0, device="cuda")     # a wider row: col_off / ld_out are honoured
    h.call("aa_to_6d", x, joints * 3, out, out.shape[1], 2, rows, joints)
    torch.cuda.synchronize()
    assert torch.all(out[:, :2] == 7.0) and torch.all(out[:, 2 + joints * 6:] == 7.0)
    return out[:, 2:2 + joints * 6].reshape(-1, 6).cpu()


def _6d_to_aa(rg, d6, joints):
    h = rg.capi.get_handle()
    rows = d6.shape[0] // joints
    x = torch.zeros(rows, joints * 6 + 3, device="cuda")
    x[:, 3:] = d6.reshape(rows, joints * 6).cuda()
    out = torch.empty(rows, joints * 3, device="cuda")
    h.call("6d_to_aa", x, x.shape[1], 3, out, joints * 3, rows, joints)
    torch.cuda.synchronize()
    return out.reshape(-1, 3).cpu()


@pytest.mark.parametrize("joints", [1, 4, 32])
def test_aa_to_6d_on_reference_goldens(rg, g, joints):
    d6 = _aa_to_6d(rg, g["aa_in"], joints)
    assert (d6 - g["d6_out"]).abs().max() <= 2e-6


@pytest.mark.parametrize("joints", [1, 4, 32])
def test_6d_to_aa_on_reference_goldens(rg, g, joints):
    aa = _6d_to_aa(rg, g["d6_in"], joints)
    # same rotation (the axis-angle of a rotation by ~pi is defined up to the sign of the axis) ...
    Rk, Rr = orot.axis_angle_to_matrix(aa), orot.axis_angle_to_matrix(g["aa_out"])
    assert (Rk - Rr).abs().max() <= 2e-5
    # ... and the same vector away from pi (matrix -> quaternion -> axis-angle amplifies the fp32 rounding of the matrix
    # in the small components of the axis: measured 2.8e-5 on one component of a 2.36 rad rotation)
    ang = g["aa_out"].norm(dim=-1)
    far = (ang - np.pi).abs() > 1e-2
    assert far.sum() >= 64
    assert (aa[far] - g["aa_out"][far]).abs().max() <= 1e-4


def test_roundtrip_on_reference_goldens(rg, g):
    """aa -> 6D -> aa on the golden inputs (near-zero, near-pi, +-2.5 rad families included)."""
    aa = _6d_to_aa(rg, _aa_to_6d(rg, g["aa_in"], 8), 8)
    Rk, Rr = orot.axis_angle_to_matrix(aa), orot.axis_angle_to_matrix(g["aa_roundtrip"])
    assert (Rk - Rr).abs().max() <= 5e-5
    ang = g["aa_roundtrip"].norm(dim=-1)
    far = (ang - np.pi).abs() > 1e-2
    assert (aa[far] - g["aa_roundtrip"][far]).abs().max() <= 1e-4
    tiny = g["aa_in"].norm(dim=-1) < 1e-6
    assert tiny.any() and aa[tiny].abs().max() <= 1e-6
