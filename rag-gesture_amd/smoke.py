"""__graft_entry__.smoke(): one small hot-path invocation on cuda:0, checked against the oracle."""
import numpy as np
import torch

from . import capi, schedule


def run():
    if not torch.cuda.is_available():
        raise capi.RgError("smoke() needs a GPU")
    from oracle import diffusion as odf  # the checker (allowed here only)

    dev = torch.device("cuda:0")
    h = capi.get_handle(0)
    sch, osch = schedule.Schedule(), odf.SpacedSchedule()
    g = np.random.Generator(np.random.PCG64(0))
    x = torch.from_numpy(g.standard_normal((2, 43, 512)).astype(np.float32))
    x0 = torch.from_numpy(g.standard_normal((2, 43, 512)).astype(np.float32))
    i = 37
    ref, _ = odf.ddim_sample(osch, lambda a, t: x0, x, i, lambda s: torch.zeros(s))
    xd, x0d = x.to(dev), x0.to(dev)
    out = torch.empty_like(xd)
    h.call("ddim_update", xd, x0d, out, capi.I64(xd.numel()), float(sch.c_recip[i]), float(sch.c_recipm1[i]),
           float(sch.c_prev_a[i]), float(sch.c_prev_b[i]))
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    if not err <= 1e-6:
        raise AssertionError("smoke: ddim_update mismatch vs oracle: %g" % err)
    print("smoke ok: ddim_update max abs err %.3g" % err)
