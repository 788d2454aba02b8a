"""GPU: the four body-part VAEs AT FULL DEPTH against the CPU oracle (oracle/vae.py, pinned on the reference's goldens in
test_oracle_golden.py), through every launch path the product uses:

  * part streams  -- four concurrent launch chains (synchronous forwards),
  * grouped       -- one chain, layer i of all four parts as ONE grouped launch (capi.OpRecorder: rg_gemm_grouped,
                     rg_layernorm_grouped, rg_mha_bf16_grouped, ...; the asynchronous pipeline),
  * single chain  -- one chain, launches one by one,
  * fused stacks  -- the whole encoder stack of a part as ONE launch (rg_venc_forward, csrc/rg_venc.hip: two chunk sequences
                     per workgroup) and the all_encoder decoder stack as one launch per block (rg_vdec_step, csrc/rg_vdec.hip:
                     four 40-row tiles per 160-token sequence) -- the defaults where the shape supports them; `*_unfused` = the
                     per-op chains instead,

at the sizes of BASELINE config 3: 16 clips (encode + decode) and 48 exemplars (encode), 8 layers, both decoder
architectures (gesture_vae.py:124-239, detr_utils.py:101-210).  Until this file the grouped launches were only compared
with the ungrouped HIP path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ("upper", "lower", "face", "hands", "transl", "exps", "contact")
ROT = ("upper", "lower", "face", "hands")


def relerr(a, b):
    return ((a - b).norm() / b.norm()).item()


def rot_relerr(a, b):
    from oracle import rotation as orot
    ma, mb = orot.axis_angle_to_matrix(a.reshape(-1, 3)), orot.axis_angle_to_matrix(b.reshape(-1, 3))
    return ((ma - mb).norm() / mb.norm()).item()


def _state(rg, vae_cfgs):
    P = {}
    for i, part in enumerate(rg.synth.PARTS):
        P.update(rg.synth.synth_vae_state(101 + i, vae_cfgs[part], prefix="gesture_rep_encoder.%s_vae." % part))
    return P


PATHS = {"part_streams": dict(part_streams=True), "grouped": dict(part_streams=False, grouped=True),
         "single_chain": dict(part_streams=False, grouped=False),
         # the encoder stacks as per-op launch chains instead of the fused one-launch encoder (rg_venc_forward)
         "part_streams_unfused": dict(part_streams=True, fused_encoder=False, fused_decoder=False),
         "grouped_unfused": dict(part_streams=False, grouped=True, fused_encoder=False, fused_decoder=False)}


@pytest.fixture(scope="module")
def oracle_results(rg):
    """The oracle's answers, once per architecture (CPU, fp32): encode of 16 clips and of 48 exemplars, decode of 16."""
    from oracle import vae as ovae
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    out = {}
    for arch in ("all_encoder", "encoder_decoder"):
        vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch=arch)           # 8 layers, 512 wide, FF 1024
        P = _state(rg, vae_cfgs)
        res = dict(vae_cfgs=vae_cfgs, P=P, enc={}, eps={}, data={})
        with torch.no_grad():
            data = rg.synth.synth_batch(48, seed=7748)
            tape = rg.synth.NoiseTape(648)
            eps = [tape.draw((48 * 10, 1, 512)) for _ in range(4)]
            d = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
            lat, mask = ovae.gesture_encode(P, vae_cfgs, d, eps)          # one oracle pass: clips never interact,
            for B in (16, 48):                                            # the first 16 clips ARE the B = 16 case
                res["enc"][B] = (lat[:B], mask[:B], d["trans"][:B])
                res["eps"][B] = [e[:B * 10] for e in eps]
                res["data"][B] = {k: (v[:B] if torch.is_tensor(v) else v) for k, v in data.items()}
            g = np.random.Generator(np.random.PCG64(99))
            z = torch.from_numpy(g.standard_normal((16, 43, 512)).astype(np.float32))
            z[:, [10, 21, 32]] = 0
            res["z"], res["dec"] = z, ovae.gesture_decode(P, vae_cfgs, z)
        out[arch] = res
    return out


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("arch", ["all_encoder", "encoder_decoder"])
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_full_depth_vaes_against_the_oracle(rg, parity, oracle_results, precision, arch, path):
    o = oracle_results[arch]
    gre = rg.vae.GestureRepEncoder(o["P"], o["vae_cfgs"], "cuda", precision, **PATHS[path])
    tag = "VAE L8 %s %s %s" % (arch, precision, path)
    for B in (16, 48):
        data = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in o["data"][B].items()}
        lat, mask = gre.encode(data["motion_upper"], data["motion_lower"], data["motion_face"], data["motion_hands"],
                               data["trans"], data["facial"], data["contact"], data["motion_mask"], o["eps"][B])
        ref_lat, ref_mask, ref_trans = o["enc"][B]
        parity.check("%s: encode B=%d latent vs oracle" % (tag, B), relerr(lat.cpu(), ref_lat), 1e-2 if precision == "bf16" else 5e-5)
        assert torch.equal(mask.cpu(), ref_mask)
        assert torch.allclose(data["trans"], ref_trans, atol=1e-6)            # the in-place re-zeroing of trans x / z
        assert lat[:, [10, 21, 32]].abs().max() == 0                           # separator rows
    dec = gre.decode(o["z"].cuda())
    for nm, a, r in zip(NAMES, dec, o["dec"]):
        e = rot_relerr(a.cpu(), r) if nm in ROT else relerr(a.cpu(), r)
        parity.check("%s: decode B=16 %s vs oracle" % (tag, nm), e, 3e-2 if precision == "bf16" else 2e-4)


@pytest.mark.parametrize("arch", ["all_encoder", "encoder_decoder"])
def test_launch_paths_agree_bit_for_bit(rg, oracle_results, arch):
    """grouped == single chain == part streams (same kernels on the same rows in another order of issue)."""
    o = oracle_results[arch]
    outs = {}
    for path, kw in PATHS.items():
        gre = rg.vae.GestureRepEncoder(o["P"], o["vae_cfgs"], "cuda", "bf16", **kw)
        data = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in o["data"][16].items()}
        lat, _ = gre.encode(data["motion_upper"], data["motion_lower"], data["motion_face"], data["motion_hands"],
                            data["trans"], data["facial"], data["contact"], data["motion_mask"], o["eps"][16])
        outs[path] = [lat] + list(gre.decode(o["z"].cuda()))
    torch.cuda.synchronize()
    for path, base in (("grouped", "part_streams"), ("single_chain", "part_streams"), ("grouped_unfused", "part_streams_unfused")):
        for i, (a, b) in enumerate(zip(outs[path], outs[base])):
            assert torch.equal(a, b), (path, i, (a - b).abs().max().item())
    # fused and unfused encoders: the same arithmetic class (bf16 operands, fp32 accumulation), another order of operations
    for i, (a, b) in enumerate(zip(outs["grouped"], outs["grouped_unfused"])):
        if i in (0, 5, 6, 7):        # latent, transl, exps, contact (the rotations are compared through matrices above)
            assert ((a - b).norm() / b.norm()).item() <= 1.5e-2, i


@pytest.mark.parametrize("kw", [dict(normalize_before=True), dict(ff_size=768, activation="relu", num_heads=8)])
def test_hyper_parameters_outside_the_fused_kernels_run_the_chains_and_meet_the_oracle(rg, parity, kw):
    """The fused VAE stacks (rg_venc_forward / rg_vdec_step) are specialised for the survey's probe hyper-parameters (post-norm,
    gelu, 4 heads, FF 1024; the reference's YAMLs are not in the repository).  A checkpoint with other values must
    resolve to the per-op chains BY ITSELF -- and that path is parity-tested here at the benchmark's batch size, so a real
    checkpoint does not meet an untested engine: 16 clips, 8 layers, encode + decode against the oracle."""
    from oracle import vae as ovae
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder", **kw)
    P = _state(rg, vae_cfgs)
    gre = rg.vae.GestureRepEncoder(P, vae_cfgs, "cuda", "bf16", part_streams=False, grouped=True)      # (defaults: fused where supported)
    assert all(v.venc is None and v.vdec is None for v in gre.vaes.values()), "these shapes are not what the fused kernels are built for"
    B = 16
    data = rg.synth.synth_batch(B, seed=7749)
    tape = rg.synth.NoiseTape(649)
    eps = [tape.draw((B * 10, 1, 512)) for _ in range(4)]
    with torch.no_grad():
        d = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
        ref_lat, ref_mask = ovae.gesture_encode(P, vae_cfgs, d, eps)
        g = np.random.Generator(np.random.PCG64(98))
        z = torch.from_numpy(g.standard_normal((B, 43, 512)).astype(np.float32))
        z[:, [10, 21, 32]] = 0
        ref_dec = ovae.gesture_decode(P, vae_cfgs, z)
    d = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
    lat, mask = gre.encode(d["motion_upper"], d["motion_lower"], d["motion_face"], d["motion_hands"], d["trans"], d["facial"],
                           d["contact"], d["motion_mask"], eps)
    tag = "VAE L8 all_encoder bf16 chains, " + ", ".join("%s=%s" % kv for kv in kw.items())
    parity.check("%s: encode B=16 latent vs oracle" % tag, relerr(lat.cpu(), ref_lat), 1e-2)
    assert torch.equal(mask.cpu(), ref_mask)
    # pre-norm: the decoder's output layer reads the UN-normalised residual stream, and bf16 operands cost more there (measured
    # 2-5e-2 on the decoded rotations at 8 layers with these random weights; the same chains in precision="fp32" are at 2-5e-5,
    # profiles/dbg/prenorm_check.py): a looser bf16 bar for that architecture, stated instead of hidden
    bar = 6e-2 if kw.get("normalize_before") else 3e-2
    for nm, a, r in zip(NAMES, gre.decode(z.cuda()), ref_dec):
        e = rot_relerr(a.cpu(), r) if nm in ROT else relerr(a.cpu(), r)
        parity.check("%s: decode B=16 %s vs oracle" % (tag, nm), e, bar)
