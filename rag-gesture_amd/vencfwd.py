"""Host side of the sequence-stationary body-part VAE encoder (include/rg_gesture.h: rg_venc_forward; kernel:
csrc/rg_venc.hip): the weight / parameter streams of one TransformerVAE encoder stack, packed in the order and MFMA-fragment
layout the kernel's waves consume them.

reference: mogen/models/transformers/gesture_vae.py:111-193 (`encode_to_dist`), mogen/models/utils/detr_utils.py:101-152
(SkipTransformerEncoder: input blocks, middle block, [Linear(2 D -> D) on cat(x, skip), output block] x nb, final norm),
:335-393 (TransformerEncoderLayer.forward_post: x = norm1(x + MHA(x)); x = norm2(x + linear2(act(linear1(x))))).

Unit GEMMs (512 x 512) per block, in stream order:
    [SKIP_X, SKIP_S] (output blocks only: the two K-halves of the skip linear), Q (x 1 / sqrt(head_dim)), K, V, OUT,
    FF1_0, FF2_0, FF1_1, FF2_1
"""
import ctypes
import math

import torch

from . import capi
from .seqfwd import pack_unit

_vp = ctypes.c_void_p


class VencArgs(ctypes.Structure):
    _fields_ = [("wstream", _vp), ("pstream", _vp), ("x", _vp), ("out", _vp), ("xbuf", _vp), ("dump", _vp),
                ("nseq", ctypes.c_int), ("S", ctypes.c_int), ("nb", ctypes.c_int), ("dump_block", ctypes.c_int)]


def num_blocks(num_layers):
    """detr_utils.py:108-110: an even num_layers is rounded up to odd; blocks per side = (num_layers - 1) // 2."""
    if num_layers % 2 == 0:
        num_layers += 1
    return (num_layers - 1) // 2


def supported(vcfg, precision):
    """Shapes rg_venc_forward is specialised for (the probe hyper-parameters of SURVEY F11; anything else runs the generic
    launch chain)."""
    return (precision == "bf16" and vcfg["latent_dim"] == 512 and vcfg["num_heads"] == 4 and vcfg["ff_size"] == 1024
            and vcfg["transformer_activation"] == "gelu" and not vcfg["transformer_normalize_before"]
            and vcfg["frame_chunk_size"] + 2 <= 24 and 1 <= num_blocks(vcfg["num_layers"]) <= 8)


class VencStreams:
    """Device-resident streams of one encoder stack: wstream bf16 [NU][8][64][64][8], pstream fp32 [NU + 1][8][4][64]."""

    def __init__(self, sd, name, num_layers, heads, dev):
        """sd: the VAE's state dict (reference key names, un-prefixed); name: "encoder"."""
        D = 512
        nb = self.nb = num_blocks(num_layers)
        NU = self.NU = 8 * (2 * nb + 1) + 2 * nb
        f = lambda k: sd[k].detach().to(dev, torch.float32)
        W = torch.zeros(NU, 8, 64, 64, 8, device=dev, dtype=torch.bfloat16)
        P = torch.zeros(NU + 1, 4, D, device=dev, dtype=torch.float32)
        scale = 1.0 / math.sqrt(D // heads)
        blocks = ([("%s.input_blocks.%d" % (name, i), None) for i in range(nb)] + [(name + ".middle_block", None)] +
                  [("%s.output_blocks.%d" % (name, i), "%s.linear_blocks.%d" % (name, i)) for i in range(nb)])
        u = 0
        for blk, skip in blocks:
            if skip is not None:
                ws, bs = f(skip + ".weight"), f(skip + ".bias")          # [D, 2 D]: columns [0, D) act on x, [D, 2 D) on the skip
                W[u], W[u + 1] = pack_unit(ws[:, :D].contiguous()), pack_unit(ws[:, D:].contiguous())
                P[u, 0] = bs
                u += 2
            wi, bi = f(blk + ".self_attn.in_proj_weight"), f(blk + ".self_attn.in_proj_bias")
            W[u], P[u, 0] = pack_unit((wi[:D] * scale).contiguous()), bi[:D] * scale
            W[u + 1], P[u + 1, 0] = pack_unit(wi[D:2 * D].contiguous()), bi[D:2 * D]
            W[u + 2], P[u + 2, 0] = pack_unit(wi[2 * D:].contiguous()), bi[2 * D:]
            W[u + 3], P[u + 3, 0] = pack_unit(f(blk + ".self_attn.out_proj.weight")), f(blk + ".self_attn.out_proj.bias")
            P[u + 3, 1], P[u + 3, 2] = f(blk + ".norm1.weight"), f(blk + ".norm1.bias")
            w1, b1, w2 = f(blk + ".linear1.weight"), f(blk + ".linear1.bias"), f(blk + ".linear2.weight")
            for j in range(2):
                W[u + 4 + 2 * j], P[u + 4 + 2 * j, 0] = pack_unit(w1[j * D:(j + 1) * D].contiguous()), b1[j * D:(j + 1) * D]
                W[u + 5 + 2 * j] = pack_unit(w2[:, j * D:(j + 1) * D].contiguous())
            P[u + 5, 0] = f(blk + ".linear2.bias")
            P[u + 5, 1], P[u + 5, 2] = f(blk + ".norm2.weight"), f(blk + ".norm2.bias")
            u += 8
        capi.require(u == NU, "internal: unit count")
        P[NU, 0], P[NU, 1] = f(name + ".norm.weight"), f(name + ".norm.bias")
        self.wstream = W
        self.pstream = P.view(NU + 1, 4, 8, 64).permute(0, 2, 1, 3).contiguous()       # per wave: [4 vectors][64 features]


class VdecArgs(ctypes.Structure):
    _fields_ = [("wstream", _vp), ("pstream", _vp), ("x", _vp), ("pos", _vp), ("qimg", _vp), ("kbuf", _vp), ("vt", _vp),
                ("xbuf", _vp), ("dump", _vp), ("nseq", ctypes.c_int), ("nb", ctypes.c_int), ("step", ctypes.c_int),
                ("pad_", ctypes.c_int)]


def decoder_supported(vcfg, precision, n_chunks):
    """Shapes rg_vdec_step is specialised for: the all_encoder decoder over n_chunks + num_frames = 160 tokens."""
    return (precision == "bf16" and vcfg["decoder_arch"] == "all_encoder" and vcfg["latent_dim"] == 512 and vcfg["num_heads"] == 4
            and vcfg["ff_size"] == 1024 and vcfg["transformer_activation"] == "gelu" and not vcfg["transformer_normalize_before"]
            and n_chunks + vcfg["num_frames"] == 160 and 1 <= num_blocks(vcfg["num_layers"]) <= 8)


class VdecForward:
    """The block-fused decoder stack (include/rg_gesture.h: rg_vdec_step): 2 nb + 2 launches per decode."""

    def __init__(self, h, streams):
        self.h, self.st = h, streams

    def run(self, xseq, pos, nseq, dumps=None):
        """xseq fp32 [nseq * 160, 512] (contiguous; updated IN PLACE to the stack's output behind its final norm), pos fp32
        [nseq * 160, 512] = xseq + positional table (gesture_vae.py:221-224).  dumps: optional list that receives the state
        every launch starts its projection part from (diagnostics)."""
        for t in (xseq, pos):
            if not (t.is_contiguous() and t.dtype == torch.float32 and t.numel() == nseq * 160 * 512):
                raise capi.RgError("rg_vdec_step: x / pos must be contiguous fp32 [nseq * 160, 512] tensors")
        dev, nb = xseq.device, self.st.nb
        qimg = torch.empty(4 * nseq * 48 * 1024, device=dev, dtype=torch.uint8)
        kbuf = torch.empty(2 * nseq * 160 * 512, device=dev, dtype=torch.bfloat16)     # double-buffered by launch parity
        vt = torch.empty(2 * nseq * 512 * 160, device=dev, dtype=torch.bfloat16)
        xbuf = torch.empty(4 * nseq * nb * 8 * 12 * 64 * 4, device=dev)
        for step in range(2 * nb + 2):
            a = VdecArgs()
            a.wstream, a.pstream = self.st.wstream.data_ptr(), self.st.pstream.data_ptr()
            a.x, a.pos, a.qimg, a.kbuf, a.vt, a.xbuf = (t.data_ptr() for t in (xseq, pos, qimg, kbuf, vt, xbuf))
            d = None
            if dumps is not None:
                d = torch.zeros(4 * nseq, 48, 512, device=dev)
                dumps.append(d)
            a.dump = d.data_ptr() if d is not None else None
            a.nseq, a.nb, a.step = int(nseq), int(nb), step
            self.h.call("vdec_step", ctypes.byref(a), keep=(a, xseq, pos, qimg, kbuf, vt, xbuf, d))
        return xseq


class VencForward:
    """Launches of one stack."""

    def __init__(self, h, streams):
        self.h, self.st = h, streams

    def run(self, xseq, nseq, S, dump=None, dump_block=-1):
        """xseq fp32 [nseq * S, 512] (contiguous, device) -> encoder output fp32 [nseq * S, 512] behind the final norm."""
        if not (xseq.is_contiguous() and xseq.dtype == torch.float32 and xseq.numel() == nseq * S * 512):
            raise capi.RgError("rg_venc_forward: x must be a contiguous fp32 [nseq * S, 512] tensor")
        dev = xseq.device
        out = torch.empty(nseq * S, 512, device=dev)
        xbuf = torch.empty(((nseq + 1) // 2) * self.st.nb * 8 * 12 * 64 * 4, device=dev)
        a = VencArgs()
        a.wstream, a.pstream = self.st.wstream.data_ptr(), self.st.pstream.data_ptr()
        a.x, a.out, a.xbuf = xseq.data_ptr(), out.data_ptr(), xbuf.data_ptr()
        a.dump = dump.data_ptr() if dump is not None else None
        a.nseq, a.S, a.nb, a.dump_block = int(nseq), int(S), int(self.st.nb), int(dump_block)
        # (through Handle.call: recorded while the four parts' launch sequences are being recorded, see capi.OpRecorder; the
        #  tuple keeps the argument block and every tensor it points into alive until the launch is issued)
        self.h.call("venc_forward", ctypes.byref(a), keep=(a, xseq, out, xbuf, dump))
        return out
