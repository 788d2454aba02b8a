"""Host side of the persistent denoiser forward (include/rg_gesture.h: rg_denoiser_forward;
kernel: csrc/rg_fwd.hip): the tile schedule, the per-layer pointer table and the activation buffers
of one session.

reference: raggesture.py:1041-1085 (`forward_test` up to the CFG mix), diffusion_transformer.py:620-668,
:105-127 (`DecoderLayer`).

Schedule: a sequence (T token rows; 2 per clip) runs the chain
    EMBED, L x [QKV_SA, SAOUT, Q3_CA (conditional sequences only), MIX, FF1, FF2, FFOUT], HEAD
and every stage is cut into tiles of 64 output columns (QKV_SA / Q3_CA: one head pair).  A tile waits
until ALL earlier tiles of its sequence have completed, so its descriptor only carries that count.
The two sequences of a clip share a shard (queue); inside a shard tiles are listed stage-major, which
is a topological order of the dependency graph.
"""
import ctypes

import numpy as np
import torch

EMBED, QKV_SA, SAOUT, Q3_CA, MIX, FF1, FF2, FFOUT, HEAD = range(9)
TILES = {EMBED: 8, QKV_SA: 8, SAOUT: 8, Q3_CA: 24, MIX: 8, FF1: 16, FF2: 8, FFOUT: 8, HEAD: 8}
N_SHARD = 8
SCHED_HEADER = 16   # ints in front of the tile list
_vp = ctypes.c_void_p


class FwdLayer(ctypes.Structure):
    _fields_ = [(n, _vp) for n in (
        "w_qkv", "b_qkv", "sa_g", "sa_b", "sa_sg", "sa_sb", "w_sao", "b_sao", "w_q3", "b_q3", "ca_g", "ca_b",
        "a_pre", "ca_sg", "ca_sb", "unc_tab", "w_mix", "b_mix", "w_ff1", "b_ff1", "w_ff2", "b_ff2",
        "ff_sg", "ff_sb", "w_ffo", "b_ffo")]


class FwdArgs(ctypes.Structure):
    _fields_ = [("layers", _vp), ("L", ctypes.c_int), ("B", ctypes.c_int), ("T", ctypes.c_int), ("step", ctypes.c_int),
                ("w_embed", _vp), ("b_embed", _vp), ("tbias", _vp), ("w_out", _vp), ("b_out", _vp), ("ss", _vp),
                ("x", _vp), ("src_mask", _vp), ("qmask", _vp), ("xa", _vp), ("xb", _vp), ("xc", _vp), ("head", _vp),
                ("xb_bf", _vp), ("xc_bf", _vp), ("ysa", _vp), ("yf", _vp), ("y3", _vp), ("g", _vp),
                ("st_a", _vp), ("st_b", _vp), ("st_sa", _vp), ("st_f", _vp), ("st3", _vp),
                ("sched", _vp), ("ctrl", _vp), ("stamps", _vp)]


def stage_list(L):
    st = [(EMBED, 0)]
    for l in range(L):
        st += [(t, l) for t in (QKV_SA, SAOUT, Q3_CA, MIX, FF1, FF2, FFOUT)]
    st.append((HEAD, 0))
    return st


def build_schedule(B, L):
    """int32 array: [0..8] first tile of every shard and the total, then from int 16 on one
    (type | layer << 8, sequence, tile index, completion count to wait for) per tile."""
    shards = [[] for _ in range(N_SHARD)]
    for b in range(B):
        shards[b % N_SHARD] += [b, B + b]
    done = np.zeros(2 * B, dtype=np.int64)
    tiles, starts = [], [0]
    for seqs in shards:
        for typ, l in stage_list(L):
            n = TILES[typ]
            for s in seqs:
                if typ == Q3_CA and s >= B:
                    continue
                for nt in range(n):
                    tiles.append((typ | (l << 8), s, nt, int(done[s])))
                done[s] += n
        starts.append(len(tiles))
    out = np.zeros(SCHED_HEADER + 4 * len(tiles), dtype=np.int32)
    out[:N_SHARD + 1] = starts
    out[SCHED_HEADER:] = np.asarray(tiles, dtype=np.int32).reshape(-1)
    return out


def build_stage_schedule(B, L):
    """Stage-major tile list for rg_denoiser_forward_stages: (tiles int32 [n, 4], first int32 [n_stages + 1]).
    Inside a stage the slots are dealt to the 8 shards in turn (slot p -> shard p % 8 = the XCD the workgroup lands on
    under round-robin dispatch), so the tiles of a sequence run on the same XCD in every stage and its rows are read
    from the L2 they were written to; uneven shards are padded with type-255 slots."""
    shards = [[] for _ in range(N_SHARD)]
    for b in range(B):
        shards[b % N_SHARD] += [b, B + b]
    tiles, first = [], [0]
    for typ, l in stage_list(L):
        per = []
        for seqs in shards:
            per.append([(typ | (l << 8), s, nt, 0) for s in seqs if not (typ == Q3_CA and s >= B) for nt in range(TILES[typ])])
        depth = max(len(p) for p in per)
        for k in range(depth):
            for q in range(N_SHARD):
                tiles.append(per[q][k] if k < len(per[q]) else (0xff, 0, 0, 0))
        while tiles and tiles[-1][0] == 0xff:      # trailing padding of the stage is not launched
            tiles.pop()
        first.append(len(tiles))
    return np.asarray(tiles, dtype=np.int32).reshape(-1, 4), np.asarray(first, dtype=np.int32)


def tiles_per_sequence(L, conditional):
    return sum(TILES[t] for t, _ in stage_list(L) if conditional or t != Q3_CA)


def supported(w, T):
    """Shapes the persistent kernel is specialised for (everything the reference config uses)."""
    return (w.precision == "bf16" and w.D == 512 and w.H == 16 and w.FF == 1024 and T <= 48
            and all("unc_tab" in lw for lw in w.layers))


class PersistentForward:
    """Buffers + tables of one DenoiserSession for rg_denoiser_forward."""

    def __init__(self, sess, mode="persistent"):
        """mode: "persistent" = one launch per forward (queues, hand-offs), "stages" = one launch per stage."""
        assert mode in ("persistent", "stages")
        w = sess.w
        self.sess, self.h, self.mode = sess, sess.h, mode
        B, T, D, M, dev = sess.B, w.T, w.D, sess.M, w.dev
        bf = lambda *s: torch.empty(*s, device=dev, dtype=torch.bfloat16)
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        self.xb_bf, self.xc_bf, self.ysa, self.yf = bf(M, D), bf(M, D), bf(M, D), bf(M, D)
        self.y3 = bf(B * T, 3 * D)
        self.g = bf(M, 2 * D)
        self.st_a, self.st_b, self.st_sa, self.st_f = f(M, 8, 2), f(M, 8, 2), f(M, 8, 2), f(M, 8, 2)
        self.st3 = f(B * T, 24, 2)
        self.sched_host = build_schedule(B, w.L)
        self.n_tiles = (len(self.sched_host) - SCHED_HEADER) // 4
        self.sched = torch.from_numpy(self.sched_host).to(dev)
        self.ctrl = torch.zeros(self.h.lib.rg_fwd_ctrl_words(B), device=dev, dtype=torch.int32)
        st_tiles, self.stage_first = build_stage_schedule(B, w.L)
        self.stage_tiles = torch.from_numpy(st_tiles).to(dev)
        self._stage_first_c = (ctypes.c_int * len(self.stage_first))(*[int(v) for v in self.stage_first])
        p = lambda t: t.data_ptr()
        layers = (FwdLayer * w.L)()
        for l, lw in enumerate(w.layers):
            e = layers[l]
            e.w_qkv, e.b_qkv, e.sa_g, e.sa_b = p(lw["w_qkv"].hi), p(lw["b_qkv"]), p(lw["sa_g"]), p(lw["sa_b"])
            e.sa_sg, e.sa_sb, e.w_sao, e.b_sao = p(lw["sa_sg"]), p(lw["sa_sb"]), p(lw["w_sao"].hi), p(lw["b_sao"])
            e.w_q3, e.b_q3, e.ca_g, e.ca_b = p(lw["w_q3"].hi), p(lw["b_q3"]), p(lw["ca_g"]), p(lw["ca_b"])
            e.a_pre = p(sess.a_pre[l])
            e.ca_sg, e.ca_sb, e.unc_tab = p(lw["ca_sgs"]), p(lw["ca_sbs"]), p(lw["unc_tab"])
            e.w_mix, e.b_mix = p(lw["w_mix"].hi), p(lw["b_mix"])
            e.w_ff1, e.b_ff1, e.w_ff2, e.b_ff2 = p(lw["w_ff1"].hi), p(lw["b_ff1"]), p(lw["w_ff2"].hi), p(lw["b_ff2"])
            e.ff_sg, e.ff_sb, e.w_ffo, e.b_ffo = p(lw["ff_sg"]), p(lw["ff_sb"]), p(lw["w_ffo"].hi), p(lw["b_ffo"])
        self.layers = torch.frombuffer(bytearray(bytes(layers)), dtype=torch.uint8).to(dev)
        a = self.args = FwdArgs()
        a.layers, a.L, a.B, a.T = p(self.layers), w.L, B, T
        a.w_embed, a.b_embed, a.tbias = p(w.w_embed.hi), p(w.b_embed), p(w.tbias)
        a.w_out, a.b_out, a.ss = p(w.w_out.hi), p(w.b_out), p(w.ss)
        a.src_mask, a.qmask = p(sess.src_mask), p(sess.qmask)
        a.xa, a.xb, a.xc, a.head = p(sess.xa), p(sess.xb), p(sess.xc), p(sess.head)
        a.xb_bf, a.xc_bf, a.ysa, a.yf, a.y3, a.g = p(self.xb_bf), p(self.xc_bf), p(self.ysa), p(self.yf), p(self.y3), p(self.g)
        a.st_a, a.st_b, a.st_sa, a.st_f, a.st3 = p(self.st_a), p(self.st_b), p(self.st_sa), p(self.st_f), p(self.st3)
        a.sched, a.ctrl, a.stamps = p(self.sched), p(self.ctrl), None

    def run(self, x, step, stamps=None):
        from . import capi
        a = self.args
        assert x.is_contiguous() and x.dtype == torch.float32 and x.numel() == self.sess.B * self.sess.w.T * self.sess.w.D
        a.x, a.step = x.data_ptr(), int(step)
        a.stamps = stamps.data_ptr() if stamps is not None else None
        s = torch.cuda.current_stream().cuda_stream
        if self.mode == "stages":
            rc = self.h.lib.rg_denoiser_forward_stages(self.h._h, ctypes.byref(a), ctypes.c_void_p(self.stage_tiles.data_ptr()),
                                                       self._stage_first_c, len(self.stage_first) - 1, ctypes.c_void_p(s))
        else:
            rc = self.h.lib.rg_denoiser_forward(self.h._h, ctypes.byref(a), ctypes.c_void_p(s))
        if rc != 0:
            raise capi.RgError("rg_denoiser_forward failed (%d): %s" % (rc, self.h.lib.rg_last_error(self.h._h).decode()))
        return self.sess.head

    def aborted(self):
        """True if a dependency wait of the last run gave up (reads back one word: diagnostics / tests only)."""
        return int(self.ctrl[N_SHARD * 32].item()) != 0
