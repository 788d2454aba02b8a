"""CPU: the stream packing of the sequence-stationary forward (rag-gesture_amd/seqfwd.py) against the indexing the kernel
uses (csrc/rg_seq.hip), written out element by element, and the ctypes mirror of rg_seq_args against the C header."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch


def _bf(t):
    return t.to(torch.bfloat16).float()


def test_pack_unit_is_the_mfma_operand_image(rg):
    SQ = rg.seqfwd
    W = torch.randn(512, 512)
    pk = SQ.pack_unit(W).float()            # [8 waves][64 fragments][64 lanes][8]
    Wb = _bf(W)
    rng = np.random.default_rng(0)
    for _ in range(2000):
        w, s, j, lane, e = (int(rng.integers(n)) for n in (8, 16, 4, 64, 8))
        nl, g = lane & 15, lane >> 4
        # kernel: fragment 4 s + j of wave w, lane (nl, g), element e = A[i = nl][k = 8 g + e] of block (j, s)
        assert pk[w, 4 * s + j, lane, e] == Wb[64 * w + 16 * j + nl, 32 * s + 8 * g + e]


def test_pack_kv_orders_heads_then_key_value(rg):
    SQ = rg.seqfwd
    Wk, Wv = torch.randn(512, 512), torch.randn(512, 512)
    pk = SQ.pack_kv(Wk, Wv).float()         # [8][128][64][8]
    src = (_bf(Wk), _bf(Wv))
    rng = np.random.default_rng(1)
    for _ in range(2000):
        w, h, kv, s, j2, lane, e = (int(rng.integers(n)) for n in (8, 2, 2, 16, 2, 64, 8))
        nl, g = lane & 15, lane >> 4
        frag = ((h * 2 + kv) * 16 + s) * 2 + j2          # consumption order: head, key | value, step, block of the head
        assert pk[w, frag, lane, e] == src[kv][64 * w + 32 * h + 16 * j2 + nl, 32 * s + 8 * g + e]


def test_a_fragments_enumerate_the_contraction_like_the_query_accumulators(rg):
    SQ = rg.seqfwd
    A = torch.randn(2, 3, 2, 16, 32, 32)    # [L][3][B][H][i][j]
    fr = SQ.a_fragments(A)                  # [L][3][B][8 waves][2 heads][2 blocks][64][8]
    assert fr.shape == (2, 3, 2, 8, 2, 2, 64, 8) and fr.dtype == torch.bfloat16
    rng = np.random.default_rng(2)
    for _ in range(2000):
        l, c, b, w, hh, jb, lane, e = (int(rng.integers(n)) for n in (2, 3, 2, 8, 2, 2, 64, 8))
        jj, g = lane & 15, lane >> 4
        i = 4 * g + e if e < 4 else 16 + 4 * g + e - 4
        a = A[l, c, b, 2 * w + hh, i, 16 * jb + jj]
        assert fr[l, c, b, w, hh, jb, lane, e].float() == _bf(a)


def test_seq_args_layout_matches_the_header(rg, tmp_path):
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    S = rg.seqfwd.SeqArgs
    fields = [n for n, _ in S._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rg_gesture.h"', 'int main(void) {',
             '  printf("size %zu\\n", sizeof(rg_seq_args));']
    lines += ['  printf("%s %%zu\\n", offsetof(rg_seq_args, %s));' % (f, f) for f in fields]
    lines += ['  return 0;', '}']
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    r = subprocess.run([cc, "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    assert int(out["size"]) == ctypes.sizeof(S)
    for f in fields:
        assert int(out[f]) == getattr(S, f).offset, f
