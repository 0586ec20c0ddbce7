"""The column-frame form of the recurrence (csrc/lanes.h pk_max3_fr / fr_pack, DESIGN.md "column frame"): every stored value
carries + phi(column), phi growing by gapE per column, so that E needs no decrement, the add / subtract of a row are plain
32-bit adds on the packed pair and the maxima are three-input binary16 maxima on the bit patterns.

 * the arithmetic facts, with numpy (no GPU, no emulator): non-negative int16 below 0x7C00 order like binary16 numbers, a
   "dead" operand (0x8000 + value) is a negative finite binary16 number; the packed profile entry gives exact per-half sums;
   the frame cell computes the H matrix of the plain recurrence whatever the renormalisation period;
 * the real kernel source on the SIMT emulator with a SMALL renormalisation period (SSW_GPU_FRAME_K=16/64: renormalised every
   block), random scoring systems, pairs of unequal lengths (dead rows in one half), tiles with halos -- against the reference.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from parity import compare_batch, make_reads
from sswutil import dna_matrix, random_ref

HERE = os.path.dirname(os.path.abspath(__file__))


def test_patterns_order_like_binary16_and_dead_operands_lose():
    k = np.arange(0, 0x7C00, dtype=np.uint16)
    f = k.view(np.float16)
    assert np.isfinite(f).all() and (np.diff(f.astype(np.float64)) > 0).all()
    dead = (np.uint16(0x8000) + np.arange(1, 0x7C00, dtype=np.uint16)).view(np.float16)      # 0x8000 + value, value below 0x7C00
    assert np.isfinite(dead).all() and (dead.astype(np.float64) < 0).all()


def test_packed_profile_entries_add_exactly_per_half():
    rng = np.random.default_rng(5)
    for _ in range(20000):
        lo = int(rng.integers(-128, 200)) if rng.random() < 0.8 else 32768      # live score (+ gapE) or FR_DEAD
        hi = int(rng.integers(-128, 200)) if rng.random() < 0.8 else 32768
        entry = (lo + (hi << 16)) & 0xffffffff                                   # fr_pack
        da, db = int(rng.integers(128, 0x7C00 - 256)), int(rng.integers(128, 0x7C00 - 256))
        x = (da + (db << 16) + entry) & 0xffffffff                               # ONE 32-bit add
        assert x & 0xffff == (da + lo) & 0xffff and x >> 16 == (db + hi) & 0xffff
        if lo == 32768: assert x & 0x8000
        if hi == 32768: assert (x >> 16) & 0x8000


def _plain(s, gO, gE):
    n, m = s.shape
    H = np.zeros((n + 1, m + 1), dtype=np.int64); E = np.zeros((n + 1, m + 2), dtype=np.int64)
    for j in range(1, m + 1):
        f = 0
        for i in range(1, n + 1):
            h = max(H[i - 1, j - 1] + s[i - 1, j - 1], E[i, j], f, 0)
            H[i, j] = h
            E[i, j + 1] = max(E[i, j] - gE, h - gO, 0)
            f = max(f - gE, h - gO, 0)
    return H


def _frame(s, gO, gE, base, K):
    """the frame cell as csrc chain_rows_fr computes it, one lane (all rows), renormalised every K columns"""
    n, m = s.shape
    phi = lambda j: base + ((j - 1) % K + 1) * gE                                 # frame of column j (1-based), period K
    c1 = gO - gE
    Hp = np.full(n + 1, phi(1) - gE, dtype=np.int64)                              # column 0 in the frame of column 0
    Ef = np.full(n + 1, phi(1), dtype=np.int64)
    out = np.zeros((n + 1, m + 1), dtype=np.int64)
    for j in range(1, m + 1):
        if j > 1 and (j - 1) % K == 0:
            Hp -= K * gE; Ef -= K * gE
        fl = phi(j) + gE
        Hn = np.empty_like(Hp); Hn[0] = phi(j)
        f = 0                                                                      # lane 0: the zero the DPP move fills in
        for i in range(1, n + 1):
            x = Hp[i - 1] + s[i - 1, j - 1] + gE
            assert x >= 0 and Ef[i] >= 0 and f >= 0
            h = max(x, Ef[i], f)
            t = h - c1
            assert t >= 0
            Ef[i] = max(Ef[i], t, fl)
            f = max(f, t) - gE
            Hn[i] = h
            out[i, j] = h - phi(j)
        Hp = Hn
    return out


def test_frame_cell_equals_plain_cell_whatever_the_period():
    rng = np.random.default_rng(3)
    for it in range(30):
        n, m = int(rng.integers(5, 40)), int(rng.integers(20, 120))
        s = rng.integers(-6, 7, size=(n, m))
        gE = int(rng.integers(1, 5)); gO = gE + int(rng.integers(1, 8))
        base = 6 + gO + 2 * gE + 8
        for K in (16, 64, 1024):
            assert (_frame(s, gO, gE, base, K) == _plain(s, gO, gE)).all(), (it, K)


@pytest.mark.parametrize("K", ["16", "64", ""])
def test_frame_form_in_the_fill_kernel_on_the_emulator(emu_lib_path, K):
    """k_fill<R, frame>: random scoring systems, read pairs of unequal length (dead rows in one half of the registers), several
    tiles with halos, renormalisation every 16 / 64 steps (and the default period): every field against the reference"""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import ssw_amd
from parity import compare_batch, make_reads
from sswutil import dna_matrix, random_ref
lib = ssw_amd.load(%r)
ctx = ssw_amd.Context(0, lib)
rng = np.random.default_rng(11)
total = 0
for case in range(6):
    match, mism = int(rng.integers(1, 6)), int(rng.integers(1, 7))
    gE = int(rng.integers(1, 4)); gO = gE + int(rng.integers(1, 7))
    mat = dna_matrix(match, mism)
    ref = random_ref(int(rng.integers(900, 2600)), 100 + case, 4)
    lens = [int(x) for x in rng.integers(20, 170, size=5)]
    reads = make_reads(rng, ref, 5, lens, 4)
    flag = int(rng.choice([0, 1, 2]))
    Q = ctx.upload(reads); T = ctx.upload([ref])
    res, cig = ctx.align_batch(Q, T, mat, 5, gO, gE, flag, 0, 0, -1, 2)
    tm = ctx.timing()
    Q.free(); T.free()
    assert "frame" in tm["fill_kernel"], tm["fill_kernel"]
    bad = compare_batch(res, cig, reads, [ref], mat, 5, gO, gE, flag, 0, 0, -1, 2)
    assert not bad, "case %%d (%%d/-%%d/%%d/%%d flag %%d): " %% (case, match, mism, gO, gE, flag) + "\n".join(bad)
    total += len(reads)
ctx.close()
print("ok", total)
''' % (os.path.join(os.path.dirname(HERE), "complete-striped-smith-waterman-library_amd"), HERE, emu_lib_path)
    env = dict(os.environ, SSW_GPU_NO_DB="1")
    if K:
        env["SSW_GPU_FRAME_K"] = K
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=1800)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]


def test_frame_form_with_large_gap_penalties_and_extreme_matrices_on_the_emulator(emu_lib_path):
    """gap extension up to 255 (renormalisation period 64, frame offsets of the order of 10^4), matrices using the whole int8 range,
    a protein alphabet -- short queries (k_fill), long queries (strips + window passes) and the database path; the emulator aborts on
    any operand outside the frame form's range, the records are compared with the reference"""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import ssw_amd
from parity import compare_batch, make_reads
from sswutil import blosum50, random_ref
lib = ssw_amd.load(%r)
ctx = ssw_amd.Context(0, lib)
rng = np.random.default_rng(23)
frames = 0
for case, (gO, gE, hi, lo) in enumerate(((120, 60, 40, -90), (255, 254, 9, -9), (90, 20, 127, -128), (30, 7, 5, -128), (200, 100, 2, -3), (16, 15, 60, -60))):
    n = 5 if case %% 2 == 0 else 24
    mat = rng.integers(lo, hi + 1, size=(n, n)).astype(np.int8)
    for k in range(n): mat[k, k] = hi
    mat = np.ascontiguousarray(mat.reshape(-1))
    nc = n - 1 if n == 5 else 20
    ref = rng.integers(0, nc, size=1300, dtype=np.int8)
    short = make_reads(rng, ref, 5, [150, 33, 90, 200, 61], nc, sub=0.05, ins=0.01, dele=0.01)
    longq = make_reads(rng, ref, 2, [500, 401], nc, sub=0.03, ins=0.01, dele=0.01, frac_random=0.0)
    for reads, flag in ((short, 2), (longq, 2), (short, 0)):
        refs = [ref] if flag else [ref[:400].copy(), ref[300:900].copy(), ref[100:333].copy(), ref[700:1300].copy(), ref[:77].copy()]
        Q = ctx.upload(reads); T = ctx.upload(refs)
        res, cig = ctx.align_batch(Q, T, mat, n, gO, gE, flag, 0, 0, -1, 2)
        frames += "frame" in ctx.timing()["fill_kernel"]
        Q.free(); T.free()
        bad = compare_batch(res, cig, reads, refs, mat, n, gO, gE, flag, 0, 0, -1, 2)
        assert not bad, "case %%d gaps %%d/%%d flag %%d: " %% (case, gO, gE, flag) + "\n".join(bad)
ctx.close()
print("ok", frames)
''' % (os.path.join(os.path.dirname(HERE), "complete-striped-smith-waterman-library_amd"), HERE, emu_lib_path)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]
    assert int(r.stdout.split()[1]) >= 10          # most of these launches did run in the frame form (the rest: buckets beyond its range)


@pytest.mark.gpu
@pytest.mark.parametrize("K", ["16", "64"])
def test_frame_renormalised_often_on_the_gpu(gpu_hctx, K, monkeypatch):
    """the default period is 1024 steps: tests on kilobase targets would never renormalise on the hardware.  Config 2's first 4000 reads
    against the 1 Mb target (k_fill), 600 long reads (strips + window passes) and a protein search with the frame renormalised every
    16 / 64 steps, against the full-size fixtures / the reference"""
    import workloads as W
    gpu_ctx = gpu_hctx      # SSW_GPU_FRAME_K is a test hook: libssw_hooks.so (conftest.py)
    monkeypatch.setenv("SSW_GPU_FRAME_K", K)
    z = np.load(os.path.join(HERE, "golden", "full", "config2_block0.npz"))
    ref, reads, p = W.dna_config(2, 0)
    k = 4000
    Q = gpu_ctx.upload(list(reads[:k])); T = gpu_ctx.upload([ref])
    try:
        res, cig = gpu_ctx.align_batch(Q, T, dna_matrix(2, 2), 5, 3, 1, 2, 0, 0, -1, 2)
    finally:
        Q.free(); T.free()
    g = res[:, 0]
    got = np.stack([g["score1"], g["score2"], g["ref_begin1"], g["ref_end1"], g["read_begin1"], g["read_end1"], g["ref_end2"], g["cigarLen"], g["flag"]], axis=1).astype(np.int32)
    assert (got == z["fields"][:k]).all()
    z4 = np.load(os.path.join(HERE, "golden", "full", "config4_block0.npz"))
    ref4, reads4, p4 = W.dna_config(4, 0)
    k4 = 600
    Q = gpu_ctx.upload(list(reads4[:k4])); T = gpu_ctx.upload([ref4])
    try:
        res, cig = gpu_ctx.align_batch(Q, T, dna_matrix(2, 2), 5, 3, 1, 2, 0, 0, p4["mask_len"], 2)
    finally:
        Q.free(); T.free()
    g = res[:, 0]
    got = np.stack([g["score1"], g["score2"], g["ref_begin1"], g["ref_end1"], g["read_begin1"], g["read_end1"], g["ref_end2"], g["cigarLen"], g["flag"]], axis=1).astype(np.int32)
    assert (got == z4["fields"][:k4]).all()
    z5 = np.load(os.path.join(HERE, "golden", "full", "config5_block0.npz"))
    db, qs, mat = W.protein_config(0)
    Q = gpu_ctx.upload(qs[:16]); T = gpu_ctx.upload(db)
    try:
        hits = gpu_ctx.search_db(Q, T, mat, 24, 3, 1, -1, 2, 2500)
    finally:
        Q.free(); T.free()
    rows = np.stack([hits["score1"], hits["score2"], hits["ref_end1"], hits["read_end1"], hits["ref_end2"]], axis=2).astype(np.int32)
    assert (rows == z5["first16"]).all()
