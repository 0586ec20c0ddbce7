"""Pins oracle/ssw_oracle.c (the CPU restatement) to the UNMODIFIED reference (oracle/_ref/libssw_ref.so):
whole-driver differential tests, per-kernel tests of the static SSE2 kernels and of banded_sw, and the
reference's own demo known answers.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from sswutil import (RES_FIELDS, OrcEnd, _ptr, blosum50, dna_matrix, encode_dna, i8p, i32p, mutate, oracle_align, random_ref,
                     ref_align, u32p)


def _rand_case(rng, kind, supported_only):
    if kind == "dna":
        n, nc = 5, 4
        mat = dna_matrix(int(rng.integers(1, 4)), int(rng.integers(1, 6)))
        refLen = int(rng.integers(20, 600))
        ref = random_ref(refLen, int(rng.integers(1 << 30)), 4, 0.02)
        rl = int(rng.integers(5, 200))
        sub = 0.05
    else:
        n, nc = 24, 20
        mat = blosum50()
        refLen = int(rng.integers(20, 400))
        ref = rng.integers(0, 20, size=refLen, dtype=np.int8)
        rl = int(rng.integers(5, 300))
        sub = 0.2
    if rng.random() < 0.7 and refLen > rl + 20:
        off = int(rng.integers(0, refLen - rl - 16))
        read = mutate(ref[off:off + rl + 8], rng, sub, 0.02, 0.02, nc)[:rl]
        if len(read) < 5:
            read = rng.integers(0, nc, size=rl, dtype=np.int8)
    else:
        read = rng.integers(0, nc, size=rl, dtype=np.int8)
    if supported_only:
        gapE = int(rng.integers(1, 4))
        gapO = gapE + int(rng.integers(1, 6))
    else:
        gapO, gapE = int(rng.integers(0, 8)), int(rng.integers(0, 5))
    flag = int(rng.choice([0, 1, 2, 8, 9, 15, 4, 6]))
    filters = int(rng.choice([0, 0, 20, 60]))
    filterd = int(rng.choice([0, 30, 1000]))
    maskLen = int(rng.choice([len(read) // 2, 15, 10, 40]))
    ss = int(rng.choice([2, 2, 2, 0, 1]))
    return read, mat, n, ref, gapO, gapE, flag, filters, filterd, maskLen, ss


@pytest.mark.parametrize("kind,model,ncases", [("dna", 0, 1500), ("dna", 1, 1500), ("aa", 0, 600), ("aa", 1, 600)])
def test_driver_differential(reflib, kind, model, ncases):
    """orc_align (lane model: every parameter; plain model: gapO > gapE) == reference ssw_init + ssw_align."""
    rng = np.random.default_rng(100 + model + (7 if kind == "aa" else 0))
    n_null = n_word = n_flag1 = 0
    for _ in range(ncases):
        args = _rand_case(rng, kind, supported_only=(model == 1))
        a = ref_align(*args)
        b = oracle_align(*args, model=model)
        bb = (None if b[0] is None else {k: b[0][k] for k in RES_FIELDS}, b[1])
        assert a == bb, (args[4:], len(args[0]), len(args[3]))
        n_null += a[0] is None
        if b[0] is not None:
            n_word += b[0]["used_word"]
            n_flag1 += b[0]["flag"] == 1
    assert n_word > 0   # both the 8-bit and the 16-bit rule were exercised
    assert n_word < ncases


def test_static_kernels(reflib, oracle):
    """sw_sse2_byte / sw_sse2_word (forward and reverse) vs the lane model, kernel by kernel."""
    rng = np.random.default_rng(5)
    for _ in range(1200):
        read, mat, n, ref, gapO, gapE, _, _, _, maskLen, _ = _rand_case(rng, "dna" if rng.random() < 0.7 else "aa", False)
        bias = int(-min(0, int(mat.min())))
        rev = int(rng.integers(0, 2))
        term = int(rng.choice([255, 20, 40]))
        out = np.zeros(5, dtype=np.int32)
        ends = (OrcEnd * 2)()
        reflib.refwrap_sw_byte(_ptr(ref, i8p), rev, len(ref), _ptr(read, i8p), len(read), _ptr(mat, i8p), n, gapO, gapE, term, bias,
                               maskLen, _ptr(out, i32p))
        oracle.orc_striped(1, _ptr(ref, i8p), rev, len(ref), _ptr(read, i8p), len(read), _ptr(mat, i8p), n, gapO, gapE, term, bias,
                           maskLen, ends, None)
        assert list(out) == [ends[0].score, ends[0].ref, ends[0].read, ends[1].score, ends[1].ref]
        term = int(rng.choice([65535, 20, 40]))
        reflib.refwrap_sw_word(_ptr(ref, i8p), rev, len(ref), _ptr(read, i8p), len(read), _ptr(mat, i8p), n, gapO, gapE, term,
                               maskLen, _ptr(out, i32p))
        oracle.orc_striped(0, _ptr(ref, i8p), rev, len(ref), _ptr(read, i8p), len(read), _ptr(mat, i8p), n, gapO, gapE, term, 0,
                           maskLen, ends, None)
        assert list(out) == [ends[0].score, ends[0].ref, ends[0].read, ends[1].score, ends[1].ref]


def test_banded_sw(reflib, oracle):
    """banded_sw on arbitrary sub-rectangles and band widths (including ones that make it fail)."""
    rng = np.random.default_rng(6)
    mat = dna_matrix(2, 2)
    for _ in range(1500):
        refLen = int(rng.integers(3, 80))
        ref = rng.integers(0, 4, size=refLen, dtype=np.int8)
        read = mutate(ref, rng, 0.1, 0.05, 0.05)
        if len(read) < 2:
            continue
        band = int(rng.integers(1, 12))
        score = int(rng.integers(1, 2 * len(read)))
        gapE = int(rng.integers(1, 3)); gapO = gapE + int(rng.integers(0, 4))
        cap = refLen + len(read) + 8
        a = np.zeros(cap, dtype=np.uint32); b = np.zeros(cap, dtype=np.uint32)
        la = reflib.refwrap_banded_sw(_ptr(ref, i8p), _ptr(read, i8p), refLen, len(read), score, gapO, gapE, band, _ptr(mat, i8p), 5,
                                      _ptr(a, u32p), cap)
        lb = oracle.orc_banded(_ptr(ref, i8p), _ptr(read, i8p), refLen, len(read), score, gapO, gapE, band, _ptr(mat, i8p), 5,
                               _ptr(b, u32p), cap)
        assert la == lb
        if la > 0:
            assert list(a[:la]) == list(b[:lb])


def test_mark_mismatch(reflib, oracle):
    rng = np.random.default_rng(8)
    mat = dna_matrix(2, 2)
    done = 0
    for _ in range(400):
        ref = rng.integers(0, 4, size=int(rng.integers(40, 200)), dtype=np.int8)
        off = int(rng.integers(0, len(ref) - 30))
        read = mutate(ref[off:off + 60], rng, 0.08, 0.03, 0.03)
        read = np.concatenate([rng.integers(0, 4, size=5, dtype=np.int8), read, rng.integers(0, 4, size=4, dtype=np.int8)])
        d, cig = ref_align(read, mat, 5, ref, 3, 1, 1, 0, 0, 15)
        if d is None or not cig:
            continue
        done += 1
        cig = np.array(cig, dtype=np.uint32)
        out = np.zeros(len(cig) + 2 * len(read) + 4, dtype=np.uint32)
        olen = C.c_int32(0)
        nm = oracle.orc_mark_mismatch(d["ref_begin1"], d["read_begin1"], d["read_end1"], _ptr(ref, i8p), _ptr(read, i8p), len(read),
                                      _ptr(cig, u32p), len(cig), _ptr(out, u32p), C.byref(olen))
        # reference: operates on a malloc'ed cigar it frees; hand it a libc buffer
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        buf = libc.malloc(4 * len(cig))
        C.memmove(buf, cig.ctypes.data, 4 * len(cig))
        pc = C.cast(buf, u32p)
        cl = C.c_int32(len(cig))
        nm_ref = reflib.mark_mismatch(d["ref_begin1"], d["read_begin1"], d["read_end1"], _ptr(ref, i8p), _ptr(read, i8p), len(read),
                                      C.byref(pc), C.byref(cl))
        assert nm == nm_ref and olen.value == cl.value
        assert [int(pc[i]) for i in range(cl.value)] == [int(x) for x in out[:olen.value]]
        libc.free.argtypes = [C.c_void_p]
        libc.free(C.cast(pc, C.c_void_p))
    assert done > 200


def test_blosum50_matches_reference_table():
    """tests/sswutil.blosum50() is the table the reference CLI uses (reference src/main.c:43-69)."""
    path = "/root/reference/src/main.c"
    if not os.path.exists(path):
        pytest.skip("reference sources not present")
    import re
    src = open(path).read()
    body = src[src.index("mat50[] = {"):src.index("};", src.index("mat50[] = {"))]
    body = re.sub(r"//[^\n]*", "", body.split("{", 1)[1])
    vals = [int(x) for x in re.findall(r"-?\d+", body)]
    assert vals == [int(x) for x in blosum50()]


def test_example_c_known_answer():
    """reference src/example.c:105-156 -- score 21/8, ref 8..21, read 0..14, 9M1I5M."""
    d, cig = oracle_align(encode_dna("CTGAGCCGGTAAATC"), dna_matrix(2, 2), 5,
                          encode_dna("CAGCCTTTCTGACCCGGAAATCAAAATAGGCACAACAAA"), 3, 1, 1, 0, 0, 15)
    assert (d["score1"], d["score2"], d["ref_begin1"], d["ref_end1"], d["read_begin1"], d["read_end1"], d["ref_end2"]) == \
        (21, 8, 8, 21, 0, 14, 4)
    assert cig == [(9 << 4) | 0, (1 << 4) | 1, (5 << 4) | 0]
