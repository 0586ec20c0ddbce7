"""Parity in the regimes where 16-bit arithmetic clips or where the host switches between the forms of the fill recurrence:

* scores at and beyond 32767 -- the reference's 16-bit kernel saturates (`_mm_adds_epi16`, reference src/ssw.c:483) and
  has no overflow detection, so score1 = 32767 and everything derived from it (end positions, the reverse pass that
  terminates on that value, the traceback that cannot re-score its CIGAR) is part of the contract;
* the range limit of the column-frame form (csrc/ssw_host.c ssw_frame_params: bucket bound + frame offsets < 31744), on both
  sides of which k_fill / k_chainq / k_filldb run as <.., frame> or as plain int16 -- with reads that REACH the bound (exact
  copies of padded length x the match score), so the largest frame values really occur;
* the boundaries rounds 1-2 had (31744 of the two-row maximum, 2048 of the f16 form, max(mat) <= 49 of the database path): kept
  as ordinary cases -- nothing special happens there any more.

The same seeded cases run on the CPU emulator (small, `not gpu`) and on the MI355X (`gpu`), against the compiled reference.
"""
import numpy as np
import pytest

import ssw_amd
from parity import compare_batch, make_reads
from sswutil import blosum50, random_ref


def _mat(match, mismatch, n_score=0):
    m = np.full((5, 5), n_score, dtype=np.int64)
    for i in range(4):
        for j in range(4):
            m[i, j] = match if i == j else -mismatch
    return m.astype(np.int8).reshape(-1).copy()


def _clean(ref, rng, length, sub=0.0):
    off = int(rng.integers(0, len(ref) - length))
    r = ref[off:off + length].copy()
    if sub > 0:
        hit = rng.random(length) < sub
        r[hit] = (r[hit] + 1 + rng.integers(0, 3, size=int(hit.sum()))) % 4
    return np.ascontiguousarray(r, dtype=np.int8)


def cases(scale):
    """scale 0: emulator sizes; 1: GPU sizes.  -> list of (name, reads, refs, mat, n, gapO, gapE, flag, maskLen, score_size, env)"""
    out = []
    rng = np.random.default_rng(4242)
    big = scale > 0
    ref = random_ref(2600 if big else 900, 31, 4, 0.002)
    nrep = 24 if big else 3

    # --- saturation at 32767: short queries (k_fill, form 0), all flags that change the path
    for match, mism, gO, gE in ((127, 127, 20, 5), (127, 30, 40, 3), (100, 128, 3, 1)):
        lens = [300, 384, 259, 258, 257, 330, 370] + [int(x) for x in rng.integers(255, 385, size=nrep)]
        reads = [_clean(ref, rng, L, sub=0.01 if i % 3 == 2 else 0.0) for i, L in enumerate(lens)]
        reads += make_reads(rng, ref, 2, [340, 290], 4, sub=0.03, ins=0.01, dele=0.01, frac_random=0.5)
        for flag in ((0, 1, 2, 8, 15) if big else (0, 2)):
            out.append(("sat_fill_m%d_x%d_f%d" % (match, mism, flag), reads, [ref], _mat(match, mism), 5, gO, gE, flag, -1, 2, {}))
    out.append(("sat_fill_word_only", reads, [ref], _mat(127, 127), 5, 20, 5, 1, 15, 1, {}))

    # --- saturation in the strip kernel (long queries), 64-lane and 16-lane chains
    lens = [700, 500, 385, 640] + ([1200, 2000, 1025, 900] if big else [])
    reads = [_clean(ref, rng, L, sub=0.01 if i % 2 else 0.0) for i, L in enumerate(lens)]
    reads += make_reads(rng, ref, 2, [450, 600], 4, sub=0.03, ins=0.01, dele=0.01, frac_random=0.0)
    for flag in (0, 2):
        out.append(("sat_strip_f%d" % flag, reads, [ref], _mat(127, 127), 5, 20, 5, flag, -1, 2, {}))
    out.append(("sat_strip_16lanes", reads[:4], [ref], _mat(120, 90), 5, 9, 2, 1, -1, 2, {"SSW_GPU_XLANES": "16"}))
    out.append(("sat_strip_thin", reads[:4], [ref], _mat(127, 127), 5, 20, 5, 0, -1, 2, {"SSW_GPU_XR": "3"}))

    # --- 31744: the boundary between the two int16 forms of k_fill.  16 R max(mat) < 31744 selects the max3 form.
    for R, mm in ((24, 82), (24, 83), (16, 123), (16, 124), (20, 99), (20, 100)):
        L = 16 * R
        reads = [_clean(ref, rng, L), _clean(ref, rng, L - 1), _clean(ref, rng, L - 9, sub=0.01), _clean(ref, rng, L, sub=0.02)]
        out.append(("form_boundary_R%d_m%d" % (R, mm), reads, [ref], _mat(mm, 11), 5, 7, 2, 1, -1, 2, {}))
        out.append(("form_boundary_R%d_m%d_int16" % (R, mm), reads, [ref], _mat(mm, 11), 5, 7, 2, 0, -1, 2, {"SSW_GPU_FILL_F16": "0"}))

    # --- the range limit of the column-frame form: gaps 7/2 -> K = 1024, offsets up to ~2114 (16-lane chains) / ~2210 (64-lane chains)
    for R, mm in ((24, 77), (24, 78), (16, 115), (16, 116), (20, 92), (20, 93)):       # 16 R mm = 29568 / 29952, 29440 / 29696, 29440 / 29760
        L = 16 * R
        reads = [_clean(ref, rng, L), _clean(ref, rng, L - 1), _clean(ref, rng, L - 9, sub=0.01), _clean(ref, rng, L, sub=0.02)]
        out.append(("frame_limit_R%d_m%d" % (R, mm), reads, [ref], _mat(mm, 11), 5, 7, 2, 1, -1, 2, {}))
    for mm in (41, 42):       # strip kernel: 704 rows x 41 = 28864 (frame), x 42 = 29568 (plain int16)
        reads = [_clean(ref, rng, 704), _clean(ref, rng, 700, sub=0.01), _clean(ref, rng, 690)]
        out.append(("frame_limit_strip_m%d" % mm, reads, [ref], _mat(mm, 11), 5, 7, 2, 2, -1, 2, {}))
    for K in ("16", "64"):    # ... and with the frame renormalised every 16 / 64 steps, scores still near the top of the range
        reads = [_clean(ref, rng, 384), _clean(ref, rng, 383), _clean(ref, rng, 300, sub=0.02)]
        out.append(("frame_limit_K%s" % K, reads, [ref], _mat(80, 11), 5, 7, 2, 1, -1, 2, {"SSW_GPU_FRAME_K": K}))

    # --- 2048: the guard of the f16 form, 16 R max(mat) <= 2047, reads that reach the bound
    for R, mm in ((8, 15), (8, 16), (10, 12), (10, 13), (4, 31), (4, 32), (2, 63), (2, 64), (1, 127), (16, 7), (16, 8), (21, 6), (21, 7)):
        L = 16 * R
        reads = [_clean(ref, rng, L), _clean(ref, rng, max(1, L - 7)), _clean(ref, rng, L, sub=0.03)]
        reads += make_reads(rng, ref, 3, [L, max(1, L - 3), max(1, L - 15)], 4, sub=0.04, ins=0.01, dele=0.01, frac_random=0.2)
        out.append(("f16_guard_R%d_m%d" % (R, mm), reads, [ref], _mat(mm, min(2 * mm, 127)), 5, 5, 2, 2 if R > 2 else 0, -1, 2, {}))

    # --- database search (k_filldb): max(mat) <= 49 keeps 640 rows below 31744; 50 must take the other path
    dbn = 24 if big else 6
    db = [np.ascontiguousarray(ref[int(o):int(o) + int(L)]) for o, L in zip(rng.integers(0, len(ref) - 700, size=dbn), rng.integers(200, 700, size=dbn))]
    qs = [db[0][:640].copy(), db[1][:384].copy(), db[2][:400].copy(), db[3][:150].copy(), _clean(ref, rng, 640, sub=0.02), _clean(ref, rng, 333)]
    for mm in (46, 47, 49, 50):       # 640 x 46 = 29440: frame; 47: plain int16 in k_filldb; 50: outside the fused kernel
        out.append(("db_guard_m%d" % mm, qs, db, _mat(mm, 20), 5, 11, 3, 0, -1, 2, {}))
    # protein matrix scaled so that long homologous queries go far beyond 255 and towards the 16-bit limit
    b50 = blosum50().astype(np.int64)
    aa = [rng.integers(0, 20, size=int(L), dtype=np.int8) for L in rng.integers(150, 640, size=dbn)]
    qa = [aa[0].copy(), aa[1][:300].copy(), aa[2].copy(), rng.integers(0, 20, size=200, dtype=np.int8)]
    out.append(("db_blosum_x3", qa, aa, np.clip(b50 * 3, -128, 127).astype(np.int8), 24, 30, 6, 0, -1, 2, {}))
    out.append(("db_blosum_x8_saturating", qa, aa[:4], np.clip(b50 * 8, -128, 127).astype(np.int8), 24, 60, 9, 0, -1, 2, {}))
    return out


def _check(ctx, case, monkeypatch):
    name, reads, refs, mat, n, gapO, gapE, flag, maskLen, ss, env = case
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    Q = ctx.upload(reads); T = ctx.upload(refs)
    try:
        res, cig = ctx.align_batch(Q, T, mat, n, gapO, gapE, flag, 0, 0, maskLen, ss)
    finally:
        Q.free(); T.free()
    bad = compare_batch(res, cig, reads, refs, mat, n, gapO, gapE, flag, 0, 0, maskLen, ss)
    assert not bad, name + "\n" + "\n".join(bad)
    return res


_EMU = cases(0)
_GPU = cases(1)


@pytest.fixture(scope="module")
def ectx(emu_lib_path):
    ctx = ssw_amd.Context(0, ssw_amd.load(emu_lib_path))
    yield ctx
    ctx.close()


@pytest.mark.parametrize("case", _EMU, ids=[c[0] for c in _EMU])
def test_clipping_regimes_emulated(ectx, case, monkeypatch):
    res = _check(ectx, case, monkeypatch)
    if case[0].startswith("sat_"):
        assert int(res["score1"].max()) == 32767     # the case does reach the clipping value


@pytest.mark.gpu
@pytest.mark.parametrize("case", _GPU, ids=[c[0] for c in _GPU])
def test_clipping_regimes_gpu(gpu_ctx, gpu_hctx, case, monkeypatch):
    res = _check(gpu_hctx if case[-1] else gpu_ctx, case, monkeypatch)      # cases that force a kernel form: libssw_hooks.so (conftest.py)
    if case[0].startswith("sat_"):
        assert int(res["score1"].max()) == 32767
