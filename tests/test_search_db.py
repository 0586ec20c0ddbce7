"""ssw_gpu_search_db (include/ssw_gpu.h): the streamed database search must hand out, chunk by chunk, exactly the
score1 / score2 / ref_end1 / read_end1 / ref_end2 of a plain ssw_gpu_align_batch over the same queries and targets --
which the parity tests pin to the reference -- whatever the chunk size, also when the fused kernel does not cover the
batch (a query above 640 residues -> generic path), and must stop when the caller's function says so."""
import os

import numpy as np
import pytest

import ssw_amd
from parity import compare_batch
from sswutil import blosum50, dna_matrix, random_ref
import workloads as W


@pytest.fixture(scope="module")
def ectx(emu_lib_path):
    ctx = ssw_amd.Context(0, ssw_amd.load(emu_lib_path))
    yield ctx
    ctx.close()


def _same(hits, res):
    for f in ("score1", "score2", "ref_end1", "read_end1", "ref_end2"):
        assert (hits[f] == res[f]).all(), f


def _case(ctx, qs, db, mat, n, gapO, gapE, chunks, check_ref=True):
    Q = ctx.upload(qs); T = ctx.upload(db)
    try:
        res, cig = ctx.align_batch(Q, T, mat, n, gapO, gapE, 0, 0, 0, -1, 2)
        if check_ref:
            bad = compare_batch(res, cig, qs, db, mat, n, gapO, gapE, 0, 0, 0, -1, 2)
            assert not bad, "\n".join(bad)
        for chunk in chunks:
            seen = []
            hits = ctx.search_db(Q, T, mat, n, gapO, gapE, -1, 2, chunk)
            _same(hits, res)
            got = np.zeros_like(hits)
            def on_chunk(tfirst, h):
                seen.append((tfirst, h.shape[1])); got[:, tfirst:tfirst + h.shape[1]] = h; return 0
            assert ctx.search_db(Q, T, mat, n, gapO, gapE, -1, 2, chunk, on_chunk) == 0
            _same(got, res)
            step = chunk if chunk > 0 else 2048
            assert seen == [(t0, min(step, len(db) - t0)) for t0 in range(0, len(db), step)]
        return res
    finally:
        Q.free(); T.free()


def test_streamed_search_equals_batch_emulated(ectx):
    db, qs, mat = W.protein_config(0, queries=200, db_entries=40)
    qs = qs[:9] + [qs[100], np.zeros(0, dtype=np.int8)]      # a planted homolog and an empty query
    db = db[:13] + [np.zeros(0, dtype=np.int8)]
    _case(ectx, qs, db, mat, 24, 3, 1, chunks=(5, 1, 14, 0))


def test_streamed_search_generic_path_and_stop_emulated(ectx):
    rng = np.random.default_rng(5)
    ref = random_ref(1500, 8, 4)
    db = [np.ascontiguousarray(ref[o:o + L]) for o, L in ((0, 300), (200, 90), (700, 411), (50, 33), (900, 260))]
    qs = [np.ascontiguousarray(ref[100:800]), np.ascontiguousarray(ref[10:160]), rng.integers(0, 4, size=77, dtype=np.int8)]   # 700 residues: not fused
    _case(ectx, qs, db, dna_matrix(2, 2), 5, 3, 1, chunks=(2,))
    Q = ectx.upload(qs[1:]); T = ectx.upload(db)
    calls = []
    rc = ectx.search_db(Q, T, dna_matrix(2, 2), 5, 3, 1, -1, 2, 2, lambda t0, h: (calls.append(t0), 7)[1] if t0 >= 2 else calls.append(t0))
    assert rc == 7 and calls == [0, 2]
    Q.free(); T.free()


@pytest.mark.gpu
def test_streamed_search_equals_batch_gpu(gpu_ctx):
    db, qs, mat = W.protein_config(0, queries=700, db_entries=300)
    res = _case(gpu_ctx, qs, db, mat, 24, 3, 1, chunks=(64, 300, 7), check_ref=False)
    sub = list(range(0, 700, 23))
    Q = gpu_ctx.upload([qs[i] for i in sub]); T = gpu_ctx.upload(db[:40])
    r2, c2 = gpu_ctx.align_batch(Q, T, mat, 24, 3, 1, 0, 0, 0, -1, 2)
    Q.free(); T.free()
    bad = compare_batch(r2, c2, [qs[i] for i in sub], db[:40], mat, 24, 3, 1, 0, 0, 0, -1, 2)
    assert not bad, "\n".join(bad)
    for f in ("score1", "score2", "ref_end1", "read_end1", "ref_end2"):
        assert (r2[f] == res[f][sub][:, :40]).all()


# ---- scores around 2048 (where the f16 form of rounds 1-2 saturated and workgroups had to repeat): the column-frame form has no limit
#      there -- same records as the plain int16 form and as the reference ----
def _around_2048_case():
    """n = 5 alphabet, A-A scores 7, C-C 1, G-G 5, every mismatch -4: the self-alignment of A^292 C^k scores 2044 + k -- 2047, 2048,
    2049-2051 -- next to unrelated short sequences and longer homologs far above; several sequences per size class so that the
    chains of one workgroup see very different scores."""
    mat = np.full((5, 5), -4, dtype=np.int8)
    mat[0, 0] = 7; mat[1, 1] = 1; mat[2, 2] = 5; mat[3, 3] = 2
    mat[4, :] = 0; mat[:, 4] = 0
    rng = np.random.default_rng(77)
    seqs = [np.array([0] * 292 + [1] * k, dtype=np.int8) for k in (0, 2, 3, 4, 5, 7)]
    seqs.append(np.array([0] * 300 + [2] * 100, dtype=np.int8))          # 2600: another size class
    seqs.append(np.array([0] * 120 + [1] * 30, dtype=np.int8))           # 870
    seqs += [rng.integers(0, 4, size=int(L), dtype=np.int8) for L in (295, 299, 150, 301, 64)]
    return seqs, np.ascontiguousarray(mat.reshape(-1))


def _around_2048_check(ctx, monkeypatch):
    seqs, mat = _around_2048_case()
    res = _case(ctx, seqs, seqs, mat, 5, 3, 1, chunks=(4, 0))
    s = np.array([[int(res["score1"][i, j]) for j in range(6)] for i in range(6)])
    assert s[2, 2] == 2047 and s[3, 3] == 2048 and s[4, 4] == 2049 and s[5, 5] == 2051 and s[0, 5] == 2044
    assert int(res["score1"][6, 6]) == 2600
    tm = ctx.timing()
    assert "frame" in tm["fill_kernel"] and tm["db_repeats"] == 0, tm
    monkeypatch.setenv("SSW_GPU_DB_FORM", "0")         # the plain int16 form gives the same records
    res2 = _case(ctx, seqs, seqs, mat, 5, 3, 1, chunks=(0,), check_ref=False)
    assert "frame" not in ctx.timing()["fill_kernel"]
    monkeypatch.delenv("SSW_GPU_DB_FORM")
    for f in ("score1", "score2", "ref_end1", "read_end1", "ref_end2"):
        assert (res2[f] == res[f]).all(), f
    monkeypatch.setenv("SSW_GPU_FRAME_K", "16")        # renormalised every 16 steps
    res3 = _case(ctx, seqs, seqs, mat, 5, 3, 1, chunks=(3,), check_ref=False)
    monkeypatch.delenv("SSW_GPU_FRAME_K")
    for f in ("score1", "score2", "ref_end1", "read_end1", "ref_end2"):
        assert (res3[f] == res[f]).all(), f


def test_scores_around_2048_frame_and_int16_forms_emulated(ectx, monkeypatch):
    _around_2048_check(ectx, monkeypatch)


@pytest.mark.gpu
def test_scores_around_2048_frame_and_int16_forms_gpu(gpu_hctx, monkeypatch):
    _around_2048_check(gpu_hctx, monkeypatch)      # SSW_GPU_DB_FORM / SSW_GPU_FRAME_K: libssw_hooks.so (conftest.py)


def _small_budget_check(ctx_factory, monkeypatch, nq):
    """many queries of ONE length against >= 4 targets with a small column-maximum budget -- a size class whose pairs
    do not fit a stream's slice of the scratch is cut into launches of fewer pairs instead of failing (or asking for terabytes)"""
    monkeypatch.setenv("SSW_GPU_CM_BUDGET_MB", "1")
    rng = np.random.default_rng(5)
    ref = random_ref(3000, 21, 4)
    db = [np.ascontiguousarray(ref[o:o + L]) for o, L in ((0, 900), (500, 1200), (1500, 700), (100, 333), (2000, 1000))]
    qs = [np.ascontiguousarray(ref[o:o + 100]) for o in rng.integers(0, 2900, size=nq)]
    own = ctx_factory()
    try:
        _case(own, qs, db, dna_matrix(2, 2), 5, 3, 1, chunks=(0, 2))
    finally:
        own.close()


def test_size_class_larger_than_the_scratch_budget_emulated(emu_lib_path, monkeypatch):
    _small_budget_check(lambda: ssw_amd.Context(0, ssw_amd.load(emu_lib_path)), monkeypatch, 14)


@pytest.mark.gpu
def test_size_class_larger_than_the_scratch_budget_gpu(gpu_ctx, monkeypatch):
    _small_budget_check(lambda: ssw_amd.Context(0, gpu_ctx.lib), monkeypatch, 400)


# ---- random scoring systems through the fused kernel (f16 form first): matrices up to its limit of 49, large gap penalties, all
#      three score_size modes, short mask lengths -- against the reference, pair by pair
def _random_db_sweep(ctx, seeds, nq, nt, maxlen):
    for seed in seeds:
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.choice([4, 5, 12, 24]))
        hi = int(rng.choice([2, 5, 15, 49]))
        mat = rng.integers(-min(hi, 20), hi + 1, size=(n, n)).astype(np.int8)
        mat = np.maximum(mat, mat.T)                       # symmetric like real matrices (not required, just typical)
        for k in range(n): mat[k, k] = max(1, int(rng.integers(1, hi + 1)))
        gapE = int(rng.integers(1, 6)); gapO = gapE + int(rng.integers(1, 12))
        ss = int(rng.choice([0, 1, 2])); maskLen = int(rng.choice([-1, 15, 8, 40]))
        base = rng.integers(0, n, size=maxlen, dtype=np.int8)
        def seq():
            L = int(rng.integers(1, maxlen + 1))
            if rng.random() < 0.5:                         # related to the base sequence: high scores, long diagonals
                o = int(rng.integers(0, maxlen - L + 1)); s = base[o:o + L].copy()
                flip = rng.random(L) < 0.1; s[flip] = rng.integers(0, n, size=int(flip.sum()), dtype=np.int8)
                return np.ascontiguousarray(s)
            return rng.integers(0, n, size=L, dtype=np.int8)
        qs = [seq() for _ in range(nq)]; db = [seq() for _ in range(nt)]
        Q = ctx.upload(qs); T = ctx.upload(db)
        try:
            res, cig = ctx.align_batch(Q, T, np.ascontiguousarray(mat.reshape(-1)), n, gapO, gapE, 0, 0, 0, maskLen, ss)
        finally:
            Q.free(); T.free()
        bad = compare_batch(res, cig, qs, db, np.ascontiguousarray(mat.reshape(-1)), n, gapO, gapE, 0, 0, 0, maskLen, ss)
        assert not bad, "seed %d (n %d, max %d, gaps %d/%d, score_size %d, maskLen %d):\n%s" % (seed, n, hi, gapO, gapE, ss, maskLen, "\n".join(bad[:8]))


def test_random_scoring_systems_database_path_emulated(ectx):
    _random_db_sweep(ectx, range(24), nq=6, nt=8, maxlen=150)


@pytest.mark.gpu
def test_random_scoring_systems_database_path_gpu(gpu_ctx):
    _random_db_sweep(gpu_ctx, range(200), nq=24, nt=40, maxlen=420)
