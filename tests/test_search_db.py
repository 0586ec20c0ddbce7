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


# ---- the f16 form of the fused kernel saturates at 2048: workgroups that see it repeat in the int16 form ----
def _f16_limit_case():
    """n = 5 alphabet, A-A scores 7, C-C 1, G-G 5, every mismatch -4: the self-alignment of A^292 C^k scores 2044 + k --
    2047 (exact in f16), 2048 (the saturation value itself) and 2049-2051 (beyond it), next to unrelated short sequences
    and longer homologs far above the limit; several sequences per size class so that chains of one workgroup disagree."""
    mat = np.full((5, 5), -4, dtype=np.int8)
    mat[0, 0] = 7; mat[1, 1] = 1; mat[2, 2] = 5; mat[3, 3] = 2
    mat[4, :] = 0; mat[:, 4] = 0
    rng = np.random.default_rng(77)
    seqs = [np.array([0] * 292 + [1] * k, dtype=np.int8) for k in (0, 2, 3, 4, 5, 7)]
    seqs.append(np.array([0] * 300 + [2] * 100, dtype=np.int8))          # 2600: far above, another size class
    seqs.append(np.array([0] * 120 + [1] * 30, dtype=np.int8))           # 870: stays f16
    seqs += [rng.integers(0, 4, size=int(L), dtype=np.int8) for L in (295, 299, 150, 301, 64)]
    return seqs, np.ascontiguousarray(mat.reshape(-1))


def _f16_limit_check(ctx, monkeypatch):
    seqs, mat = _f16_limit_case()
    monkeypatch.setenv("SSW_GPU_DB_F16", "1")          # f16 form first in every call (most workgroups of this case repeat: the library would pause it)
    res = _case(ctx, seqs, seqs, mat, 5, 3, 1, chunks=(4, 0))
    s = np.array([[int(res["score1"][i, j]) for j in range(6)] for i in range(6)])
    assert s[2, 2] == 2047 and s[3, 3] == 2048 and s[4, 4] == 2049 and s[5, 5] == 2051 and s[0, 5] == 2044
    assert int(res["score1"][6, 6]) == 2600
    tm = ctx.timing()
    assert "f16 first" in tm["fill_kernel"] and tm["db_repeats"] > 0, tm
    monkeypatch.setenv("SSW_GPU_DB_F16", "0")          # the int16 form alone gives the same records
    res2 = _case(ctx, seqs, seqs, mat, 5, 3, 1, chunks=(0,), check_ref=False)
    assert ctx.timing()["db_repeats"] == 0
    monkeypatch.delenv("SSW_GPU_DB_F16")
    for f in ("score1", "score2", "ref_end1", "read_end1", "ref_end2"):
        assert (res2[f] == res[f]).all(), f
    # left to itself the library notices the repeats and pauses the f16 form on this context
    own = ssw_amd.Context(0, ctx.lib)          # (a context of its own: the pause would outlive this test on the shared one)
    Q = own.upload(seqs); T = own.upload(seqs)
    try:
        own.align_batch(Q, T, mat, 5, 3, 1, 0, 0, 0, -1, 2)
        first = own.timing()
        own.align_batch(Q, T, mat, 5, 3, 1, 0, 0, 0, -1, 2)
        second = own.timing()
    finally:
        Q.free(); T.free(); own.close()
    assert first["db_repeats"] > 0 and second["db_repeats"] == 0 and "f16" not in second["fill_kernel"], (first, second)


def test_f16_form_limit_and_int16_repeat_emulated(ectx, monkeypatch):
    _f16_limit_check(ectx, monkeypatch)


@pytest.mark.gpu
def test_f16_form_limit_and_int16_repeat_gpu(gpu_ctx, monkeypatch):
    _f16_limit_check(gpu_ctx, monkeypatch)


def _streamed_pause_check(ctx_factory):
    """a database searched against itself, streamed in small chunks: the self-hits above 2048 make most workgroups of the first
    chunks repeat in the int16 form, the library then starts the following chunks in the int16 form -- same records either way"""
    seqs, mat = _f16_limit_case()
    seqs = (seqs[:5] + seqs[8:9]) * 2        # 12 entries, long self-hits in every chunk of 3
    own = ctx_factory()
    Q = own.upload(seqs); T = own.upload(seqs)
    try:
        hits = own.search_db(Q, T, mat, 5, 3, 1, -1, 2, 3)
        tm = own.timing()
        res, _ = own.align_batch(Q, T, mat, 5, 3, 1, 0, 0, 0, -1, 2)
    finally:
        Q.free(); T.free(); own.close()
    _same(hits, res)
    assert "f16 first" in tm["fill_kernel"] and tm["db_repeats"] > 0      # it started with the f16 form and noticed
    forced = ctx_factory()                  # the same with the f16 form forced in every chunk: more repeats
    os.environ["SSW_GPU_DB_F16"] = "1"
    try:
        Q = forced.upload(seqs); T = forced.upload(seqs)
        hits2 = forced.search_db(Q, T, mat, 5, 3, 1, -1, 2, 3)
        tm2 = forced.timing()
        Q.free(); T.free()
    finally:
        del os.environ["SSW_GPU_DB_F16"]; forced.close()
    _same(hits2, res)
    assert tm["db_repeats"] < tm2["db_repeats"], (tm, tm2)


def test_streamed_search_pauses_f16_form_emulated(emu_lib_path):
    _streamed_pause_check(lambda: ssw_amd.Context(0, ssw_amd.load(emu_lib_path)))


@pytest.mark.gpu
def test_streamed_search_pauses_f16_form_gpu(gpu_ctx):
    _streamed_pause_check(lambda: ssw_amd.Context(0, gpu_ctx.lib))


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
