"""ssw_gpu_search_db (include/ssw_gpu.h): the streamed database search must hand out, chunk by chunk, exactly the
score1 / score2 / ref_end1 / read_end1 / ref_end2 of a plain ssw_gpu_align_batch over the same queries and targets --
which the parity tests pin to the reference -- whatever the chunk size, also when the fused kernel does not cover the
batch (a query above 640 residues -> generic path), and must stop when the caller's function says so."""
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
