"""Begin positions / CIGARs against MANY targets in one batch call (the reference's loop: ssw_align with flag 2 and a score filter for
every (read, target) pair, src/main.c:493-506; gating src/ssw.c:916, 938).  The flagged database path (ssw_host.c dbx_chunk: fused
search -> k_select -> one batched reverse pass + traceback over the survivors as (query, target) jobs) must return, for every pair,
exactly what the reference returns -- all s_align fields and every CIGAR word -- for every flag / filter combination, in chunks of
targets too, and the same records as the per-target loop it replaces (SSW_GPU_NO_DBX=1)."""
import os

import numpy as np
import pytest

import ssw_amd
import workloads as W
from sswutil import _ptr, blosum50, dna_matrix, i8p, i32p, i64p, mutate, random_ref, u32p

FIELDS = ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2", "cigarLen", "flag")


@pytest.fixture(scope="module")
def ectx(emu_lib_path):
    ctx = ssw_amd.Context(0, ssw_amd.load(emu_lib_path))
    yield ctx
    ctx.close()


def reference_records(reflib, qs, db, mat, n, gapO, gapE, flag, filters, filterd, maskLen, threads=4):
    qc, qo = W.pack(qs) if sum(len(q) for q in qs) else (np.zeros(1, dtype=np.int8), np.zeros(len(qs) + 1, dtype=np.int64))
    tc, to = W.pack(db)
    res = np.zeros((len(qs), len(db), 10), dtype=np.int32)
    hsh = np.zeros((len(qs), len(db)), dtype=np.uint32)
    reflib.refwrap_bench_dbx(_ptr(qc, i8p), _ptr(qo, i64p), len(qs), _ptr(tc, i8p), _ptr(to, i64p), len(db), _ptr(mat, i8p), n, gapO, gapE,
                             flag, filters, filterd, maskLen, threads, _ptr(res, i32p), _ptr(hsh, u32p))
    return res, hsh


def check_against_reference(res, cig, exp, exph, what=""):
    got = np.stack([res[f] for f in FIELDS], axis=2).astype(np.int32)
    null = exp[..., 9] == 1
    assert (res["status"][null] == 1).all() and (res["status"][~null] == 0).all(), what
    bad = np.argwhere((got != exp[..., :9]).any(axis=2) & ~null)
    assert len(bad) == 0, "%s: %d of %d records differ; first (query, target) %s: got %s expected %s" % (
        what, len(bad), got.shape[0] * got.shape[1], bad[0].tolist(), got[tuple(bad[0])].tolist(), exp[tuple(bad[0])].tolist())
    n_cig = 0
    for q, t in np.argwhere(res["cigarLen"] > 0):
        o = int(res["cigar_off"][q, t]); k = int(res["cigarLen"][q, t])
        assert W.fnv1a_words(cig[o:o + k]) == int(exph[q, t]), "%s: CIGAR of (query %d, target %d) differs" % (what, q, t)
        n_cig += 1
    assert ((res["cigarLen"] == 0) == (exph == 0)).all(), what
    return n_cig


def _protein_case(seed=11, nq=14, nt=23):
    """config-5-like proteins with planted homologs (so that a score filter separates them), one query above 384 residues
    (the 385..640 classes of the fused kernel, window passes on the strip kernel), an empty query and an empty target"""
    rng = np.random.default_rng(seed)
    db, qs, mat = W.protein_config(7, queries=nq, db_entries=nt)
    qs = [np.ascontiguousarray(q[:200]) for q in qs]
    db = [np.ascontiguousarray(t[:220]) for t in db]
    for i in range(0, nq, 3):
        qs[i] = np.ascontiguousarray(mutate(db[int(rng.integers(0, nt))], rng, 0.1, 0.02, 0.02, 20)[:200])
    qs[4] = np.ascontiguousarray(np.concatenate([db[3], db[5]])[:430])           # 430 residues: a class of 385..640
    qs[7] = np.zeros(0, dtype=np.int8)
    db[9] = np.zeros(0, dtype=np.int8)
    return qs, db, mat


@pytest.mark.parametrize("flag,filters,filterd", [(2, -85, 0), (2, 0, 0), (1, 0, 0), (8, 0, 0), (6, -50, 180), (10, -70, 0), (15, 90, 150)])
def test_flagged_database_search_vs_reference_emulated(ectx, reflib, flag, filters, filterd):
    qs, db, mat = _protein_case()
    auto = filters < 0
    if auto:      # a score filter at that percentile of the pairs' scores (gaps 3/1 on BLOSUM50: unrelated pairs score in the hundreds too)
        e0, _ = reference_records(reflib, qs, db, mat, 24, 3, 1, 0, 0, 0, -1)
        filters = int(np.percentile(e0[..., 0], -filters))
    exp, exph = reference_records(reflib, qs, db, mat, 24, 3, 1, flag, filters, filterd, -1)
    Q = ectx.upload(qs); T = ectx.upload(db)
    try:
        res, cig = ectx.align_batch(Q, T, mat, 24, 3, 1, flag, filters, filterd, -1, 2)
        tm = ectx.timing()
    finally:
        Q.free(); T.free()
    n_cig = check_against_reference(res, cig, exp, exph, "flag %d filters %d filterd %d" % (flag, filters, filterd))
    assert "k_filldb" in tm["fill_kernel"]                     # the fused search ran (not the per-target loop)
    if flag == 2 and auto:
        assert 0 < n_cig < 0.3 * len(qs) * len(db)             # the filter really separated the high-scoring pairs from the rest
    if flag == 8:
        assert n_cig == 0 and (res["ref_begin1"][exp[..., 0] > 0] >= 0).all()


def test_flagged_database_search_in_target_chunks_equals_the_per_target_loop_emulated(ectx, reflib):
    """a scratch budget that cuts the targets into chunks (and the survivors into traceback slabs), next to the per-target loop
    the path replaces: the same records and the same CIGARs; DNA with a query above 640 residues (not fused: per-target path
    for that query, in the same call)"""
    rng = np.random.default_rng(3)
    ref = random_ref(4000, 21, 4)
    db = [np.ascontiguousarray(ref[o:o + L]) for o, L in ((0, 300), (200, 190), (700, 411), (50, 133), (900, 260), (1500, 350), (2100, 280), (3000, 500))]
    qs = [np.ascontiguousarray(ref[100:800]), np.ascontiguousarray(ref[210:360]), rng.integers(0, 4, size=77, dtype=np.int8),
          np.ascontiguousarray(mutate(ref[1520:1800], rng, 0.05, 0.01, 0.01, 4)), np.ascontiguousarray(ref[2150:2300])]
    mat = dna_matrix(2, 2)
    exp, exph = reference_records(reflib, qs, db, mat, 5, 3, 1, 2, 60, 0, -1)
    Q = ectx.upload(qs); T = ectx.upload(db)
    old = ectx.lib.ssw_gpu_get_budget(ectx.h)
    try:
        ectx.lib.ssw_gpu_set_budget(ectx.h, 1 << 20)
        os.environ["SSW_GPU_DB_TSUB"] = "3"; os.environ["SSW_GPU_DBX_SLAB"] = "4"      # chunks of 3 targets, traceback slabs of 4 survivors
        try:
            res, cig = ectx.align_batch(Q, T, mat, 5, 3, 1, 2, 60, 0, -1, 2, mark_mismatch=False)
        finally:
            del os.environ["SSW_GPU_DB_TSUB"], os.environ["SSW_GPU_DBX_SLAB"]
        n_cig = check_against_reference(res, cig, exp, exph, "chunked")
        assert n_cig > 8                                        # more survivors than one slab of one chunk holds
        os.environ["SSW_GPU_NO_DBX"] = "1"
        try:
            res2, cig2 = ectx.align_batch(Q, T, mat, 5, 3, 1, 2, 60, 0, -1, 2)
        finally:
            del os.environ["SSW_GPU_NO_DBX"]
        check_against_reference(res2, cig2, exp, exph, "per-target loop")
        # SAM-style rewrite (mark_mismatch on the device) through both paths: same words, same edit distance
        r3, c3 = ectx.align_batch(Q, T, mat, 5, 3, 1, 2, 60, 0, -1, 2, mark_mismatch=True)
        os.environ["SSW_GPU_NO_DBX"] = "1"
        try:
            r4, c4 = ectx.align_batch(Q, T, mat, 5, 3, 1, 2, 60, 0, -1, 2, mark_mismatch=True)
        finally:
            del os.environ["SSW_GPU_NO_DBX"]
        assert (r3["cigarLen"] == r4["cigarLen"]).all() and (r3["edit_distance"] == r4["edit_distance"]).all()
        for q, t in np.argwhere(r3["cigarLen"] > 0):
            a = c3[int(r3["cigar_off"][q, t]):int(r3["cigar_off"][q, t]) + int(r3["cigarLen"][q, t])]
            b = c4[int(r4["cigar_off"][q, t]):int(r4["cigar_off"][q, t]) + int(r4["cigarLen"][q, t])]
            assert (a == b).all()
    finally:
        ectx.lib.ssw_gpu_set_budget(ectx.h, old)
        Q.free(); T.free()


@pytest.mark.gpu
@pytest.mark.parametrize("flag,filters,filterd", [(2, 400, 0), (2, 0, 0), (6, 300, 400)])
def test_flagged_database_search_vs_reference_gpu(gpu_ctx, reflib, flag, filters, filterd):
    """config-5 shapes on the MI355X: 192 queries x 400 entries with planted homologs; filters 400 keeps the homologs and the
    high tail of the unrelated pairs (gaps 3/1 on BLOSUM50 are the linear regime: unrelated 300 x 300 pairs score ~330),
    filters 0 sends all 76 800 pairs through the reverse pass and the traceback"""
    db, qs, mat = W.protein_config(2, queries=192, db_entries=400)
    exp, exph = reference_records(reflib, qs, db, mat, 24, 3, 1, flag, filters, filterd, -1, threads=16)
    Q = gpu_ctx.upload(qs); T = gpu_ctx.upload(db)
    try:
        res, cig = gpu_ctx.align_batch(Q, T, mat, 24, 3, 1, flag, filters, filterd, -1, 2)
        tm = gpu_ctx.timing()
    finally:
        Q.free(); T.free()
    n_cig = check_against_reference(res, cig, exp, exph, "flag %d filters %d filterd %d" % (flag, filters, filterd))
    assert n_cig > 0
    if os.environ.get("SSW_GPU_NO_DBX") != "1":                     # (the suite is also run with the per-target loop forced: same records)
        assert "k_filldb" in tm["fill_kernel"]
