"""BASELINE.json's configs 2-5 at their stated sizes on the MI355X, through the C-ABI of libssw.so, against the answers of the
unmodified reference (oracle/_ref, built from /root/reference/src/ssw.c) that scripts/make_expected.py computed in the build
container and committed under tests/golden/full/ -- the same comparison scripts/gpu_parity_full.py prints, as pytest, so that the
driver's `pytest -m gpu` run sees the full-size inputs (a few GPU-seconds in total):

  config 2  all 100 000 reads x 150 bp vs the 1 Mb target: flag 2 (every s_align field + FNV-1a of every CIGAR word) and flag 0
  config 3  the whole 20 000-read block 0 vs the 5 Mb target, flag 2 and flag 0
  config 4  all 10 000 x 10 kb reads vs the 100 kb target, maskLen 5000, flag 2 (banded traceback on the GPU)
  config 2 under the pure 8-bit scoring 1/-3/5/2 (all 100 000 reads), a seeded sample of read blocks 1..7 of configs 2 and 3 (what the
  ranks of an N-GPU run compute), and the README's benchmark shape (config 6: 1000 mixed-length reads vs 4.94 Mb, both scorings)
  config 5  2 048 queries x all 10 000 DB entries (2.05e7 alignments), streamed database search: one checksum per query and the
            full records of the first 16 queries

plus one random database search in which SOME workgroups of a chunk hold high-scoring pairs and others do not (a scoring system
that lets self-alignments of config-5-length proteins pass 2048, next to unrelated sequences), against the reference itself.
"""
import os

import numpy as np
import pytest

import workloads as W
from sswutil import blosum50, dna_matrix

pytestmark = pytest.mark.gpu
FULL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full")


def _fields(g):
    return np.stack([g["score1"], g["score2"], g["ref_begin1"], g["ref_end1"], g["read_begin1"], g["read_end1"], g["ref_end2"], g["cigarLen"],
                     g["flag"]], axis=1).astype(np.int32)


def _cigar_hashes(g, cig):
    out = np.zeros(len(g), dtype=np.uint32)
    for i in range(len(g)):
        n = int(g["cigarLen"][i])
        if n > 0:
            o = int(g["cigar_off"][i])
            out[i] = W.fnv1a_words(cig[o:o + n])
    return out


def _dna_full(ctx, cfg, min_reads, scoring=(2, 2, 3, 1), tag=""):
    z = np.load(os.path.join(FULL, "config%d%s_block0.npz" % (cfg, tag)))
    exp, eh = z["fields"], z["cigar_fnv"]
    k = len(exp)
    assert k >= min_reads, "tests/golden/full/config%d%s_block0.npz holds %d reads (scripts/make_expected.py)" % (cfg, tag, k)
    ref, reads, p = W.dna_config(cfg, 0)
    reads = reads[:k]
    mat = dna_matrix(scoring[0], scoring[1])
    gO, gE = scoring[2], scoring[3]
    Q = ctx.upload(list(reads)); T = ctx.upload([ref])
    try:
        res, cig = ctx.align_batch(Q, T, mat, 5, gO, gE, 2, 0, 0, p["mask_len"], 2)
        g = res[:, 0]
        got = _fields(g)
        bad = (got != exp).any(axis=1) | (_cigar_hashes(g, cig) != eh)
        assert not bad.any(), "config %d flag 2: %d of %d alignments differ from the reference; first: read %d got %s expected %s" % (
            cfg, int(bad.sum()), k, int(np.flatnonzero(bad)[0]), got[np.flatnonzero(bad)[0]].tolist(), exp[np.flatnonzero(bad)[0]].tolist())
        assert (got[:, 7] > 0).sum() > 0.9 * k                   # the CIGARs were really produced (and compared)
        if cfg != 4:                                             # (config 4 is stated with the CIGAR on)
            res0, _ = ctx.align_batch(Q, T, mat, 5, gO, gE, 0, 0, 0, p["mask_len"], 2)
            if tag == "_u8":                                     # max score 150 < 255 - 3, every read decided by the 8-bit rules
                tm = ctx.timing()
                assert tm["n_word"] == 0 and tm["n_byte"] > 0.9 * k
            g0 = _fields(res0[:, 0])
            cols = [0, 1, 3, 5, 6]
            bad0 = (g0[:, cols] != exp[:, cols]).any(axis=1) | (g0[:, 2] != -1) | (g0[:, 4] != -1) | (g0[:, 7] != 0) | (g0[:, 8] != 0)
            assert not bad0.any(), "config %d flag 0: %d of %d alignments differ; first: read %d" % (cfg, int(bad0.sum()), k, int(np.flatnonzero(bad0)[0]))
    finally:
        Q.free(); T.free()
    return k


def test_config2_all_100k_reads_vs_1mb(gpu_ctx):
    assert _dna_full(gpu_ctx, 2, 100_000) == 100_000


def test_config3_whole_read_block_vs_5mb(gpu_ctx):
    assert _dna_full(gpu_ctx, 3, 20_000) == 20_000


def test_config4_long_reads_with_traceback(gpu_ctx):
    assert _dna_full(gpu_ctx, 4, 10_000) == 10_000


def test_config2_pure_8bit_scoring(gpu_ctx):
    """the 100 000 reads of config 2 under match 1 / mismatch 3 / gaps 5, 2 (README.md:71) -- pure u8 semantics"""
    assert _dna_full(gpu_ctx, 2, 100_000, scoring=(1, 3, 5, 2), tag="_u8") == 100_000


def test_per_rank_blocks_of_configs_2_and_3(gpu_ctx):
    """What rank r of an N-GPU bench line computes: read block r.  The seeded 2 000-read sample of blocks 1..7 that
    bench.py's `parity.per_rank` checks, here through the C-ABI in one process (block 0 is covered in full above)."""
    mat = dna_matrix(2, 2)
    for cfg in (2, 3):
        z = np.load(os.path.join(FULL, "config%d_blocks_sample.npz" % cfg))
        ref = None
        T = None
        try:
            for b in (1, len(z["idx"]) - 1):                      # two of the blocks keep the test short; bench.py checks each rank's own
                ref_b, reads, p = W.dna_config(cfg, b)
                if T is None:
                    ref = ref_b; T = gpu_ctx.upload([ref])
                sub = reads[z["idx"][b]]
                Q = gpu_ctx.upload(list(sub))
                try:
                    res, cig = gpu_ctx.align_batch(Q, T, mat, 5, 3, 1, 2, 0, 0, p["mask_len"], 2)
                finally:
                    Q.free()
                g = res[:, 0]
                bad = (_fields(g) != z["fields"][b]).any(axis=1) | (_cigar_hashes(g, cig) != z["cigar_fnv"][b])
                assert not bad.any(), "config %d block %d: %d of %d sampled reads differ from the reference" % (cfg, b, int(bad.sum()), len(sub))
        finally:
            if T is not None:
                T.free()


def test_config6_mixed_read_lengths_readme_shape(gpu_ctx):
    """the shape of the one benchmark the reference publishes (README.md:62-74): 1000 reads of 25-540 bp vs a 4.94 Mb genome,
    default penalties and -m1 -x3 -o5 -e2 -- ~33 geometry buckets with a few pairs each, the longest reads on the strip kernel"""
    z = np.load(os.path.join(FULL, "config6_block0.npz"))
    ref, reads, p = W.mixed_config(0)
    assert (np.array([len(r) for r in reads]) == z["lens"]).all()
    Q = gpu_ctx.upload(reads); T = gpu_ctx.upload([ref])
    try:
        for key, sc in (("default", (2, 2, 3, 1)), ("m1x3o5e2", (1, 3, 5, 2))):
            mat = dna_matrix(sc[0], sc[1])
            res, cig = gpu_ctx.align_batch(Q, T, mat, 5, sc[2], sc[3], 2, 0, 0, -1, 2)
            g = res[:, 0]
            bad = (_fields(g) != z["fields_" + key]).any(axis=1) | (_cigar_hashes(g, cig) != z["cigar_fnv_" + key])
            assert not bad.any(), "config 6 (%s): %d of %d reads differ; first: read %d (len %d)" % (
                key, int(bad.sum()), len(reads), int(np.flatnonzero(bad)[0]), len(reads[int(np.flatnonzero(bad)[0])]))
    finally:
        Q.free(); T.free()


def test_config5_streamed_database_search(gpu_ctx):
    z = np.load(os.path.join(FULL, "config5_block0.npz"))
    k, nt = int(z["nq"]), int(z["nt"])
    db, qs, mat = W.protein_config(0)
    assert nt == len(db)
    qs = qs[:k]
    Q = gpu_ctx.upload(qs); T = gpu_ctx.upload(db)
    try:
        hits = gpu_ctx.search_db(Q, T, mat, 24, 3, 1, -1, 2, 512)
    finally:
        Q.free(); T.free()
    rows = np.stack([hits["score1"], hits["score2"], hits["ref_end1"], hits["read_end1"], hits["ref_end2"]], axis=2).astype(np.int32)
    assert (rows[:16] == z["first16"]).all()
    wrong = np.flatnonzero(W.row_checksums(rows.reshape(k, -1)) != z["row_checksum"])
    assert len(wrong) == 0, "config 5: %d of %d queries have a wrong checksum over their %d records (first: query %d)" % (len(wrong), k, nt, int(wrong[0]))


def test_database_search_mixed_high_and_low_scoring_workgroups(gpu_ctx, reflib):
    """config-5 length distribution, a scoring system with large matches (x3 BLOSUM50: self-alignments of 300-aa proteins score
    ~ 6000, far above 2048), queries that are copies / mutated copies of SOME DB entries: in one chunk a few workgroups hold
    high-scoring pairs and most do not.  Every record against the reference."""
    from sswutil import _ptr, i8p, i32p, i64p, mutate
    rng = np.random.default_rng(77)
    db, qs, _ = W.protein_config(3, queries=96, db_entries=700)
    mat = np.clip(blosum50().astype(np.int32) * 3, -128, 49).astype(np.int8)          # (the fused kernel takes entries up to 49)
    db = list(db)
    for i in range(0, 96, 3):                                     # every third query: an exact or mutated copy of a DB entry
        src = db[int(rng.integers(0, len(db)))]
        qs[i] = np.ascontiguousarray(src if i % 2 == 0 else mutate(src, rng, 0.05, 0.01, 0.01, 20)[:1000])
    Q = gpu_ctx.upload(qs); T = gpu_ctx.upload(db)
    try:
        hits = gpu_ctx.search_db(Q, T, mat, 24, 5, 2, -1, 2, 256)
        tm = gpu_ctx.timing()
    finally:
        Q.free(); T.free()
    got = np.stack([hits["score1"], hits["score2"], hits["ref_end1"], hits["read_end1"], hits["ref_end2"]], axis=2).astype(np.int32)
    qc, qo = W.pack(qs); tc, to = W.pack(db)
    exp = np.zeros((len(qs), len(db), 5), dtype=np.int32)
    reflib.refwrap_bench_db(_ptr(qc, i8p), _ptr(qo, i64p), len(qs), _ptr(tc, i8p), _ptr(to, i64p), len(db), _ptr(mat, i8p), 24, 5, 2, -1, 4, _ptr(exp, i32p))
    bad = np.argwhere((got != exp).any(axis=2))
    assert len(bad) == 0, "%d of %d records differ; first (query, entry) %s: got %s expected %s" % (
        len(bad), got.shape[0] * got.shape[1], bad[0].tolist(), got[tuple(bad[0])].tolist(), exp[tuple(bad[0])].tolist())
    assert (exp[..., 0] >= 2048).sum() >= 16 and (exp[..., 0] < 2048).mean() > 0.9      # the mix the test is about
    assert tm["fill_launches"] >= 1
