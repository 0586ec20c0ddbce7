#!/usr/bin/env python3
"""Reference results for the full-size parity runs (scripts/gpu_parity_full.py), computed HERE (build container) by the
unmodified reference -- oracle/_ref/libssw_ref.so, built from /root/reference/src/ssw.c by oracle/Makefile -- on the
seeded workloads of tests/workloads.py, and committed as compressed fixtures under tests/golden/full/.

  config2_block0.npz  all 100 000 reads of BASELINE config 2, flag 2 (scores, ends, begins, CIGAR length + FNV-1a of the CIGAR)
  config3_block0.npz  all 20 000 reads of read block 0 of config 3 (5 Mb target), flag 2
  config4_block0.npz  all 10 000 reads of config 4 (10 kb reads, 100 kb target, maskLen 5000), flag 2
  config2_u8_block0.npz   (`20`) the 100 000 reads of config 2 under 1/-3/5/2 (pure 8-bit rules), flag 2
  config6_block0.npz  (`6`) the 1000 mixed-length reads of the README's benchmark shape vs the 4.94 Mb genome, defaults and -m1 -x3 -o5 -e2, flag 2
  config{2,3}_blocks_sample.npz  (`21`, `31`) a seeded 2 000-read sample of read blocks 0..7 (rank r of an N-GPU run works on block r), flag 2
  config3_blocks_sample_8_49.npz (`32`) a seeded 500-read sample of read blocks 8..49: with blocks 0..7 above, all 50 blocks of config 3 at its stated size
  config5_block0.npz  first 2 048 queries of query block 0 against all 10 000 DB entries (2.05e7 alignments): one 64-bit
                      checksum per query over its 10 000 x (score1 score2 ref_end1 read_end1 ref_end2), full records of
                      the first 16 queries

Flag-2 records also check score-only (flag 0) runs: score1, score2, ref_end1, read_end1, ref_end2 do not depend on the flag.
Usage: python scripts/make_expected.py [2 3 4 5] [--threads N]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sswutil import _ptr, dna_matrix, i8p, i32p, i64p, ref_lib, u32p   # noqa: E402
import workloads as W   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "full")
FIELDS = "score1 score2 ref_begin1 ref_end1 read_begin1 read_end1 ref_end2 cigarLen flag"


def run_dna(R, cfg, count, threads, scoring=(2, 2, 3, 1), tag=""):
    ref, reads, p = W.dna_config(cfg, 0)
    reads = np.ascontiguousarray(reads[:count])
    off = np.arange(count + 1, dtype=np.int64) * p["read_len"]
    mat = dna_matrix(scoring[0], scoring[1])
    res = np.zeros((count, 10), dtype=np.int32)
    hsh = np.zeros(count, dtype=np.uint32)
    secs = R.refwrap_bench_hash(_ptr(reads, i8p), _ptr(off, i64p), count, _ptr(ref, i8p), len(ref), _ptr(mat, i8p), 5, scoring[2], scoring[3], 2, 0, 0,
                                p["mask_len"], threads, _ptr(res, i32p), _ptr(hsh, u32p))
    assert (res[:, 9] == 0).all()
    cells = float(count) * p["read_len"] * p["ref_len"]
    np.savez_compressed(os.path.join(OUT, "config%d%s_block0.npz" % (cfg, tag)), fields=res[:, :9], cigar_fnv=hsh,
                        meta=np.array([cfg, count, p["read_len"], p["ref_len"], p["seed_ref"], p["seed_reads"]], dtype=np.int64),
                        scoring=np.array(scoring, dtype=np.int64))
    print("config %d%s: %d reads, %.1f s on %d threads, %.1f GCUPS; fields: %s" % (cfg, tag, count, secs, threads, cells / secs / 1e9, FIELDS), flush=True)


def run_mixed(R, threads):
    """config 6 (the README's benchmark shape, tests/workloads.py mixed_config), both scorings the README quotes, flag 2"""
    ref, reads, p = W.mixed_config(0)
    qc, qo = W.pack(reads)
    n = len(reads)
    out = {}
    for tag, sc in (("default", (2, 2, 3, 1)), ("m1x3o5e2", (1, 3, 5, 2))):
        mat = dna_matrix(sc[0], sc[1])
        res = np.zeros((n, 10), dtype=np.int32)
        hsh = np.zeros(n, dtype=np.uint32)
        secs = R.refwrap_bench_hash(_ptr(qc, i8p), _ptr(qo, i64p), n, _ptr(ref, i8p), len(ref), _ptr(mat, i8p), 5, sc[2], sc[3], 2, 0, 0,
                                    -1, threads, _ptr(res, i32p), _ptr(hsh, u32p))
        assert (res[:, 9] == 0).all()
        out["fields_" + tag] = res[:, :9].copy(); out["cigar_fnv_" + tag] = hsh
        cells = float(qo[-1]) * len(ref)
        print("config 6 %s: %d reads (%d..%d bp, mean %.0f), %.1f s on %d threads, %.1f GCUPS" % (tag, n, min(map(len, reads)), max(map(len, reads)),
                                                                                              qo[-1] / n, secs, threads, cells / secs / 1e9), flush=True)
    np.savez_compressed(os.path.join(OUT, "config6_block0.npz"), lens=np.diff(qo), **out)


SAMPLE = 2000      # reads per block in the per-rank fixtures


def sample_indices(cfg, block, nreads, k=SAMPLE):
    """the seeded read sample of block `block` that the per-rank parity of an N-GPU bench line is checked on"""
    return np.sort(np.random.default_rng(770_000 + 100 * cfg + block).choice(nreads, size=min(k, nreads), replace=False))


def run_dna_blocks(R, cfg, blocks, threads, first=0, k=SAMPLE, tag=""):
    """per-rank fixtures: rank r of `bench.py --gpus N` works on read block r; a seeded sample of `k` reads of every block
    first .. blocks-1, flag 2 (the five score / end fields also check the score-only run)"""
    idxs, fields, hashes = [], [], []
    mat = dna_matrix(2, 2)
    t0 = time.time()
    for b in range(first, blocks):
        ref, reads, p = W.dna_config(cfg, b)
        idx = sample_indices(cfg, b, len(reads), k)
        sub = np.ascontiguousarray(reads[idx])
        off = np.arange(len(idx) + 1, dtype=np.int64) * p["read_len"]
        res = np.zeros((len(idx), 10), dtype=np.int32)
        hsh = np.zeros(len(idx), dtype=np.uint32)
        R.refwrap_bench_hash(_ptr(sub, i8p), _ptr(off, i64p), len(idx), _ptr(ref, i8p), len(ref), _ptr(mat, i8p), 5, 3, 1, 2, 0, 0,
                             p["mask_len"], threads, _ptr(res, i32p), _ptr(hsh, u32p))
        assert (res[:, 9] == 0).all()
        idxs.append(idx); fields.append(res[:, :9].copy()); hashes.append(hsh)
        print("config %d block %d: %d sampled reads done, %.0f s" % (cfg, b, len(idx), time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(OUT, "config%d_blocks_sample%s.npz" % (cfg, tag)), idx=np.stack(idxs), fields=np.stack(fields), cigar_fnv=np.stack(hashes),
                        meta=np.array([cfg, blocks, k, first], dtype=np.int64))


def run_protein(R, nq, threads):
    db, qs, mat = W.protein_config(0)
    qs = qs[:nq]
    qc, qo = W.pack(qs); tc, to = W.pack(db)
    res = np.zeros((nq, len(db), 5), dtype=np.int32)
    secs = R.refwrap_bench_db(_ptr(qc, i8p), _ptr(qo, i64p), nq, _ptr(tc, i8p), _ptr(to, i64p), len(db), _ptr(mat, i8p), 24, 3, 1, -1, threads,
                              _ptr(res, i32p))
    assert (res != -9).all()
    sums = W.row_checksums(res.reshape(nq, -1))
    np.savez_compressed(os.path.join(OUT, "config5_block0.npz"), row_checksum=sums, first16=res[:16], nq=np.int64(nq), nt=np.int64(len(db)))
    cells = float(qo[-1]) * float(to[-1])
    print("config 5: %d queries x %d entries, %.1f s on %d threads, %.1f GCUPS" % (nq, len(db), secs, threads, cells / secs / 1e9), flush=True)


def run_protein_full(R, threads, block=2048):
    """all 50 000 x 10 000 alignments of config 5 (query block 0 of the bench = the stated size): order-independent checksums
    per block of 2048 queries over all 10 000 entries (tests/workloads.py words_checksum)"""
    db, qs, mat = W.protein_config(0)
    tc, to = W.pack(db)
    out = []
    t0 = time.time()
    for b0 in range(0, len(qs), block):
        sub = qs[b0:b0 + block]
        qc, qo = W.pack(sub)
        res = np.zeros((len(sub), len(db), 5), dtype=np.int32)
        R.refwrap_bench_db(_ptr(qc, i8p), _ptr(qo, i64p), len(sub), _ptr(tc, i8p), _ptr(to, i64p), len(db), _ptr(mat, i8p), 24, 3, 1, -1, threads,
                           _ptr(res, i32p))
        assert (res != -9).all()
        w0, w1 = W.hit_words(res[..., 0], res[..., 1], res[..., 2], res[..., 3], res[..., 4])
        out.append(W.words_checksum(w0, w1))
        print("config 5 full: queries %d..%d done, %.0f s" % (b0, b0 + len(sub) - 1, time.time() - t0), flush=True)
        np.savez_compressed(os.path.join(OUT, "config5_full_block0.npz"), block=np.int64(block), nq=np.int64(len(qs)), nt=np.int64(len(db)),
                            done=np.int64(b0 + len(sub)), xor0=np.array([o[0] for o in out], dtype=np.uint64),
                            xor1=np.array([o[1] for o in out], dtype=np.uint64), sums=np.array([o[2] for o in out], dtype=np.uint64))


def main():
    args = sys.argv[1:]
    threads = max(1, (os.cpu_count() or 2) - 1)
    if "--threads" in args:
        i = args.index("--threads"); threads = int(args[i + 1]); del args[i:i + 2]
    which = [int(a) for a in args] or [4, 5, 3, 2]
    R = ref_lib(required=True)
    os.makedirs(OUT, exist_ok=True)
    t0 = time.time()
    for cfg in which:
        if cfg == 2:
            run_dna(R, 2, 100_000, threads)
        elif cfg == 3:
            run_dna(R, 3, 20_000, threads)
        elif cfg == 4:
            run_dna(R, 4, 10_000, threads)
        elif cfg == 20:      # config 2 under the pure 8-bit scoring of SURVEY 8d (ii): 1/-3/5/2, every read decided by the u8 kernel
            run_dna(R, 2, 100_000, threads, scoring=(1, 3, 5, 2), tag="_u8")
        elif cfg == 6:
            run_mixed(R, threads)
        elif cfg == 21:      # per-rank fixtures of config 2 (blocks 0..7)
            run_dna_blocks(R, 2, 8, threads)
        elif cfg == 31:
            run_dna_blocks(R, 3, 8, threads)
        elif cfg == 32:      # config 3 at its stated size (1M reads = 50 blocks of 20 000): a 500-read sample of blocks 8..49
            run_dna_blocks(R, 3, 50, threads, first=8, k=500, tag="_8_49")
        elif cfg == 5:
            run_protein(R, 2048, threads)
        elif cfg == 50:
            run_protein_full(R, threads)
    print("done in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
