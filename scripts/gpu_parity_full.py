#!/usr/bin/env python3
"""Full-size parity on the MI355X (run through gpurun): the seeded workloads of BASELINE configs 2-5 (tests/workloads.py) through
the C-ABI of libssw.so, compared with the unmodified reference's answers that scripts/make_expected.py computed in the build
container (tests/golden/full/*.npz):

  config 2  all 100 000 reads, flag 2: every s_align field + FNV-1a of every CIGAR word; and flag 0 (what bench.py times)
  config 3  10 000 reads of read block 0 vs the 5 Mb target, flag 2
  config 4  1 500 reads (10 kb) vs the 100 kb target, maskLen 5000, flag 2
  config 5  2 048 queries x all 10 000 DB entries (2.05e7 alignments), streamed search: one checksum per query + full records of 16

One JSON line per config on stdout and in gpurun_out/parity_config<N>.json.  Usage: python scripts/gpu_parity_full.py [2 3 4 5]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "complete-striped-smith-waterman-library_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ssw_amd          # noqa: E402
import workloads as W   # noqa: E402
from sswutil import dna_matrix   # noqa: E402

FULL = os.path.join(ROOT, "tests", "golden", "full")
OUT = os.path.join(ROOT, "gpurun_out")


def fields_of(g):
    return np.stack([g["score1"], g["score2"], g["ref_begin1"], g["ref_end1"], g["read_begin1"], g["read_end1"], g["ref_end2"], g["cigarLen"],
                     g["flag"]], axis=1).astype(np.int32)


def cigar_hashes(g, cig):
    out = np.zeros(len(g), dtype=np.uint32)
    for i in range(len(g)):
        n = int(g["cigarLen"][i])
        if n > 0:
            o = int(g["cigar_off"][i])
            out[i] = W.fnv1a_words(cig[o:o + n])
    return out


def dna(ctx, cfg):
    z = np.load(os.path.join(FULL, "config%d_block0.npz" % cfg))
    exp, eh = z["fields"], z["cigar_fnv"]
    k = len(exp)
    ref, reads, p = W.dna_config(cfg, 0)
    reads = reads[:k]
    mat = dna_matrix(2, 2)
    Q = ctx.upload(list(reads)); T = ctx.upload([ref])
    line = {"config": cfg, "workload": p["name"], "alignments": int(k), "against": "unmodified reference (oracle/_ref), tests/golden/full/config%d_block0.npz" % cfg}
    t0 = time.time()
    res, cig = ctx.align_batch(Q, T, mat, 5, 3, 1, 2, 0, 0, p["mask_len"], 2)
    line["gpu_seconds_flag2"] = round(time.time() - t0, 3)
    g = res[:, 0]
    got = fields_of(g)
    bad = (got != exp).any(axis=1) | (cigar_hashes(g, cig) != eh)
    line["flag2"] = {"fields": "score1 score2 ref_begin1 ref_end1 read_begin1 read_end1 ref_end2 cigarLen flag + FNV-1a of every CIGAR word",
                     "mismatching_alignments": int(bad.sum()), "with_cigar": int((got[:, 7] > 0).sum()), "traceback_failed_flag1": int((got[:, 8] == 1).sum())}
    if bad.any():
        i = int(np.flatnonzero(bad)[0]); line["flag2"]["first_mismatch"] = {"read": i, "got": got[i].tolist(), "expected": exp[i].tolist()}
    res0, _ = ctx.align_batch(Q, T, mat, 5, 3, 1, 0, 0, 0, p["mask_len"], 2)
    g0 = fields_of(res0[:, 0])
    cols = [0, 1, 3, 5, 6]
    bad0 = (g0[:, cols] != exp[:, cols]).any(axis=1) | (g0[:, 2] != -1) | (g0[:, 4] != -1) | (g0[:, 7] != 0)
    line["flag0"] = {"fields": "score1 score2 ref_end1 read_end1 ref_end2 (begins -1, no CIGAR)", "mismatching_alignments": int(bad0.sum())}
    tm = ctx.timing()
    line["mix"] = {"word_rules": int(tm["n_word"]), "byte_rules": int(tm["n_byte"])}
    Q.free(); T.free()
    return line


def protein(ctx):
    z = np.load(os.path.join(FULL, "config5_block0.npz"))
    k, nt = int(z["nq"]), int(z["nt"])
    db, qs, mat = W.protein_config(0)
    qs = qs[:k]
    Q = ctx.upload(qs); T = ctx.upload(db)
    t0 = time.time()
    hits = ctx.search_db(Q, T, mat, 24, 3, 1, -1, 2, 512)
    secs = time.time() - t0
    rows = np.stack([hits["score1"], hits["score2"], hits["ref_end1"], hits["read_end1"], hits["ref_end2"]], axis=2).astype(np.int32)
    bad_rows = int((W.row_checksums(rows.reshape(k, -1)) != z["row_checksum"]).sum())
    bad16 = int((rows[:16] != z["first16"]).any(axis=2).sum())
    Q.free(); T.free()
    return {"config": 5, "workload": "BASELINE config 5 (first %d queries of query block 0 x all %d DB entries), streamed search" % (k, nt),
            "alignments": k * nt, "gpu_seconds": round(secs, 3), "queries_with_wrong_checksum": bad_rows, "mismatching_alignments_in_first_16_queries": bad16,
            "fields": "score1 score2 ref_end1 read_end1 ref_end2", "against": "unmodified reference (oracle/_ref), tests/golden/full/config5_block0.npz"}


def main():
    which = [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5]
    os.makedirs(OUT, exist_ok=True)
    ctx = ssw_amd.Context(0)
    for cfg in which:
        line = protein(ctx) if cfg == 5 else dna(ctx, cfg)
        s = json.dumps(line)
        print(s, flush=True)
        with open(os.path.join(OUT, "parity_config%d.json" % cfg), "w") as f:
            f.write(s + "\n")
    ctx.close()


if __name__ == "__main__":
    main()
