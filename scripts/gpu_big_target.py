#!/usr/bin/env python3
"""GPU box: short reads against ONE very long target (a chromosome, not a bacterial genome) -- the sizes where 32-bit products of columns and
bytes would overflow if any were left: 50 Mb, 250 Mb and (with --gb) 1 Gb, batch ABI with flag 0 and flag 2 and the single-pair drop-in ABI,
every record and CIGAR against the unmodified reference (oracle/_ref; its ssw_align takes an int32 refLen too).  Reads: copies from the start,
the middle and the LAST columns of the target (with substitutions and indels), an unrelated one, 150 and 400 residues.
usage: gpu_big_target.py [--gb]        -> one JSON line per target length"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "complete-striped-smith-waterman-library_amd"))
import ssw_amd          # noqa: E402
from parity import compare_batch   # noqa: E402
from sswutil import dna_matrix, mutate   # noqa: E402

ctx = ssw_amd.Context(0)
mat = dna_matrix(2, 2)
sizes = [50_000_000, 250_000_000] if "--gb" not in sys.argv else [1_000_000_000, 2_000_000_000]      # (--gb: two reads only -- the reference needs a minute per read there; int32 ends at 2 147 483 647)
for L in sizes:
    rng = np.random.default_rng(L % 1000 + 7)
    ref = rng.integers(0, 4, size=L, dtype=np.int8)
    reads = []
    for off, rl in ((0, 150), (L // 2 + 12345, 400), (L - 150, 150), (L - 401, 400)) if L < 1_000_000_000 else ((L - 150, 150),):
        reads.append(np.asarray(mutate(ref[off:off + rl].copy(), rng, 0.03, 0.01, 0.01, 4), dtype=np.int8))
    reads.append(rng.integers(0, 4, size=150, dtype=np.int8))
    out = {"target_len": L, "reads": [len(r) for r in reads]}
    for flag in (0, 2):
        Q = ctx.upload(reads); T = ctx.upload([ref])
        t0 = time.time()
        try:
            res, cig = ctx.align_batch(Q, T, mat, 5, 3, 1, flag, 0, 0, -1, 2)
            out["flag%d_seconds" % flag] = round(time.time() - t0, 3)
            t0 = time.time()
            bad = compare_batch(res, cig, reads, [ref], mat, 5, 3, 1, flag, 0, 0, -1, 2, max_report=2)
            out["flag%d_reference_seconds" % flag] = round(time.time() - t0, 1)
            out["flag%d_wrong" % flag] = len(bad)
            if bad: out["flag%d_first" % flag] = bad[0][:400]
            out["flag%d_score1_ref_end1" % flag] = [[int(r["score1"]), int(r["ref_end1"])] for r in res[:, 0]]
        except Exception as e:      # noqa: BLE001
            out["flag%d_failed" % flag] = str(e)[:300]
        finally:
            Q.free(); T.free()
    print(json.dumps(out), flush=True)
