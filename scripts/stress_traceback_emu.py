#!/usr/bin/env python3
"""Build container (no GPU): randomized stress of the traceback kernels on the SIMT emulator -- reads with random deletions / insertions /
unrelated reads, random scoring, teams of 1 / 4 / 16 wavefronts -- every record and CIGAR against the unmodified reference (tests/parity.py).
usage: stress_traceback_emu.py <seconds> <seed>     (end of round 4: 1 386 cases in 4 x 7 minutes, 0 mismatches)
       stress_traceback_emu.py <seconds> <seed> --narrow            the 16-lane anti-diagonal teams: short reads / targets, bands spanning the target, hand-overs
       stress_traceback_emu.py <seconds> <seed> --free-gap-open     the gapO = 0 regime with every CIGAR flag (tests/parity.py free_gap_open_case):
                                                                    no call may fail, every record / CIGAR (mostly `flag 1`) as the reference's"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'complete-striped-smith-waterman-library_amd'))
import ssw_amd
from parity import compare_batch, free_gap_open_case, make_reads, narrow_band_batches
from sswutil import dna_matrix, random_ref, blosum50
lib = ssw_amd.load(os.path.join(ROOT, 'tests', 'emu', 'libssw_emu.so'))
t_end = time.time() + float(sys.argv[1])
seed = int(sys.argv[2]); it = 0; nbad = 0
if "--narrow" in sys.argv:      # the 16-lane traceback teams (k_trace_diag, opt-in): tests/parity.py narrow_band_batches
    os.environ["SSW_GPU_TRACE_DIAG"] = "1"
    ctx = ssw_amd.Context(0, lib)
    rng = np.random.default_rng(seed); nal = 0
    while time.time() < t_end:
        for reads, ref, mat, gapO, gapE, flag in narrow_band_batches(rng, 1):
            it += 1; nal += len(reads)
            Q = ctx.upload(reads); T = ctx.upload([ref])
            res, cig = ctx.align_batch(Q, T, mat, 5, gapO, gapE, flag, 0, 0, -1, 2)
            Q.free(); T.free()
            bad = compare_batch(res, cig, reads, [ref], mat, 5, gapO, gapE, flag, 0, 0, -1, 2)
            if bad:
                nbad += 1; print("MISMATCH batch", it, "seed", seed, bad[:2])
    print("narrow bands: batches", it, "alignments", nal, "mismatching batches", nbad)
    sys.exit(1 if nbad else 0)
if "--free-gap-open" in sys.argv:
    ctx = ssw_amd.Context(0, lib)
    rng = np.random.default_rng(seed); nal = nflag1 = nfail = 0
    while time.time() < t_end:
        reads, ref, mat, n, gapO, gapE, flag, filterd, maskLen = free_gap_open_case(rng); it += 1
        Q = ctx.upload(reads); T = ctx.upload([ref])
        try:
            res, cig = ctx.align_batch(Q, T, mat, n, gapO, gapE, flag, 0, filterd, maskLen, 2)
        except Exception as e:
            nfail += 1; print("FAILED CALL", it, e); continue
        finally:
            Q.free(); T.free()
        bad = compare_batch(res, cig, reads, [ref], mat, n, gapO, gapE, flag, 0, filterd, maskLen, 2)
        nal += len(reads); nflag1 += int((res["flag"] == 1).sum())
        if bad:
            nbad += 1; print("MISMATCH call", it, bad[:2])
    print("free gap open: calls", it, "alignments", nal, "with flag 1", nflag1, "failed calls", nfail, "mismatching calls", nbad)
    sys.exit(1 if nbad or nfail else 0)
while time.time() < t_end:
    rng = np.random.default_rng(seed + it); it += 1
    teams = rng.choice(["", "1", "4", "16"])
    if teams: os.environ["SSW_GPU_TRACE_WAVES"] = teams
    else: os.environ.pop("SSW_GPU_TRACE_WAVES", None)
    ctx = ssw_amd.Context(0, lib)
    L = int(rng.integers(300, 1800))
    ref = random_ref(L + int(rng.integers(50, 1200)), int(rng.integers(1, 1000)), 4)
    reads = []
    for k in range(int(rng.integers(2, 6))):
        kind = rng.integers(0, 4)
        n = int(rng.integers(40, min(len(ref) - 10, 1500)))
        o = int(rng.integers(0, len(ref) - n))
        if kind == 0: r = ref[o:o + n].copy()
        elif kind == 1:
            cut = int(rng.integers(5, max(6, n // 2))); a = int(rng.integers(1, max(2, n - cut - 1)))
            r = np.concatenate([ref[o:o + a], ref[o + a + cut:o + n]])            # deletion
        elif kind == 2:
            ins = int(rng.integers(5, 400)); a = int(rng.integers(1, n - 1))
            r = np.concatenate([ref[o:o + a], rng.integers(0, 4, size=ins, dtype=np.int8), ref[o + a:o + n]])
        else: r = rng.integers(0, 4, size=n, dtype=np.int8)
        reads.append(np.ascontiguousarray(r, dtype=np.int8))
    gO, gE = [(3, 1), (5, 2), (9, 2), (2, 1)][int(rng.integers(0, 4))]
    mat = dna_matrix(int(rng.integers(1, 4)), int(rng.integers(1, 5)))
    Q = ctx.upload(reads); T = ctx.upload([ref])
    res, cig = ctx.align_batch(Q, T, mat, 5, gO, gE, 2, 0, 0, -1, 2)
    bad = compare_batch(res, cig, reads, [ref], mat, 5, gO, gE, 2, 0, 0, -1, 2)
    Q.free(); T.free(); ctx.close()
    if bad:
        nbad += 1; print("MISMATCH seed", seed + it - 1, teams, bad[:2]); break
print("iterations", it, "mismatching", nbad)
