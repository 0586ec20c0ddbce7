#!/usr/bin/env python3
"""Differential fuzz of the per-GPU work queues (ssw_gpu_pool_*, include/ssw_gpu.h) against ONE context's batch call on the same inputs: 1..4 workers
(on the devices there are: workers share a device when there are fewer), random block sizes (1 read .. the whole set), alphabets 1..128, every gap regime
and flag, mark_mismatch, empty reads and targets, several target sets on one long-lived pool.  The pool's records, CIGAR words and edit distances must be
identical to the single context's whatever worker took which block (the batch call itself is pinned to the reference by gpu_fuzz.py); the statistics must
add up to the blocks and reads handed out.
usage: pool_fuzz.py <seconds> <seed> [--emu | --lib <path>]        -> one JSON line"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "complete-striped-smith-waterman-library_amd"))
import ssw_amd          # noqa: E402
from parity import make_reads   # noqa: E402
from sswutil import blosum50, dna_matrix   # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
libpath = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else os.path.join(ROOT, "tests", "emu", "libssw_emu.so") if "--emu" in sys.argv else None
lib = ssw_amd.load(libpath)
ndev = max(1, lib.ssw_gpu_device_count())
rng = np.random.default_rng(seed)
ctx = ssw_amd.Context(0, lib)
t_end = time.time() + secs
calls = alns = wrong = pools = 0
first = []
FIELDS = ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2", "cigarLen", "edit_distance", "flag", "status")
while time.time() < t_end:
    workers = int(rng.integers(1, 5))
    pool = ssw_amd.Pool([i % ndev for i in range(workers)], lib)
    pools += 1
    try:
        for _ in range(int(rng.integers(1, 4))):      # several target sets on one pool
            kind = rng.random()
            if kind < 0.4:
                n, nc, mat = 5, 4, dna_matrix(int(rng.integers(1, 6)), int(rng.integers(0, 7)))
            elif kind < 0.55:
                n, nc, mat = 24, 20, blosum50()
            else:
                n = int(rng.integers(1, 25)) if rng.random() < 0.85 else int(rng.integers(25, 129)); nc = n - 1 if n > 4 else n
                mat = np.ascontiguousarray(rng.integers(-12, 13, size=(n, n)).astype(np.int8).reshape(-1))
            nt = 1 if rng.random() < 0.6 else int(rng.integers(2, 8))
            refs = [rng.integers(0, nc, size=0 if rng.random() < 0.07 else int(rng.integers(1, 601)), dtype=np.int8) for _ in range(nt)]
            pool.set_targets(refs)
            T = ctx.upload(refs)
            for _ in range(int(rng.integers(1, 4))):  # several read sets per target set
                if rng.random() < 0.75:
                    gapE = int(rng.integers(1, 5)); gapO = gapE + int(rng.integers(1, 12))
                else:
                    gapO = int(rng.integers(0, 6)); gapE = gapO + int(rng.integers(0, 4))
                nq = int(rng.integers(1, 40))
                lens = np.where(rng.random(nq) < 0.07, 0, rng.integers(1, 500 if rng.random() < 0.2 else 160, size=nq))
                reads = make_reads(rng, max(refs, key=len), nq, lens, nc, sub=0.06 if nc > 1 else 0.0, frac_random=0.3)
                flag = int(rng.choice([0, 0, 1, 2, 8, 9, 15])); mm = bool(flag & 7) and rng.random() < 0.3
                ss = int(rng.choice([2, 2, 0, 1])); block = int(rng.choice([0, 1, int(rng.integers(1, nq + 1)), nq]))
                calls += 1; alns += nq * nt
                bad = []
                try:
                    rp, cp = pool.align(reads, mat, n, gapO, gapE, flag, score_size=ss, block=block, mark_mismatch=mm)
                    Q = ctx.upload(reads)
                    try:
                        rs, cs = ctx.align_batch(Q, T, mat, n, gapO, gapE, flag, 0, 0, -1, ss, mark_mismatch=mm)
                    finally:
                        Q.free()
                    for f in FIELDS:
                        if not (rp[f] == rs[f]).all(): bad.append("field " + f)
                    for q in range(nq):
                        for t in range(nt):
                            a, b = rp[q, t], rs[q, t]
                            k = int(a["cigarLen"])
                            if k > 0 and int(b["cigarLen"]) == k:
                                if not (cp[int(a["cigar_off"]):int(a["cigar_off"]) + k] == cs[int(b["cigar_off"]):int(b["cigar_off"]) + k]).all(): bad.append("CIGAR of (%d, %d)" % (q, t))
                            elif k <= 0 and int(a["cigar_off"]) != -1: bad.append("cigar_off of an alignment without CIGAR")
                    st = pool.stats()
                    eff = block if block > 0 else None
                    if sum(s["queries"] for s in st) != nq: bad.append("statistics: %d reads handed out, %d in the set" % (sum(s["queries"] for s in st), nq))
                    if eff and sum(s["blocks"] for s in st) != -(-nq // eff): bad.append("statistics: %d blocks for %d reads in blocks of %d" % (sum(s["blocks"] for s in st), nq, eff))
                except Exception as e:      # noqa: BLE001
                    bad.append("call failed: " + str(e)[:200])
                if bad:
                    wrong += 1
                    if len(first) < 5: first.append({"what": bad[:3], "workers": workers, "n": n, "gapO": gapO, "gapE": gapE, "flag": flag, "ss": ss, "nq": nq, "nt": nt, "block": block, "mm": mm})
            T.free()
    finally:
        pool.close()
ctx.close()
print(json.dumps({"fuzz": "per-GPU work queues (ssw_gpu_pool) vs one context", "seconds": secs, "seed": seed, "library": libpath or "libssw.so on the GPU", "devices": ndev, "pools_opened": pools,
                  "calls": calls, "alignments": alns, "calls_wrong": wrong, "first": first}))
sys.exit(1 if wrong else 0)
