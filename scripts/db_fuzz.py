#!/usr/bin/env python3
"""Differential fuzz of the STREAMED database search (ssw_gpu_search_db, include/ssw_gpu.h) against the batch call and, through it, the unmodified
reference: random alphabets (1..128 letters), matrices, gap regimes (gapO <= gapE and alphabets above 32 letters take the generic path behind the
same entry point), query / entry lengths 0..700 (above 640 residues: generic path), 1..40 entries, chunk sizes 0 (default) / 1 / random, a caller
function that stops the stream at a random chunk, score_size 0 / 1 / 2.  Checked per call: (a) the batch records equal the reference's (tests/parity.py),
(b) the assembled hits equal the batch records field by field (a pair the reference answers with NULL carries ref_end2 = -2), (c) the chunks arrive in
order with the promised sizes, (d) a stop code comes back as the call's return value and nothing later is delivered.
--budget: every call under a scratch budget drawn anew (1 MiB .. 16 MiB, now and then the default), up to 60 queries x 200 entries: size classes that do not fit
their slice of the scratch are cut into several launches, the generic path's batches are chunked.  A call that the library REFUSES for its budget counts separately.
usage: db_fuzz.py <seconds> <seed> [--emu | --lib <path>] [--budget]        -> one JSON line"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "complete-striped-smith-waterman-library_amd"))
import ssw_amd          # noqa: E402
from parity import compare_batch, make_reads   # noqa: E402
from sswutil import blosum50, dna_matrix   # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
libpath = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else os.path.join(ROOT, "tests", "emu", "libssw_emu.so") if "--emu" in sys.argv else None
ctx = ssw_amd.Context(0, ssw_amd.load(libpath))
rng = np.random.default_rng(seed)
t_end = time.time() + secs
calls = pairs = wrong = stops = generic = refused = 0
small_budget = "--budget" in sys.argv
emu_ = "--emu" in sys.argv or "--lib" in sys.argv
first = []
FIELDS = ("score1", "score2", "ref_end1", "read_end1", "ref_end2")
while time.time() < t_end:
    kind = rng.random()
    if kind < 0.3:
        n, nc, mat = 24, 20, blosum50()
    elif kind < 0.5:
        n, nc, mat = 5, 4, dna_matrix(int(rng.integers(1, 6)), int(rng.integers(0, 7)))
    else:
        n = int(rng.integers(1, 25)) if rng.random() < 0.85 else int(rng.integers(25, 129)); nc = n - 1 if n > 4 else n
        mat = np.ascontiguousarray(rng.integers(-12, 13, size=(n, n)).astype(np.int8).reshape(-1))
    if rng.random() < 0.8:
        gapE = int(rng.integers(1, 5)); gapO = gapE + int(rng.integers(1, 12))
    else:
        gapO = int(rng.integers(0, 6)); gapE = gapO + int(rng.integers(0, 4))
    nt = int(rng.integers(1, 41)); nq = int(rng.integers(1, 9))
    tmax = 700 if rng.random() < 0.1 else 300
    if small_budget:      # (draws of its own)
        if rng.random() < 0.6: nt = int(rng.integers(20, 201 if not emu_ else 61)); nq = int(rng.integers(4, 61 if not emu_ else 17))
        ctx.lib.ssw_gpu_set_budget(ctx.h, int(rng.choice([1, 1, 2, 4, 16, 0])) << 20)
    db = [rng.integers(0, nc, size=0 if rng.random() < 0.06 else int(rng.integers(1, tmax + 1)), dtype=np.int8) for _ in range(nt)]
    lens = rng.integers(1, 701, size=nq) if rng.random() < 0.25 else rng.integers(1, 330, size=nq)
    lens = np.where(rng.random(nq) < 0.06, 0, lens)
    qs = make_reads(rng, max(db, key=len), nq, lens, nc, sub=0.15 if nc > 1 else 0.0, frac_random=0.4)
    ss = int(rng.choice([2, 2, 2, 0, 1])); maskLen = int(rng.choice([-1, -1, 15, 40]))
    chunk = int(rng.choice([0, 1, int(rng.integers(1, nt + 1)), int(rng.integers(1, nt + 1))]))
    generic += int(gapO <= gapE or n > 32 or int(lens.max()) > 640)
    calls += 1; pairs += nq * nt
    Q = ctx.upload(qs); T = ctx.upload(db)
    bad = []
    try:
        res, cig = ctx.align_batch(Q, T, mat, n, gapO, gapE, 0, 0, 0, maskLen, ss)
        bad = compare_batch(res, cig, qs, db, mat, n, gapO, gapE, 0, 0, 0, maskLen, ss, max_report=2)
        hits = ctx.search_db(Q, T, mat, n, gapO, gapE, maskLen, ss, chunk)
        null = res["status"] == 1
        for f in FIELDS:
            want = res[f].astype(np.int64)
            if f == "ref_end2": want = np.where(null, -2, want)
            elif null.any(): want = np.where(null, hits[f].astype(np.int64), want)      # (the other fields of a NULL pair are not specified)
            if not (hits[f].astype(np.int64) == want).all(): bad.append("assembled hits differ from the batch records in " + f)
        seen = []
        stop_at = int(rng.integers(0, nt)) if rng.random() < 0.3 else -1
        code = int(rng.choice([7, -5, 1]))

        def on_chunk(tfirst, h):
            seen.append((tfirst, h.shape[1]))
            for f in FIELDS:
                if not (h[f] == hits[f][:, tfirst:tfirst + h.shape[1]]).all(): bad.append("a streamed chunk differs from the assembled hits in " + f)
            return code if stop_at >= 0 and tfirst <= stop_at < tfirst + h.shape[1] else 0
        rc = ctx.search_db(Q, T, mat, n, gapO, gapE, maskLen, ss, chunk, on_chunk)
        step = chunk if chunk > 0 else 2048
        want_seen = [(t0, min(step, nt - t0)) for t0 in range(0, nt, step)]
        if stop_at >= 0:
            stops += 1
            want_seen = [w for w in want_seen if w[0] <= stop_at]
            if rc != code: bad.append("stop code %d came back as %d" % (code, rc))
        elif rc != 0: bad.append("search_db returned %d" % rc)
        if seen != want_seen: bad.append("chunks %s, expected %s" % (seen[:4], want_seen[:4]))
    except Exception as e:      # noqa: BLE001
        if small_budget and "budget" in str(e): refused += 1; bad = []
        else: bad.append("call failed: " + str(e)[:200])
    finally:
        Q.free(); T.free()
    if bad:
        wrong += 1
        if len(first) < 5: first.append({"what": bad[0][:300], "n": n, "gapO": gapO, "gapE": gapE, "ss": ss, "nq": nq, "nt": nt, "chunk": chunk, "qlens": [len(q) for q in qs][:8]})
print(json.dumps({"fuzz": "streamed database search", "seconds": secs, "seed": seed, "library": libpath or "libssw.so on the GPU", "calls": calls, "pairs": pairs, "calls_wrong": wrong,
                  "calls_on_the_generic_path": generic, "small_budgets": small_budget, "calls_refused_for_the_budget": refused, "calls_stopped_by_the_caller": stops, "first": first}))
sys.exit(1 if wrong else 0)
