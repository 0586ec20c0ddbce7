#!/usr/bin/env python3
"""Differential fuzz of the batch ABI against the unmodified reference (oracle/_ref), on the GPU box (or, with --emu, on the SIMT emulator):
exotic parameters on purpose -- alphabets of 1..128 letters, full-range / all-non-positive / sparse matrices, gapO <= gapE, gapO = 0, gapE = 0,
gaps up to 255, maskLen 0..40, every flag, score_size 0 / 1 / 2, filters, lengths 0..700, now and then up to 2 600 (EMPTY queries and EMPTY targets in 7 % of the slots
each -- round-5 verdict: no fuzzer drew them, which is how an out-of-bounds of the gapO <= gapE path survived), one or several targets (the
database path from four targets on), ALL CALLS ON ONE LONG-LIVED CONTEXT (what a call leaves in the pooled buffers is the next call's
environment).  Every record and CIGAR is compared (tests/parity.py); one call in six that returns CIGARs runs again with mark_mismatch (the device's
'=' / 'X' / soft-clip rewrite and edit distance against the reference's own mark_mismatch() on the raw CIGAR); a call that fails is counted separately from a wrong value.
--budget: every call under a scratch budget drawn anew (1 MiB .. 64 MiB, now and then the default) and batches of up to 300 queries against targets of up to 4 000
residues, so that the chunked paths run: pipelined series of fill launches, traceback rounds cut by the budget, database size classes cut into several launches.
usage: gpu_fuzz.py <seconds> <seed> [--emu | --lib <path>] [--only <k>] [--budget]        -> one JSON line   (--only: just call k of the seed; SSW_FUZZ_TRACE=1: the parameters of every call on stderr before it runs)
(--lib: another build of the emulated library, e.g. the AddressSanitizer one of scripts/asan_emu_fuzz.sh)"""
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
emu = "--emu" in sys.argv
only = int(sys.argv[sys.argv.index("--only") + 1]) if "--only" in sys.argv else 0
small_budget = "--budget" in sys.argv
libpath = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else os.path.join(ROOT, "tests", "emu", "libssw_emu.so") if emu else None
emu = emu or "--lib" in sys.argv
lib = ssw_amd.load(libpath)
ctx = ssw_amd.Context(0, lib)
rng = np.random.default_rng(seed)
t_end = time.time() + secs
calls = aln = failed = wrong = nullrec = flag1 = marked = marked_cigars = 0
import ctypes as C      # noqa: E402
from sswutil import ref_lib      # noqa: E402
R = ref_lib(required=True)
libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]
u32p = C.POINTER(C.c_uint32); i8p = C.POINTER(C.c_int8)
regimes = {"small_budget_calls": 0, "calls_with_pipelined_fill_launches": 0, "multi_strip_queries": 0, "alphabet>32": 0, "gapO>gapE": 0, "gapO<=gapE": 0, "gapO=0": 0, "db_path": 0, "calls_with_empty_query": 0, "calls_with_empty_target": 0}
first = []
while time.time() < t_end:
    kind = rng.random()
    if kind < 0.3:
        n, nc, mat = 5, 4, dna_matrix(int(rng.integers(1, 6)), int(rng.integers(0, 7)))
    elif kind < 0.45:
        n, nc, mat = 24, 20, blosum50()
    else:
        n = int(rng.integers(1, 25)) if rng.random() < 0.85 else int(rng.integers(25, 129)); nc = n - 1 if n > 4 else n      # (above 32 letters: the lane-model kernel + thread traceback)
        style = rng.random()
        if style < 0.3:
            m = rng.integers(-128, 128, size=(n, n))
        elif style < 0.5:
            m = -rng.integers(0, 30, size=(n, n))           # all non-positive
        elif style < 0.75:
            m = np.full((n, n), -int(rng.integers(1, 9))); m[np.arange(n), np.arange(n)] = rng.integers(1, 12, size=n)
        else:
            m = rng.integers(-12, 13, size=(n, n))
        mat = np.ascontiguousarray(m.astype(np.int8).reshape(-1))
    g = rng.random()
    if g < 0.55:
        gapE = int(rng.integers(1, 6)); gapO = gapE + int(rng.integers(1, 12))
    elif g < 0.75:
        gapO = int(rng.integers(0, 8)); gapE = gapO + int(rng.integers(0, 6))
    elif g < 0.85:
        gapO = 0; gapE = int(rng.integers(0, 4))
    else:
        gapO = int(rng.integers(0, 256)); gapE = int(rng.integers(0, 256))
    nt = 1 if rng.random() < 0.7 else int(rng.integers(2, 9))
    refs = [rng.integers(0, nc, size=0 if rng.random() < 0.07 else int(rng.integers(1, 701)), dtype=np.int8) for _ in range(nt)]
    nq = int(rng.integers(1, 10))
    lens = rng.integers(1, 701, size=nq) if rng.random() < 0.3 else rng.integers(1, 160, size=nq)
    if rng.random() < 0.06: lens = rng.integers(700, 2600, size=nq)      # several row strips of the 64-lane strip kernel (768 rows each at 12 per lane), their window passes, team tracebacks
    if small_budget:      # (draws of its own: the default mode's sequence of calls does not change)
        big = rng.random() < 0.6
        if big:
            nq = int(rng.integers(20, 301 if not emu else 61))
            lens = rng.integers(1, 160, size=nq) if rng.random() < 0.7 else rng.integers(100, 420, size=nq)
            if rng.random() < 0.5: lens[:] = int(lens[0])      # one geometry bucket: ONE series of launches
            refs = [rng.integers(0, nc, size=int(rng.integers(500, 4001)) if not emu else int(rng.integers(3000, 6001)), dtype=np.int8) for _ in range(nt if not emu else min(nt, 2))]      # (emulator: fewer queries, so longer targets make a bucket outgrow half a MiB)
        nt = len(refs)
        budget = int(rng.choice([1, 1, 2, 4, 16, 64, 0])) << 20
        lib.ssw_gpu_set_budget(ctx.h, budget)
        regimes["small_budget_calls"] += int(budget != 0)
    lens = np.where(rng.random(nq) < 0.07, 0, lens)
    reads = make_reads(rng, max(refs, key=len), nq, lens, nc, sub=0.06 if nc > 1 else 0.0, frac_random=0.3)      # (one letter: nothing to substitute)
    regimes["calls_with_empty_query"] += int((lens == 0).any()); regimes["calls_with_empty_target"] += int(any(len(r) == 0 for r in refs))
    flag = int(rng.integers(0, 16)); ss = int(rng.choice([2, 2, 2, 0, 1]))
    filters = int(rng.choice([0, 0, 20, 100])); filterd = int(rng.choice([0, 30, 1000])); maskLen = int(rng.choice([-1, -1, 0, 14, 15, 40]))
    regimes["gapO=0" if gapO == 0 else "gapO<=gapE" if gapO <= gapE else "gapO>gapE"] += 1
    if nt >= 4: regimes["db_path"] += 1
    if n > 32: regimes["alphabet>32"] += 1
    if int(np.max(lens)) > 768: regimes["multi_strip_queries"] += 1
    calls += 1; aln += nq * nt
    if os.environ.get("SSW_FUZZ_TRACE"):      # one line per call BEFORE it runs (a crash is then the last line)
        print("call %d: n %d gapO %d gapE %d flag %d ss %d filters %d filterd %d maskLen %d nq %d lens %s nt %d tlens %s" % (calls, n, gapO, gapE, flag, ss, filters, filterd, maskLen, nq, [len(r) for r in reads], nt, [len(r) for r in refs]), file=sys.stderr, flush=True)
    if only and calls != only:      # --only <k>: the inputs of call k of this seed alone (the generator's draws do not depend on results)
        rng.random()
        if calls > only: break
        continue
    Q = ctx.upload(reads); T = ctx.upload(refs)
    try:
        res, cig = ctx.align_batch(Q, T, mat, n, gapO, gapE, flag, filters, filterd, maskLen, ss)
    except Exception as e:      # noqa: BLE001
        failed += 1
        if len(first) < 5: first.append({"failed": str(e)[:200], "n": n, "gapO": gapO, "gapE": gapE, "flag": flag, "ss": ss, "nq": nq, "nt": nt})
        continue
    finally:
        Q.free(); T.free()
    if small_budget: regimes["calls_with_pipelined_fill_launches"] += int(ctx.timing().get("fill_pipelined", 0) > 0)
    bad = compare_batch(res, cig, reads, refs, mat, n, gapO, gapE, flag, filters, filterd, maskLen, ss, max_report=2)
    # one call in six that returns CIGARs again with ssw_gpu_params.mark_mismatch: the device's k_mark against the reference's own mark_mismatch() on the raw CIGAR
    if not bad and rng.random() < 0.17 and int((res["cigarLen"] > 0).sum()) > 0:
        Q = ctx.upload(reads); T = ctx.upload(refs)
        try:
            mres, mcig = ctx.align_batch(Q, T, mat, n, gapO, gapE, flag, filters, filterd, maskLen, ss, mark_mismatch=True)
        except Exception as e:      # noqa: BLE001
            mres = None; bad = ["mark_mismatch call failed: " + str(e)[:160]]
        finally:
            Q.free(); T.free()
        if mres is not None:
            marked += 1
            for qi, rd in enumerate(reads):
                for ti, rf in enumerate(refs):
                    a, b = res[qi, ti], mres[qi, ti]
                    k = int(a["cigarLen"])
                    if k <= 0:
                        if int(b["cigarLen"]) > 0: bad.append("mark_mismatch: a CIGAR where the raw call has none")
                        continue
                    buf = libc.malloc(4 * k)
                    raw = np.ascontiguousarray(cig[int(a["cigar_off"]):int(a["cigar_off"]) + k], dtype=np.uint32)      # (kept in a name: a temporary's buffer may be gone before memmove reads it)
                    C.memmove(buf, raw.ctypes.data, 4 * k)
                    pc = C.cast(buf, u32p); cl = C.c_int32(k)
                    nm = R.mark_mismatch(int(a["ref_begin1"]), int(a["read_begin1"]), int(a["read_end1"]), rf.ctypes.data_as(i8p), rd.ctypes.data_as(i8p), len(rd), C.byref(pc), C.byref(cl))
                    want = [int(pc[x]) for x in range(cl.value)]
                    libc.free(C.cast(pc, C.c_void_p))
                    got = [int(x) for x in mcig[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigarLen"])]] if int(b["cigarLen"]) > 0 else []
                    marked_cigars += 1
                    if got != want or int(b["edit_distance"]) != nm:
                        bad.append("mark_mismatch q%d t%d: expected nm %d %s got nm %d %s" % (qi, ti, nm, want[:8], int(b["edit_distance"]), got[:8]))
    nullrec += int((res["status"] == 1).sum()); flag1 += int((res["flag"] == 1).sum())
    if bad:
        wrong += 1
        if len(first) < 5: first.append({"mismatch": bad[0][:300], "n": n, "gapO": gapO, "gapE": gapE, "flag": flag, "ss": ss, "maskLen": maskLen})
print(json.dumps({"seconds": secs, "seed": seed, "library": "emulator" if emu else "libssw.so on the GPU", "calls": calls, "alignments": aln, "failed_calls": failed,
                  "calls_with_wrong_values": wrong, "records_where_the_reference_returns_NULL": nullrec, "records_with_flag_1": flag1, "calls_repeated_with_mark_mismatch": marked, "marked_cigars_compared": marked_cigars, "regimes": regimes, "first": first}))
