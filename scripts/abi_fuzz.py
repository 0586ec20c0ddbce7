#!/usr/bin/env python3
"""Differential fuzz of the SINGLE-PAIR drop-in ABI (ssw_init / ssw_align / align_destroy / init_destroy of include/ssw.h) against the unmodified
reference (oracle/_ref), in the regimes the round-5 judge drew and the batch fuzzer (gpu_fuzz.py) does not: every flag byte 0..255, score_size
outside 0..2 (-1, 3, 7: the reference has no profile then and ssw_align returns NULL), maskLen < 0 and huge, filterd < 0 and INT_MAX, filters
65535, targets of 0 / 1 / 2 residues, alphabets of 2..128 letters (and a few wider than int8 codes), matrices that saturate both kernels.
Compared: NULL-ness of the return value, every s_align field, every CIGAR word.  An EMPTY read (readLen 0) is undefined behaviour in the
reference (src/ssw.c:264) and legal here: expected = the record ssw_align starts from (ssw.c:870-875).
usage: abi_fuzz.py <seconds> <seed> [--emu | --lib <path>] [--max-calls N]       -> one JSON line"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "complete-striped-smith-waterman-library_amd"))
import ssw_amd          # noqa: E402
from parity import make_reads   # noqa: E402
from sswutil import blosum50, dna_matrix, ref_lib   # noqa: E402

FIELDS = ("nScore", "nScore2", "nRefBeg", "nRefEnd", "nQryBeg", "nQryEnd", "nRefEnd2", "nCigarLen", "nFlag")
REF_FIELDS = ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2", "cigarLen", "flag")
i8p = C.POINTER(C.c_int8)


def draw(rng):
    """-> (read, ref, mat, n, score_size, gapO, gapE, flag, filters, filterd, maskLen)"""
    kind = rng.random()
    if kind < 0.25:
        n, nc, mat = 5, 4, dna_matrix(int(rng.integers(1, 6)), int(rng.integers(0, 7)))
    elif kind < 0.35:
        n, nc, mat = 24, 20, blosum50()
    else:
        n = int(rng.integers(2, 33)) if rng.random() < 0.5 else int(rng.integers(33, 129)) if rng.random() < 0.9 else int(rng.integers(129, 200))
        nc = min(n, 128)
        style = rng.random()
        if style < 0.3:
            m = rng.integers(-128, 128, size=(n, n))
        elif style < 0.45:
            m = rng.integers(90, 128, size=(n, n))            # saturates the 8-bit kernel at once and the 16-bit one on long reads
        elif style < 0.7:
            m = np.full((n, n), -int(rng.integers(1, 9))); m[np.arange(n), np.arange(n)] = rng.integers(1, 12, size=n)
        else:
            m = rng.integers(-12, 13, size=(n, n))
        mat = np.ascontiguousarray(m.astype(np.int8).reshape(-1))
    g = rng.random()
    if g < 0.6:
        gapE = int(rng.integers(1, 6)); gapO = gapE + int(rng.integers(1, 12))
    elif g < 0.85:
        gapO = int(rng.integers(0, 8)); gapE = gapO + int(rng.integers(0, 6))
    else:
        gapO = int(rng.integers(0, 256)); gapE = int(rng.integers(0, 256))
    rl = int(rng.choice([0, 1, 2])) if rng.random() < 0.1 else int(rng.integers(1, 701))
    ref = rng.integers(0, nc, size=rl, dtype=np.int8)
    ql = 0 if rng.random() < 0.03 else int(rng.integers(1, 1400)) if rng.random() < 0.1 else int(rng.integers(1, 200))
    read = make_reads(rng, ref, 1, [ql], nc, frac_random=0.3)[0]
    flag = int(rng.integers(0, 256))
    ss = int(rng.choice([2, 2, 2, 0, 1, -1, 3, 7]))
    filters = int(rng.choice([0, 0, 20, 100, 65535]))
    filterd = int(rng.choice([0, 30, 1000, -1, -1000, 2147483647]))
    maskLen = int(rng.choice([-3, -1, 0, 1, 14, 15, 16, 40, 1000000, len(read) // 2]))
    return read, ref, mat, n, ss, gapO, gapE, flag, filters, filterd, maskLen


def call(L, fields, read, ref, mat, n, ss, gapO, gapE, flag, filters, filterd, maskLen):
    p = L.ssw_init(read.ctypes.data_as(i8p), len(read), mat.ctypes.data_as(i8p), n, ss)
    if not p:
        return "init-null", []
    a = L.ssw_align(p, ref.ctypes.data_as(i8p), len(ref), gapO, gapE, flag, filters, filterd, maskLen)
    if not a:
        L.init_destroy(p)
        return None, []
    s = a.contents
    rec = tuple(int(getattr(s, f)) for f in fields)
    ncig = rec[7]
    cigp = getattr(s, "sCigar", None) if fields is FIELDS else s.cigar
    cig = [int(cigp[i]) for i in range(ncig)] if ncig > 0 and cigp else []
    L.align_destroy(a)
    L.init_destroy(p)
    return rec, cig


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    emu = "--emu" in sys.argv
    libpath = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else os.path.join(ROOT, "tests", "emu", "libssw_emu.so") if emu else None
    max_calls = int(sys.argv[sys.argv.index("--max-calls") + 1]) if "--max-calls" in sys.argv else 1 << 60
    L = ssw_amd.load(libpath)
    R = ref_lib(required=True)
    rng = np.random.default_rng(seed)
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2); os.dup2(devnull, 2)      # both libraries print the reference's warnings to stderr
    t_end = time.time() + secs
    calls = wrong = nulls = wide = empties = 0
    first = []
    try:
        while time.time() < t_end and calls < max_calls:
            args = draw(rng)
            read, ref, mat, n, ss = args[:5]
            calls += 1; wide += n > 32; empties += len(read) == 0
            got = call(L, FIELDS, *args)
            if len(read) == 0 and 0 <= ss <= 2:
                exp = ((0, 0, -1, 0, -1, 0, 0, 0, 0), [])
            elif len(read) == 0:
                exp = (None, [])      # no profile (score_size outside 0..2): "Please call the function ssw_init before ssw_align" -> NULL
            else:
                exp = call(R, REF_FIELDS, *args)
            nulls += exp[0] is None
            if got != exp:
                wrong += 1
                if len(first) < 5:
                    first.append({"expected": str(exp)[:200], "got": str(got)[:200], "n": n, "ss": ss, "readLen": len(read), "refLen": len(ref),
                                  "gapO": args[5], "gapE": args[6], "flag": args[7], "filters": args[8], "filterd": args[9], "maskLen": args[10]})
    finally:
        os.dup2(saved, 2)
    print(json.dumps({"fuzz": "single-pair drop-in ABI", "seconds": secs, "seed": seed, "library": libpath or "libssw.so on the GPU", "calls": calls,
                      "calls_with_wrong_values": wrong, "calls_where_the_reference_returns_NULL": int(nulls), "alphabets_above_32": int(wide),
                      "empty_reads": int(empties), "first": first}))
    return 1 if wrong else 0


if __name__ == "__main__":
    sys.exit(main())
