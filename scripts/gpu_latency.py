#!/usr/bin/env python3
"""GPU box: what ONE ssw_align call of the drop-in ABI costs (include/ssw.h: the reference's callers loop "for each read: ssw_init; for each
target: ssw_align", src/main.c:506, ssw_cpp.cpp:342, pyssw.py:129), per read length / target length / flag.  One JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "complete-striped-smith-waterman-library_amd"))
import ssw_amd          # noqa: E402
from sswutil import SAlign, dna_matrix, i8p, random_ref, sample_reads   # noqa: E402

lib = ssw_amd.load()
lib.ssw_init.argtypes = [i8p, C.c_int32, i8p, C.c_int32, C.c_int8]; lib.ssw_init.restype = C.c_void_p
lib.ssw_align.argtypes = [C.c_void_p, i8p, C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int32, C.c_int32]; lib.ssw_align.restype = C.POINTER(SAlign)
lib.align_destroy.argtypes = [C.POINTER(SAlign)]; lib.init_destroy.argtypes = [C.c_void_p]
mat = dna_matrix(2, 2)
out = {}
for ref_len in (() if os.environ.get('LAT_QUICK') else (10_000, 1_000_000)):
    ref = random_ref(ref_len, 1, 4)
    for rl in (150, 1000):
        reads = sample_reads(ref, 200, rl, seed=5)
        for flag in (0, 2):
            ts = []
            for i, r in enumerate(reads):
                r = np.ascontiguousarray(r)
                t0 = time.perf_counter()
                p = lib.ssw_init(r.ctypes.data_as(i8p), len(r), mat.ctypes.data_as(i8p), 5, 2)
                a = lib.ssw_align(p, ref.ctypes.data_as(i8p), len(ref), 3, 1, flag, 0, 0, rl // 2)
                ts.append(time.perf_counter() - t0)
                assert a and a.contents.score1 > 0
                lib.align_destroy(a); lib.init_destroy(p)
            ts = np.array(ts[20:]) * 1e3      # the first calls create the implicit context and size its buffers
            out["read %d x target %d, flag %d" % (rl, ref_len, flag)] = {"median_ms": round(float(np.median(ts)), 3), "p90_ms": round(float(np.percentile(ts, 90)), 3),
                                                                          "gcups_of_a_caller_loop": round(rl * ref_len / float(np.median(ts)) / 1e6, 1)}
# the same loop from several caller threads (every OS thread gets its own implicit context, include/ssw_gpu.h "Threads"; ctypes releases the GIL
# inside the calls): what an unmodified multi-threaded caller of ssw_align gets.  The threads are PERSISTENT: each warms up (its first call creates
# its context: streams, events, pooled buffers -- milliseconds), all meet at a barrier, then the timed calls.  (Round 4's version of this loop
# started NEW threads for the timed round, i.e. timed 16 context creations against 16 calls each: its "16 threads slower than 4" was the harness.)
import threading
ref = random_ref(1_000_000, 1, 4)
reads = [np.ascontiguousarray(r) for r in sample_reads(ref, 512, 150, seed=6)]
PER_THREAD = 256
for nth in [int(x) for x in os.environ.get('LAT_THREADS', '1,4,8,16').split(',')]:
    for flag in (0, 2):
        bar = threading.Barrier(nth + 1)
        def work(k):
            def one(i):
                r = reads[i % len(reads)]
                p = lib.ssw_init(r.ctypes.data_as(i8p), len(r), mat.ctypes.data_as(i8p), 5, 2)
                a = lib.ssw_align(p, ref.ctypes.data_as(i8p), len(ref), 3, 1, flag, 0, 0, 75)
                lib.align_destroy(a); lib.init_destroy(p)
            for i in range(8):
                one(k + i)
            bar.wait()
            for i in range(PER_THREAD):
                one(k * PER_THREAD + i)
            bar.wait()
        ths = [threading.Thread(target=work, args=(k,)) for k in range(nth)]
        for t in ths: t.start()
        bar.wait(); t0 = time.perf_counter()
        bar.wait(); dt = time.perf_counter() - t0
        for t in ths: t.join()
        ncalls = nth * PER_THREAD
        out["%d caller threads, read 150 x target 1000000, flag %d" % (nth, flag)] = {"calls_per_s": round(ncalls / dt), "ms_per_call_aggregate": round(dt / ncalls * 1e3, 3),
                                                                                      "gcups": round(ncalls * 150 * 1e6 / dt / 1e9, 1), "calls": ncalls}
out["threads_note"] = ("persistent caller threads, 8 warm-up calls each before the timed region (a thread's first call creates its implicit context); "
                       "Python callers: ctypes releases the GIL inside ssw_align, the marshalling around it holds it")
print(json.dumps(out))
