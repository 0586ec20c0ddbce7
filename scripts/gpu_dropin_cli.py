#!/usr/bin/env python3
"""GPU box: the reference's OWN ssw_test (main.c, compiled unmodified against include/ssw.h + libssw.so: oracle/_ref/ssw_test_dropin -- one
ssw_align call per read) beside the batched front end (ssw_test_gpu) on the same files: N reads of 150 bp vs a 1 Mb target, with and
without -c.  Wall times, GCUPS, and that the two print byte-identical stdout.   usage: gpu_dropin_cli.py [reads, default 10000]"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sswutil import random_ref, sample_reads   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
work = "/tmp/dropin_cli"
os.makedirs(work, exist_ok=True)
ref = random_ref(1_000_000, 1, 4)
reads = sample_reads(ref, n, 150, seed=9)
L = "ACGT"
with open(os.path.join(work, "ref.fa"), "w") as f:
    f.write(">ref\n" + "".join(L[c] for c in ref) + "\n")
with open(os.path.join(work, "reads.fq"), "w") as f:
    for i, r in enumerate(reads):
        s = "".join(L[c] for c in r)
        f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
exes = {"reference main.c on libssw.so, one ssw_align per read (ssw_test_dropin)": os.path.join(ROOT, "oracle", "_ref", "ssw_test_dropin"),
        "batched front end (ssw_test_gpu)": os.path.join(ROOT, "complete-striped-smith-waterman-library_amd", "ssw_test_gpu")}
out = {"workload": "%d reads x 150 bp vs a 1 Mb target, default penalties" % n, "cells": float(n) * 150 * 1e6}
for opts in ([], ["-c"]):
    digests = {}
    for name, exe in exes.items():
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([exe] + opts + [os.path.join(work, "ref.fa"), os.path.join(work, "reads.fq")], capture_output=True)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        digests[name] = hashlib.md5(r.stdout).hexdigest()
        out["%s %s" % (name, " ".join(opts) or "(scores only)")] = {"rc": r.returncode, "wall_s": round(best, 3), "ms_per_read": round(best / n * 1e3, 3),
                                                               "gcups": round(out["cells"] / best / 1e9, 1), "stdout_bytes": len(r.stdout)}
    out["stdout identical %s" % (" ".join(opts) or "(scores only)")] = len(set(digests.values())) == 1
print(json.dumps(out))
