#!/usr/bin/env python3
"""GPU box: what the scratch budget costs and buys on config 2 (100 000 x 150 bp vs 1 Mb, score only): for every budget a FRESH context,
the first call (device allocations inside) and a second call (buffers kept).  usage: gpu_budget_sweep.py [GiB,GiB,...]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "complete-striped-smith-waterman-library_amd"))
import ssw_amd, workloads as W
from sswutil import dna_matrix
lib = ssw_amd.load()
ref, reads, p = W.dna_config(2, 0)
mat = dna_matrix(2, 2)
budgets = [float(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8, 16, 64, 172]
out = []
cells = float(reads.shape[0]) * 150 * len(ref)
for gib in budgets:
    ctx = ssw_amd.Context(0, lib)
    lib.ssw_gpu_set_budget(ctx.h, int(gib * 2 ** 30))
    Q = ctx.upload(list(reads)); T = ctx.upload([ref])
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        ctx.align_batch(Q, T, mat, 5, 3, 1, 0, 0, 0, -1, 2, want_cigar=False)
        ts.append(time.perf_counter() - t0)
    tm = ctx.timing()
    out.append({"budget_gib": gib, "first_call_s": round(ts[0], 3), "second_call_s": round(ts[1], 3), "third_call_s": round(ts[2], 3), "fill_launches": tm["fill_launches"],
                "gcups_steady": round(cells / min(ts[1:]) / 1e9, 1), "gcups_first_call": round(cells / ts[0] / 1e9, 1)})
    print(out[-1], flush=True)
    Q.free(); T.free(); ctx.close()
print(json.dumps({"workload": "config 2: 100 000 x 150 bp vs 1 Mb, score only, fresh context per budget", "runs": out}))
