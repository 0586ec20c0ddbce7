#!/usr/bin/env python3
"""GPU box: the traceback rounds of a flagged database search in which every pair survives (SSW_GPU_DEBUG lines)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "complete-striped-smith-waterman-library_amd"))
import ssw_amd, workloads as W
db, qs, mat = W.protein_config(0, queries=2048, db_entries=256)
ctx = ssw_amd.Context(0, ssw_amd.load()); ctx.set_exclusive()
Q = ctx.upload(qs); T = ctx.upload(db)
for env in ({}, {"SSW_GPU_TRACE_WAVE": "1"}):
    os.environ.update(env)
    ctx.align_batch(Q, T, mat, 24, 3, 1, 2, 0, 0, -1, 2)
    os.environ["SSW_GPU_DEBUG"] = "1"
    t0 = time.perf_counter()
    ctx.align_batch(Q, T, mat, 24, 3, 1, 2, 0, 0, -1, 2)
    print("env", env, "seconds", round(time.perf_counter() - t0, 3), ctx.timing()["trace_ms"], file=sys.stderr)
    del os.environ["SSW_GPU_DEBUG"]
    for k in env: del os.environ[k]
