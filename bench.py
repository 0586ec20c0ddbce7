#!/usr/bin/env python3
"""bench.py -- GCUPS of the Smith-Waterman hot path (BASELINE.json metric) on N MI355X of one node.

`--config` picks one of BASELINE.json's workloads (tests/workloads.py generates them from fixed seeds, in read blocks, so
that rank r of an N-GPU run works on block r of ONE read set):

  2 (default)  100k x 150 bp DNA reads vs a 1 Mb target, 2/-2/3/1, score_size 2, score only            [the metric's config]
  3            150 bp reads vs a 5 Mb target, sharded by read block: 20k-read blocks (the stated subsample of the 1M reads)
  3 --full     the same at its STATED size: all 50 blocks = 1M reads, 7.5e14 cells, divided over the ranks (strong scaling), per-block parity
  4            10k x 10 kb reads vs a 100 kb target, flag 2 (begin positions + CIGAR: banded traceback on the GPU), maskLen 5000
  5            50k protein queries (~300 aa) vs a 10k-entry DB, BLOSUM50 3/1, score only, results streamed (ssw_gpu_search_db)

A "step" is one pass of the whole hot path over one batch whose sequences are already resident in HBM (forward fill,
reduction, read_end1 / begin position, traceback where the flag asks for it, results back on the host).  With N > 1 every
rank works on its own block against a replicated target / DB: no collective on the data path, weak scaling.  GCUPS counts
readLen x refLen of the forward matrix only.

`python bench.py --gpus N` with N > 1 and no launcher around it starts the N ranks itself (torch.distributed.run on 127.0.0.1, one
process per GPU -- what the driver's command line does); under a launcher the ranks are used as they are.

Rank 0 prints ONE JSON line.  Besides the contract fields:
  also            (default run of the metric's config on one GPU) config 2 under pure 8-bit scoring (1/-3/5/2), BASELINE configs 3, 4, 5 and
                  the README's benchmark shape (6) run for a few steps each AFTER the timed region: rate, roofline fraction of their fill
                  kernel, parity against tests/golden/full
  roofline        the BINDING roofline of this integer max-plus recurrence: VALU issue of the dominant fill kernel's recurrence
                  (`frac` on readLen x refLen cells; `peak_note` states the 4-cycle peak and the ISA ideal)
  roofline_hbm    HBM view of the same kernel: algorithmic bytes per launch / mean launch time (HIP events on the library's stream),
                  `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes of THIS kernel source (null when it changed since)
  value_with_h2d  the same batch with the queries uploaded inside the step (PCIe-inclusive; `value` is the resident rate)
  cpu_baseline    the unmodified reference (oracle/_ref, its SSE2 path) on rank 0's usable cores, bounded sample (also on N > 1 lines)
  parity          the GPU results of the timed batch against the reference, EVERY rank on its own read block: the committed full-size
                  fixtures (tests/golden/full, made by scripts/make_expected.py) when the workload is a preset -- all reads of block 0 on
                  one GPU, a seeded 2 000-read sample of block r on rank r of an N-GPU line (`parity.per_rank`) -- else a CPU sample
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "complete-striped-smith-waterman-library_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_LANEOPS = 256 * 4 * 16 * 2.4e9    # packed 16-bit (VOP3P) instructions issue over 4 cycles per wave64: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz
                                            # = 39.3e12 packed lane-instr/s (profiles/round1_valu_rate_probe.txt measures 38.3e12)
FULL = os.path.join(ROOT, "tests", "golden", "full")
CSRC = os.path.join(ROOT, "complete-striped-smith-waterman-library_amd", "csrc")
KERNEL_SRCS = [os.path.join(CSRC, f) for f in ("ssw_kernels.hip", "lanes.h", "ssw_dev.h")]      # everything the device code is made of
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "round6_traffic.json")                         # written by scripts/gpu_profile.sh
# VALU issue, two yardsticks: every instruction of the recurrence charged a 4-cycle slot -- what the kernels' mix actually costs
# (profiles/round3_mix_issue_probe.txt) -- and the ISA ideal in which the three 32-bit adds of a row issue in 2.2 cycles as in a pure stream
CYCLES_PER_PAIR_ROW_4CYCLE = 6.5 * 4.0
CYCLES_PER_PAIR_ROW_ISA_IDEAL = 3.5 * 4.0 + 3.0 * 2.2      # = 20.6


def usable_cores():
    """host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU box reports 256
    logical CPUs but runs the container under cpu.max = 16 CPUs; 256 threads on a 16-CPU quota only thrash)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except Exception:
            pass
    return max(1, n)


def measured_valu_probe(ctx):
    """issue rate of packed 16-bit VALU instructions (lane-op/s), measured on this device right after the timed region.  The probe kernel is a
    diagnostic (include/ssw_gpu_diag.h): the product library does not export it, so it runs from libssw_hooks.so -- the same kernels object with the
    test hooks and diagnostics -- on a context of its own.  0.0 when that library is absent."""
    import ssw_amd
    if hasattr(ctx.lib, "ssw_gpu_valu_probe"):
        return ctx.valu_probe(8192, 4000)
    path = os.path.join(os.path.dirname(ssw_amd.DEFAULT_LIB), "libssw_hooks.so")
    if not os.path.exists(path):
        return 0.0
    hctx = ssw_amd.Context(ctx.device, ssw_amd.load(path))
    try:
        return hctx.valu_probe(8192, 4000)
    finally:
        hctx.close()


def kernel_source_id():
    h = hashlib.sha256()
    for path in KERNEL_SRCS:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_issue(key):
    """SQ_INSTS_VALU / GRBM_GUI_ACTIVE statistics of the dominant fill kernel from the committed PMC passes, for this kernel source only"""
    try:
        with open(TRAFFIC_JSON) as f:
            tj = json.load(f)
        if tj.get("kernel_source_sha16") == kernel_source_id():
            return (tj.get("valu_issue") or {}).get(key)
    except Exception:
        pass
    return None


def measured_traffic(key):
    """HBM bytes per alignment of the dominant kernel from the rocprofv3 PMC passes (scripts/gpu_profile_round3.sh ->
    profiles/round3_traffic.json), only if they were taken on this very kernel source"""
    try:
        with open(TRAFFIC_JSON) as f:
            tj = json.load(f)
        ent = tj.get(key)
        if ent and tj.get("kernel_source_sha16") == kernel_source_id():
            return ent
    except Exception:
        pass
    return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5, 6],
                    help="BASELINE.json workload (default 2: the metric's config); 6 = the README's own benchmark shape (mixed read lengths vs 4.94 Mb)")
    ap.add_argument("--shared-device", action="store_true",
                    help="other processes use these GPUs too: keep the library's conservative scratch budget (default: a rank that has a GPU of its own sizes for the whole HBM)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--reads", type=int, default=None, help="reads (queries) per GPU and step")
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--ref-len", type=int, default=None)
    ap.add_argument("--match", type=int, default=2, help="match score (ssw_test -m)")
    ap.add_argument("--mismatch", type=int, default=2, help="mismatch penalty (ssw_test -x)")
    ap.add_argument("--gap-open", type=int, default=3, help="gap opening penalty (ssw_test -o)")
    ap.add_argument("--gap-extend", type=int, default=1, help="gap extension penalty (ssw_test -e)")
    ap.add_argument("--flag", type=int, default=None, help="ssw_align flag (0 = scores + end positions; 2 = + begin + CIGAR)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads (queries) in the CPU-baseline sample (-1: ~15 s of work; 0: skip)")
    ap.add_argument("--sub", type=float, default=None, help="substitution rate of the synthetic reads")
    ap.add_argument("--indel", type=float, default=None, help="insertion rate = deletion rate of the synthetic reads")
    ap.add_argument("--mask-len", type=int, default=None, help="maskLen (-1: readLen/2 per read, like the reference CLI)")
    ap.add_argument("--db-targets", type=int, default=None, help="config 5: DB entries (default 10000)")
    ap.add_argument("--db-chunk", type=int, default=2048, help="config 5: targets per streamed chunk")
    ap.add_argument("--pool", type=int, default=0,
                    help="> 0: drive the batch through the library's per-GPU work queues (ssw_gpu_pool) with this many workers "
                         "spread over the visible devices, reads on the host (single process)")
    ap.add_argument("--also", default=None,
                    help="comma-separated other BASELINE configs run for 2 steps each AFTER the timed region and attached to the line as "
                         "`also` (default: 3,4,5 when the metric's config runs as stated on one GPU; 'none' to skip)")
    ap.add_argument("--plain", action="store_true",
                    help="profiling runs: no second context under the default scratch budget after the timed region (its launches have another size and "
                         "would mix into the per-kernel statistics of rocprofv3)")
    ap.add_argument("--full", action="store_true",
                    help="config 3 at its STATED size: the 1M reads = 50 read blocks of 20 000, all of them, divided over the ranks (block b on rank b mod N: "
                         "strong scaling -- the 1/2/4/8-GPU lines are the same job); a step is one pass over all blocks")
    ap.add_argument("--blocks", type=int, default=50, help="--full: read blocks of the job (default 50 = 1M reads)")
    ap.add_argument("--lib", default=None, help=argparse.SUPPRESS)   # tests point this at the emulated library
    a = ap.parse_args(argv)
    a.quiet = False
    a.custom = any(getattr(a, k) is not None for k in ("read_len", "ref_len", "sub", "indel")) or (a.match, a.mismatch, a.gap_open, a.gap_extend) != (2, 2, 3, 1)
    if a.full and a.config != 3:
        ap.error("--full is config 3 at its stated size (use --config 3 --full)")
    if a.steps is None:
        a.steps = 1 if a.full else {2: 2, 3: 2, 4: 1, 5: 1, 6: 5}[a.config]
    if a.warmup is None:
        a.warmup = 1
    return a


def own_device(args, world, lib):
    """every rank of this run has a GPU of its own (N <= visible devices) and nobody said otherwise (--shared-device: other processes use
    these GPUs too): the contexts may size their scratch for the whole HBM (ssw_gpu_set_budget_exclusive)"""
    return world <= max(1, lib.ssw_gpu_device_count()) and not args.shared_device and os.environ.get("SSW_BENCH_SHARED_DEVICE", "0") != "1"


def init_dist():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:
        # One process per GPU.  The data path has no collective (reads are sharded, the target is replicated), so the
        # only cross-rank traffic is the timing barrier + MAX: it runs over gloo on CPU tensors, which keeps torch's own
        # HIP runtime out of the processes' data path (libssw.so drives its GPU through its own streams).
        # SSW_BENCH_BACKEND=nccl switches the barrier to RCCL.
        import torch
        import torch.distributed as dist_mod
        backend = os.environ.get("SSW_BENCH_BACKEND", "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist_mod.init_process_group(backend=backend)
        dist = dist_mod
    return world, rank, local_rank, dist


def max_over_ranks(dist, dt):
    if dist is None:
        return dt
    import torch
    dev = "cuda" if os.environ.get("SSW_BENCH_BACKEND", "gloo") == "nccl" else "cpu"
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cigar_hashes(res_col, cig):
    """FNV-1a of every alignment's CIGAR words (what oracle/ref_wrap.c stores in the fixtures)"""
    import workloads as W
    out = np.zeros(len(res_col), dtype=np.uint32)
    for i, g in enumerate(res_col):
        n = int(g["cigarLen"])
        if n > 0:
            o = int(g["cigar_off"])
            out[i] = W.fnv1a_words(cig[o:o + n])
    return out


# ====================================================================================================== DNA configs
def result_fields(g):
    return np.stack([g["score1"], g["score2"], g["ref_begin1"], g["ref_end1"], g["read_begin1"], g["read_end1"], g["ref_end2"],
                     g["cigarLen"], g["flag"]], axis=1).astype(np.int32)


def compare_with_reference_records(g, cig, exp, exp_hash, flag):
    """GPU records `g` (one per read) against reference records made with flag 2 -> (mismatching alignments, what was compared)"""
    got = result_fields(g)
    if flag == 2:
        mism = (got != exp).any(axis=1) | (cigar_hashes(g, cig) != exp_hash)
        return int(mism.sum()), "all s_align fields + FNV-1a of every CIGAR word"
    if flag == 0:   # score-only run against flag-2 records: the five fields the flag does not change; begins must be -1, no CIGAR
        cols = [0, 1, 3, 5, 6]
        mism = (got[:, cols] != exp[:, cols]).any(axis=1) | (got[:, 2] != -1) | (got[:, 4] != -1) | (got[:, 7] != 0) | (got[:, 8] != 0)
        return int(mism.sum()), "score1 score2 ref_end1 read_end1 ref_end2 (+ begins -1, no CIGAR) vs the flag-2 reference records"
    cols = [0, 1, 3, 5, 6]
    return int((got[:, cols] != exp[:, cols]).any(axis=1).sum()), "score1 score2 ref_end1 read_end1 ref_end2"


def gather_objects(dist, world, obj):
    if dist is None:
        return [obj]
    lst = [None] * world
    dist.all_gather_object(lst, obj)
    return lst


def bench_dna(args, world, rank, local_rank, dist):
    import ctypes as C
    import ssw_amd
    import workloads as W
    from sswutil import dna_matrix, random_ref
    lib = ssw_amd.load(args.lib)
    if lib.ssw_gpu_device_count() < 1:
        raise RuntimeError("bench.py: no HIP device visible; libssw.so has no CPU path")
    ndev = lib.ssw_gpu_device_count()
    scoring = (args.match, args.mismatch, args.gap_open, args.gap_extend)

    if args.config == 6:       # the README's benchmark shape: mixed read lengths
        preset = dict(W.MIXED_CONFIG)
        p = dict(preset)
        for k, a in (("reads", args.reads), ("ref_len", args.ref_len), ("flag", args.flag), ("mask_len", args.mask_len)):
            if a is not None:
                p[k] = a
        shape_preset = all(getattr(args, k) is None for k in ("read_len", "sub", "indel")) and all(p[k] == preset[k] for k in ("ref_len", "mask_len"))
        ref, rlist, _ = W.mixed_config(rank, reads=p["reads"], ref_len=p["ref_len"])
        qcodes, off = W.pack(rlist)
        rlen = int(round(float(off[-1]) / len(rlist)))
    else:
        preset = dict(W.DNA_CONFIGS[args.config])
        p = dict(preset)
        for k, a in (("reads", args.reads), ("read_len", args.read_len), ("ref_len", args.ref_len), ("flag", args.flag), ("sub", args.sub),
                     ("indel", args.indel), ("mask_len", args.mask_len)):
            if a is not None:
                p[k] = a
        shape_preset = all(p[k] == preset[k] for k in ("read_len", "ref_len", "sub", "indel", "mask_len"))
        ref = random_ref(p["ref_len"], p["seed_ref"], 4)
        reads2d = W.make_reads_fast(ref, p["reads"], p["read_len"], seed=p["seed_reads"] + rank, sub=p["sub"], ins=p["indel"], dele=p["indel"])
        rlen = p["read_len"]
        qcodes = np.ascontiguousarray(reads2d.reshape(-1))
        off = np.arange(p["reads"] + 1, dtype=np.int64) * rlen
    fixture_ok = shape_preset and p["reads"] == preset["reads"]      # (the generator's stream depends on the read count: the fixtures cover the preset batch only)
    mat = dna_matrix(args.match, args.mismatch)
    nreads, flag = p["reads"], p["flag"]
    want_cigar = (flag & 7) != 0
    name = preset["name"] if fixture_ok else "%d DNA reads of %s bp vs %.2f Mb target (config %d generator)" % (
        nreads, "%d..%d" % (int(np.diff(off).min()), int(np.diff(off).max())) if args.config == 6 else str(rlen), p["ref_len"] / 1e6, args.config)

    pool = None
    if args.pool > 0:
        pool = ssw_amd.Pool([i % ndev for i in range(args.pool)], lib)
        pool.set_targets([ref])
        ctx = None

        def step():
            return pool.align(None, mat, 5, args.gap_open, args.gap_extend, flag, 0, 0, p["mask_len"], 2, want_cigar=want_cigar, packed=(qcodes, off))
    else:
        ctx = ssw_amd.Context(local_rank % ndev, lib)
        budget = ctx.set_exclusive() if own_device(args, world, lib) else int(lib.ssw_gpu_get_budget(ctx.h))

        def upload_reads():
            qh = lib.ssw_gpu_seqs_upload(ctx.h, qcodes.ctypes.data_as(C.POINTER(C.c_int8)), off.ctypes.data_as(C.POINTER(C.c_int64)), nreads)
            if not qh:
                raise RuntimeError(ctx.error())
            Q = ssw_amd.Seqs.__new__(ssw_amd.Seqs); Q.ctx = ctx; Q.count = nreads; Q.h = qh
            return Q
        Q = upload_reads()
        T = ctx.upload([ref])

        def step(Qx=None):
            return ctx.align_batch(Qx or Q, T, mat, 5, args.gap_open, args.gap_extend, flag, 0, 0, p["mask_len"], 2, want_cigar=want_cigar)

    closed = []

    def cleanup():
        if not closed:
            closed.append(1)
            if pool is not None:
                pool.close()
            else:
                Q.free(); T.free(); ctx.close()

    def barrier():
        # align_batch() returns only after its stream is synchronised and the results are on the host, so every rank is
        # idle here; the barrier aligns the ranks' clocks around the timed region.
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    acc = {"fill_ms": 0.0, "fill_launches": 0, "fill_cells": 0, "reduce_ms": 0.0, "locate_ms": 0.0, "trace_ms": 0.0, "total_ms": 0.0, "fill_pipelined": 0}
    res = cig = None
    tm = None
    for _ in range(args.steps):
        res, cig = step()            # returns with results on the host (stream synchronised inside)
        if ctx is not None:
            tm = ctx.timing()
            for k in acc:
                acc[k] += tm[k]
    barrier()
    dt = max_over_ranks(dist, time.perf_counter() - t0)

    cells_per_step = float(off[-1]) * p["ref_len"]
    # every rank's block has (nearly, config 6: the same distribution of) the same number of cells: whole-job cells = sum over the ranks
    cells_all = sum(gather_objects(dist, world, cells_per_step))
    value = cells_all * args.steps / dt / 1e9

    # ---- parity of the timed batch, EVERY rank against the reference's records of ITS read block (tests/golden/full), gathered on rank 0
    from sswutil import ref_lib, oracle_align, _ptr, i8p, i32p, i64p
    R = ref_lib()
    g_all = res[:, 0]
    par = None
    tag = {(2, 2, 3, 1): "", (1, 3, 5, 2): "_u8"}.get(scoring)
    fix = os.path.join(FULL, "config%d%s_block0.npz" % (args.config, tag)) if tag is not None else None
    if fixture_ok and args.config == 6 and scoring in ((2, 2, 3, 1), (1, 3, 5, 2)) and rank == 0 and os.path.exists(os.path.join(FULL, "config6_block0.npz")):
        z = np.load(os.path.join(FULL, "config6_block0.npz"))
        key = "default" if scoring == (2, 2, 3, 1) else "m1x3o5e2"
        bad, what = compare_with_reference_records(g_all, cig, z["fields_" + key], z["cigar_fnv_" + key], flag)
        par = {"sample": int(nreads), "mismatching_alignments": bad, "fields": what,
               "against": "tests/golden/full/config6_block0.npz (%s): unmodified reference (oracle/_ref) on the same seeded reads" % key}
    elif fixture_ok and args.config != 6 and rank == 0 and fix is not None and os.path.exists(fix):
        z = np.load(fix)
        k = min(nreads, len(z["fields"]))
        bad, what = compare_with_reference_records(g_all[:k], cig, z["fields"][:k], z["cigar_fnv"][:k], flag)
        par = {"sample": int(k), "mismatching_alignments": bad, "fields": what,
               "against": "tests/golden/full/%s: unmodified reference (oracle/_ref) on the same seeded reads" % os.path.basename(fix)}
    elif fixture_ok and args.config in (2, 3) and scoring == (2, 2, 3, 1) and os.path.exists(os.path.join(FULL, "config%d_blocks_sample.npz" % args.config)):
        z = np.load(os.path.join(FULL, "config%d_blocks_sample.npz" % args.config))
        if rank < len(z["idx"]):
            idx = z["idx"][rank]
            bad, what = compare_with_reference_records(g_all[idx], cig, z["fields"][rank], z["cigar_fnv"][rank], flag)
            par = {"sample": int(len(idx)), "mismatching_alignments": bad, "fields": what,
                   "against": "tests/golden/full/config%d_blocks_sample.npz[block %d]: unmodified reference on a seeded sample of this rank's read block" % (args.config, rank)}
    cpu = None
    cores = usable_cores()

    def cpu_run(k, threads):
        cres = np.zeros((k, 10), dtype=np.int32)
        secs = R.refwrap_bench(_ptr(qcodes, i8p), _ptr(off, i64p), k, _ptr(ref, i8p), len(ref), _ptr(mat, i8p), 5, args.gap_open, args.gap_extend,
                               flag, 0, 0, p["mask_len"], threads, _ptr(cres, i32p))
        return secs, cres
    if rank == 0 and args.cpu_sample != 0 and R is not None:
        # the CPU baseline: the unmodified reference on this host's usable cores, a bounded sample of rank 0's batch (also on N > 1 lines:
        # it runs after the timed region while the other ranks wait at the last barrier)
        if args.cpu_sample > 0:
            ns = min(args.cpu_sample, nreads)
        else:   # pilot of one read per core, then a sample sized for ~15 s of wall-clock on all cores
            pilot = min(nreads, cores)
            s0, _ = cpu_run(pilot, cores)
            ns = int(min(nreads, max(pilot, pilot / max(s0, 1e-3) * 15.0)))
        secs, cres = cpu_run(ns, cores)
        cpu_gcups = float(off[ns]) * p["ref_len"] / secs / 1e9
        cpu = {"value": round(cpu_gcups, 2), "unit": "GCUPS", "cores": cores, "kind": "reference",
               "per_core": round(cpu_gcups / cores, 2), "host_logical_cpus": os.cpu_count(),
               "cores_note": "threads = CPUs this container may use (affinity capped by the cgroup cpu.max quota)",
               "sample": "first %d reads of rank 0's batch vs the same target, ssw_init(...,2)+ssw_align through the C API, "
                         "reference ssw.c built -O2 (oracle/_ref), one thread per core, %.1f s" % (ns, secs)}
        if par is None:
            par = {"sample": ns, "mismatching_alignments": int((result_fields(g_all[:ns]) != cres[:, :9]).any(axis=1).sum()),
                   "fields": "score1 score2 ref/read begin/end ref_end2 cigarLen flag", "against": "the CPU-baseline run of this process"}
    if par is None and R is not None and args.cpu_sample != 0:      # a rank without a fixture for its block: a small sample through the reference
        ns = min(nreads, 32)
        _, cres = cpu_run(ns, 2)
        par = {"sample": ns, "mismatching_alignments": int((result_fields(g_all[:ns]) != cres[:, :9]).any(axis=1).sum()),
               "fields": "score1 score2 ref/read begin/end ref_end2 cigarLen flag", "against": "the unmodified reference (oracle/_ref) run by this rank on its first reads"}
    if par is None and args.cpu_sample != 0:                      # no reference build on this host: the scalar port (oracle/), a couple of reads
        ns = min(nreads, args.cpu_sample if args.cpu_sample > 0 else 2)
        t1 = time.perf_counter()
        mism = 0
        for i in range(ns):
            rd = qcodes[off[i]:off[i + 1]]
            d, _ = oracle_align(rd, mat, 5, ref, args.gap_open, args.gap_extend, flag, 0, 0, p["mask_len"] if p["mask_len"] >= 0 else len(rd) // 2, 2, 0)
            g = res[i, 0]
            mism += any(int(g[k]) != d[k] for k in ("score1", "score2", "ref_end1", "read_end1", "ref_end2"))
        secs = time.perf_counter() - t1
        if rank == 0:
            cpu = {"value": round(float(off[ns]) * p["ref_len"] / secs / 1e9, 3), "unit": "GCUPS", "cores": 1,
                   "kind": "port", "sample": "%d reads, scalar lane-model oracle (oracle/_ref not shipped)" % ns}
        par = {"sample": ns, "mismatching_alignments": mism, "fields": "score1 score2 ref_end1 read_end1 ref_end2", "against": "the scalar port (oracle/)"}
    if par is not None:
        par["rank"] = rank
    per_rank = gather_objects(dist, world, par)

    out = None
    if rank == 0:
        out = {"metric": "GCUPS", "value": round(value, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int16x2 (packed; reference u8/int16 semantics)", "data": "synthetic",
               "config": {"workload": "%s; %d/-%d/%d/%d, score_size 2, flag %d" % (name, args.match, args.mismatch, args.gap_open, args.gap_extend, flag),
                          "baseline_config": args.config, "reads_per_gpu": nreads, "read_len": rlen, "ref_len": p["ref_len"],
                          "sharding": "read block r on rank r, target replicated, no collective"},
               "value_note": "whole-job rate with the sequences resident in HBM when the timed region starts (the bench contract); "
                             "value_with_h2d re-uploads the queries inside the step (SURVEY 8d's definition)"}
        if args.config == 6:
            out["config"]["readme_context"] = ("the reference's README.md:62-74 quotes ~880 s (defaults) / ~460 s (-m1 -x3 -o5 -e2) of CPU time, one thread, for its "
                                               "1000 Ion Torrent reads vs E. coli 536 (~1.1 / ~2.1 GCUPS): same shape, other data and hardware -- context, not vs_baseline")
        if pool is None:
            out["config"]["scratch_budget_gib"] = round(budget / 2.0 ** 30, 1)
        if pool is not None:
            out["config"]["pool_workers"] = args.pool
            out["config"]["note"] = "in-library per-GPU work queues (ssw_gpu_pool): reads on the host, blocks uploaded by the workers inside the step"
            out["pool_stats"] = pool.stats()
        if tm is not None:
            if tm["fill_ops_per_row"] == 6.5:
                out["dtype"] = "int16x2 in a column frame (value + phi(column); 32-bit adds on the packed pair, three-input maxima on the bit patterns); reference u8/int16 semantics"
            out["mix"] = {"word_rules": int(tm["n_word"]), "byte_rules": int(tm["n_byte"])}
            out["phases_ms_per_step"] = {"fill": round(acc["fill_ms"] / args.steps, 3), "locate": round(acc["locate_ms"] / args.steps, 3),
                                         "trace": round(acc["trace_ms"] / args.steps, 3), "reduce_and_copies": round(acc["reduce_ms"] / args.steps, 3)}
            launch_ms = acc["fill_ms"] / max(1, acc["fill_launches"])
            # ---- the binding roofline: VALU issue of the fill kernel's recurrence
            ops = tm["fill_ops_per_row"]
            fill_s = acc["fill_ms"] * 1e-3
            achieved_valu = acc["fill_cells"] * ops / 2.0 / fill_s if fill_s > 0 else 0.0        # every evaluated cell: padding rows, halo columns
            real = cells_per_step * args.steps * ops / 2.0 / fill_s if fill_s > 0 else 0.0       # readLen x refLen only
            probe = measured_valu_probe(ctx) if args.lib is None else 0.0      # (skipped on the test emulator)
            ideal = (CYCLES_PER_PAIR_ROW_ISA_IDEAL / CYCLES_PER_PAIR_ROW_4CYCLE) if ops == 6.5 else 1.0
            out["roofline"] = {"bound": "valu-issue", "kernel": tm["fill_kernel"], "achieved": round(real / 1e12, 3), "peak": round(VALU_PEAK_LANEOPS / 1e12, 2),
                               "unit": "T lane-op/s", "frac": round(real / VALU_PEAK_LANEOPS, 4),
                               "frac_with_padding_and_halo": round(achieved_valu / VALU_PEAK_LANEOPS, 4),
                               "frac_of_isa_ideal": round(real / VALU_PEAK_LANEOPS * ideal, 4),
                               "measured_peak_probe": round(probe / 1e12, 2), "ops_per_pair_row": ops,
                               "launch_ms": round(launch_ms, 3), "launches": int(acc["fill_launches"]),
                               "launches_pipelined": int(acc.get("fill_pipelined", 0)),
                               "launch_note": ("the %d fill launches of a step overlap (pipelined series: main stream / a second stream alternately, each with half of the scratch, two launches in flight): "
                                               "launch_ms = the HIP-event bracket around the series / launches; a rocprofv3 trace shows LONGER kernel durations (two launches "
                                               "share the compute units) -- compare with union_ns of profiles/*_kernel_stats.csv"
                                               % (acc["fill_launches"] // max(1, args.steps))) if acc.get("fill_pipelined", 0) else "serial launches: launch_ms is the kernel's average duration",
                               "fill_gcups_padded": round(acc["fill_cells"] / fill_s / 1e9, 1) if fill_s > 0 else 0.0,
                               "peak_note": "integer max-plus recurrence: neither MFMA nor HBM binds, the issue of vector instructions does.  `achieved` = readLen x refLen cells "
                                            "x %.1f recurrence instructions per row of a query pair / 2 / fill time; `peak` = one wave64 instruction per 4 cycles and SIMD = 256 CU x 4 SIMD "
                                            "x 16 lanes x 2.4 GHz = 39.3 T (every VOP3P, and every other vector instruction in this mix: profiles/round3_mix_issue_probe.txt).  "
                                            "`frac_of_isa_ideal`: against the ISA ideal of %.1f cycles per pair-row (the three 32-bit adds at the 2.2 cycles they take in a pure "
                                            "stream) instead of 26 -- not reachable by reordering (DESIGN.md 8b), stated for completeness" % (ops, CYCLES_PER_PAIR_ROW_ISA_IDEAL),
                               "counters": measured_issue("config%d" % args.config) if fixture_ok else None}
            # ---- the HBM view (not binding): algorithmic bytes of one fill launch (SURVEY 8d / DESIGN.md): per alignment refLen target codes read,
            # readLen + n^2 query/matrix bytes, 4*refLen column-maximum bytes written (2 rule sets x u16), 40 B result; long queries add the
            # boundary records between strips (16 B per column and pair, written once and read once)
            aln_per_launch = nreads * args.steps / max(1, acc["fill_launches"])
            bytes_per_aln = p["ref_len"] + rlen + 25 + 40 + 4 * p["ref_len"] + p["ref_len"] // 4      # (+ the 16-column group maxima: two u32 streams per pair / 16)
            survey_bytes = p["ref_len"] + rlen + 25 + 40 + 4 * p["ref_len"]      # SURVEY 8d's own formula: target + read + matrix + record + maxColumn written and re-read
            if tm["fill_strips"] > 1:
                bytes_per_aln += 16 * p["ref_len"] * (tm["fill_strips"] - 1)
            achieved = aln_per_launch * bytes_per_aln / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
            tr = measured_traffic("config%d" % args.config) if fixture_ok else None
            traffic = round(tr["hbm_bytes_per_alignment"] * aln_per_launch / (launch_ms * 1e-3) / 1e9, 2) if tr and launch_ms > 0 else None
            out["roofline_hbm"] = {"bound": "hbm", "kernel": tm["fill_kernel"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                                   "traffic_note": ("GB/s from rocprofv3 FETCH_SIZE + WRITE_SIZE of this kernel, PMC passes of this kernel source "
                                                    "(%s, source %s)" % (os.path.relpath(TRAFFIC_JSON, ROOT), kernel_source_id())) if traffic is not None else
                                                   "no PMC pass of this kernel source committed (scripts/gpu_profile.sh writes %s)" % os.path.relpath(TRAFFIC_JSON, ROOT),
                                   "launch_ms": round(launch_ms, 3), "launches": int(acc["fill_launches"]),
                                   "algorithmic_bytes_per_alignment": int(bytes_per_aln),
                                   "survey_8d_bytes_per_alignment": int(survey_bytes),
                                   "design_over_survey_8d": round(bytes_per_aln / float(survey_bytes), 2),
                                   "traffic_over_survey_8d": round(tr["hbm_bytes_per_alignment"] / float(survey_bytes), 2) if tr else None,
                                   "bytes_note": "algorithmic_bytes_per_alignment is what THIS design moves by construction (long queries: + the strip boundary records, 16 B per "
                                                 "column and strip boundary, written once and read once); survey_8d_bytes_per_alignment is SURVEY 8d's formula for an ideal single-pass kernel",
                                   "note": "HBM is not the binding resource of this path (see roofline)"}
            out["roofline"]["traffic"] = traffic      # (HBM GB/s of the same kernel from the PMC passes: the contract's key; details in roofline_hbm)
            out["roofline"]["traffic_unit"] = "GB/s of HBM traffic (FETCH_SIZE + WRITE_SIZE), see roofline_hbm"
            # the same batch under the library's DEFAULT scratch budget (min(64 GiB, half of the free HBM): what a caller gets who does not say
            # "this device is mine"), on a second context, two steps after one warm-up -- next to the exclusive budget the timed region used
            if world == 1 and not args.quiet and not args.plain and own_device(args, world, lib) and args.lib is None:
                try:
                    ctx2 = ssw_amd.Context(local_rank % ndev, lib)
                    b2 = int(lib.ssw_gpu_get_budget(ctx2.h))
                    qh2 = lib.ssw_gpu_seqs_upload(ctx2.h, qcodes.ctypes.data_as(C.POINTER(C.c_int8)), off.ctypes.data_as(C.POINTER(C.c_int64)), nreads)
                    Qd = ssw_amd.Seqs.__new__(ssw_amd.Seqs); Qd.ctx = ctx2; Qd.count = nreads; Qd.h = qh2
                    Td = ctx2.upload([ref])
                    run2 = lambda: ctx2.align_batch(Qd, Td, mat, 5, args.gap_open, args.gap_extend, flag, 0, 0, p["mask_len"], 2, want_cigar=want_cigar)
                    run2()
                    t2 = time.perf_counter(); run2(); run2(); d2 = (time.perf_counter() - t2) / 2
                    out["value_default_budget"] = {"value": round(cells_per_step / d2 / 1e9, 2), "unit": "GCUPS", "scratch_budget_gib": round(b2 / 2.0 ** 30, 1),
                                                   "fill_launches_per_step": int(ctx2.timing()["fill_launches"]),
                                                   "note": "same batch, a second context with the library's default budget; `value` ran under ssw_gpu_set_budget_exclusive "
                                                           "(%.1f GiB); budget sweeps: profiles/round6_pipeline_parts.txt, round5_budget_sweep_config2.txt" % (budget / 2.0 ** 30)}
                    Qd.free(); Td.free(); ctx2.close()
                except Exception as e:      # the metric's line must survive a failure here
                    out["value_default_budget"] = {"error": "%s: %s" % (type(e).__name__, e)}
            # PCIe-inclusive rate: one more step with the reads uploaded (and freed) inside it
            if world == 1 and not args.quiet:
                t1 = time.perf_counter()
                Q2 = upload_reads()
                step(Q2)
                Q2.free()
                out["value_with_h2d"] = round(cells_per_step / (time.perf_counter() - t1) / 1e9, 2)
        got_par = [x for x in per_rank if x]
        if got_par:
            out["parity"] = dict(got_par[0])
            out["parity"].pop("rank", None)
            if world > 1:
                out["parity"] = {"sample": int(sum(x["sample"] for x in got_par)), "mismatching_alignments": int(sum(x["mismatching_alignments"] for x in got_par)),
                                 "ranks_checked": len(got_par), "per_rank": got_par}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if not args.quiet:
            cleanup()        # this config's device buffers go first: the `also` runs get the whole device, like a run of their own
            attach_also(args, out, world)
            print(json.dumps(out))
            sys.stdout.flush()
    dump = os.environ.get("SSW_BENCH_DUMP")
    if dump:   # tests: keep every rank's shard and results for an independent check
        np.savez(os.path.join(dump, "rank%d.npz" % rank), reads=qcodes.reshape(nreads, -1) if args.config != 6 else qcodes, ref=ref, res=res)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    cleanup()
    return out, res


# ====================================================================================================== config 3 at its stated size
def full_fixture(block):
    """-> (read indices, reference records, CIGAR hashes) of the committed sample of read block `block` of config 3, or None:
    blocks 0..7 carry 2 000 sampled reads each (config3_blocks_sample.npz), blocks 8..49 500 each (config3_blocks_sample_8_49.npz)"""
    for name in ("config3_blocks_sample.npz", "config3_blocks_sample_8_49.npz"):
        path = os.path.join(FULL, name)
        if not os.path.exists(path):
            continue
        z = np.load(path)
        first = int(z["meta"][3]) if len(z["meta"]) > 3 else 0
        if first <= block < first + len(z["idx"]):
            return z["idx"][block - first], z["fields"][block - first], z["cigar_fnv"][block - first], name
    return None


def bench_config3_full(args, world, rank, local_rank, dist):
    """BASELINE config 3 as stated: 1M x 150 bp reads (50 seeded blocks of 20 000) against the 5 Mb target, score only.  The read blocks are
    divided over the ranks (block b on rank b mod N), the target is replicated, there is no collective: total work is fixed, so the lines
    of 1, 2, 4 and 8 GPUs are the same job (strong scaling).  A step = one pass over every block of the rank through ONE context.
    `value`: the blocks resident in HBM when the timed region starts (the bench contract).  `value_with_h2d` (one GPU): the reads start on
    the host and a feeder thread uploads block i + 1 while block i is being aligned (the streamed form of SURVEY 8d)."""
    import ctypes as C
    import threading
    import ssw_amd
    import workloads as W
    from sswutil import dna_matrix, random_ref, ref_lib, _ptr, i8p, i32p, i64p
    lib = ssw_amd.load(args.lib)
    if lib.ssw_gpu_device_count() < 1:
        raise RuntimeError("bench.py: no HIP device visible; libssw.so has no CPU path")
    ndev = lib.ssw_gpu_device_count()
    preset = dict(W.DNA_CONFIGS[3])
    p = dict(preset)
    for k, a in (("reads", args.reads), ("read_len", args.read_len), ("ref_len", args.ref_len)):
        if a is not None:
            p[k] = a
    stated = all(p[k] == preset[k] for k in ("reads", "read_len", "ref_len")) and not args.custom and args.blocks == 50
    nblocks, nreads, rlen = args.blocks, p["reads"], p["read_len"]
    mine = [b for b in range(nblocks) if b % world == rank]
    ref = random_ref(p["ref_len"], p["seed_ref"], 4)
    mat = dna_matrix(args.match, args.mismatch)
    off = np.arange(nreads + 1, dtype=np.int64) * rlen
    host = {b: np.ascontiguousarray(W.make_reads_fast(ref, nreads, rlen, seed=p["seed_reads"] + b, sub=p["sub"], ins=p["indel"], dele=p["indel"]).reshape(-1))
            for b in mine}
    ctx = ssw_amd.Context(local_rank % ndev, lib)
    budget = ctx.set_exclusive() if own_device(args, world, lib) else int(lib.ssw_gpu_get_budget(ctx.h))
    T = ctx.upload([ref])

    def upload(b):
        qh = lib.ssw_gpu_seqs_upload(ctx.h, host[b].ctypes.data_as(C.POINTER(C.c_int8)), off.ctypes.data_as(C.POINTER(C.c_int64)), nreads)
        if not qh:
            raise RuntimeError(ctx.error())
        Q = ssw_amd.Seqs.__new__(ssw_amd.Seqs); Q.ctx = ctx; Q.count = nreads; Q.h = qh
        return Q

    def align(Q):
        return ctx.align_batch(Q, T, mat, 5, args.gap_open, args.gap_extend, 0, 0, 0, p["mask_len"], 2, want_cigar=False)[0]

    resident = {b: upload(b) for b in mine}
    for _ in range(args.warmup):      # one block, untimed: allocations, code objects
        if mine:
            align(resident[mine[0]])
    if dist is not None:
        dist.barrier()
    acc = {"fill_ms": 0.0, "fill_launches": 0, "fill_cells": 0, "reduce_ms": 0.0, "locate_ms": 0.0, "trace_ms": 0.0, "total_ms": 0.0, "n_word": 0, "n_byte": 0, "fill_pipelined": 0}
    results = {}
    tm = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for b in mine:
            results[b] = align(resident[b])      # (returns with the records on the host)
            tm = ctx.timing()
            for k in acc:
                acc[k] += tm[k]
    if dist is not None:
        dist.barrier()
    dt = max_over_ranks(dist, time.perf_counter() - t0)
    for Q in resident.values():
        Q.free()
    cells_job = float(nblocks) * nreads * rlen * p["ref_len"]
    value = cells_job * args.steps / dt / 1e9

    # ---- the streamed form (one GPU): reads on the host, block i + 1 uploaded by a feeder thread while block i is aligned
    streamed = None
    if world == 1 and not args.quiet and mine:
        nxt = {}

        def feed(b):
            nxt[b] = upload(b)
        t1 = time.perf_counter()
        feed(mine[0])
        for i, b in enumerate(mine):
            th = None
            if i + 1 < len(mine):
                th = threading.Thread(target=feed, args=(mine[i + 1],)); th.start()
            r = align(nxt[b])
            if th is not None:
                th.join()
            nxt.pop(b).free()
            if not (result_fields(r[:, 0]) == result_fields(results[b][:, 0])).all():
                raise RuntimeError("bench.py: the streamed pass gave other records than the resident pass (block %d)" % b)
        streamed = cells_job / (time.perf_counter() - t1) / 1e9

    # ---- parity: every block of this rank against the committed reference sample of THAT block
    checked = []
    for b in mine:
        fx = full_fixture(b) if stated else None
        if fx is None:
            continue
        idx, exp, exph, name = fx
        bad, what = compare_with_reference_records(results[b][idx, 0], None, exp, exph, 0)
        checked.append({"block": b, "sample": int(len(idx)), "mismatching_alignments": bad, "against": name})
    R = ref_lib()
    cpu = None
    if not stated and R is not None and args.cpu_sample != 0 and mine:      # a shape without fixtures (tests): the first reads of every block through the reference
        for b in mine:
            ns = min(nreads, 4)
            cres = np.zeros((ns, 10), dtype=np.int32)
            R.refwrap_bench(_ptr(host[b], i8p), _ptr(off, i64p), ns, _ptr(ref, i8p), len(ref), _ptr(mat, i8p), 5, args.gap_open, args.gap_extend,
                            0, 0, 0, p["mask_len"], 1, _ptr(cres, i32p))
            bad = int((result_fields(results[b][:ns, 0]) != cres[:, :9]).any(axis=1).sum())
            checked.append({"block": b, "sample": ns, "mismatching_alignments": bad, "against": "the unmodified reference (oracle/_ref) run by this rank"})
    if rank == 0 and R is not None and args.cpu_sample != 0 and mine:
        cores = usable_cores()
        b0 = mine[0]

        def cpu_run(k, threads):
            cres = np.zeros((k, 10), dtype=np.int32)
            return R.refwrap_bench(_ptr(host[b0], i8p), _ptr(off, i64p), k, _ptr(ref, i8p), len(ref), _ptr(mat, i8p), 5, args.gap_open, args.gap_extend,
                                   0, 0, 0, p["mask_len"], threads, _ptr(cres, i32p))
        if args.cpu_sample > 0:
            ns = min(args.cpu_sample, nreads)
        else:
            pilot = min(nreads, cores)
            ns = int(min(nreads, max(pilot, pilot / max(cpu_run(pilot, cores), 1e-3) * 15.0)))
        secs = cpu_run(ns, cores)
        g = float(ns) * rlen * p["ref_len"] / secs / 1e9
        cpu = {"value": round(g, 2), "unit": "GCUPS", "cores": cores, "kind": "reference", "per_core": round(g / cores, 2),
               "sample": "first %d reads of block %d vs the same target, ssw_init(...,2)+ssw_align through the C API, reference ssw.c built -O2 "
                         "(oracle/_ref), one thread per core, %.1f s; the whole job at this rate: %.1f h" % (ns, b0, secs, cells_job / (g * 1e9) / 3600.0)}
    all_checked = [x for lst in gather_objects(dist, world, checked) for x in lst]
    all_acc = gather_objects(dist, world, acc)
    out = None
    if rank == 0:
        fill_s = sum(a["fill_ms"] for a in all_acc) * 1e-3 / world      # mean over the ranks (they work side by side)
        launches = sum(a["fill_launches"] for a in all_acc)
        ops = tm["fill_ops_per_row"] if tm else 6.5
        real = cells_job * args.steps * ops / 2.0 / (fill_s * world) if fill_s > 0 else 0.0      # per GPU
        launch_ms = sum(a["fill_ms"] for a in all_acc) / max(1, launches)
        out = {"metric": "GCUPS", "value": round(value, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "int16x2 in a column frame (value + phi(column); 32-bit adds on the packed pair, three-input maxima on the bit patterns); reference u8/int16 semantics",
               "data": "synthetic",
               "config": {"workload": "%s: %d x %d bp reads (%d seeded blocks of %d) vs %.3g Mb target; %d/-%d/%d/%d, score_size 2, flag 0"
                                      % ("BASELINE config 3 at its stated size" if stated else "config-3 generator, block job", nblocks * nreads, rlen, nblocks, nreads, p["ref_len"] / 1e6, args.match, args.mismatch, args.gap_open, args.gap_extend),
                          "baseline_config": 3, "reads": nblocks * nreads, "read_blocks": nblocks, "reads_per_block": nreads, "read_len": rlen, "ref_len": p["ref_len"],
                          "cells": cells_job, "blocks_per_gpu": [len([b for b in range(nblocks) if b % world == r]) for r in range(world)],
                          "scratch_budget_gib": round(budget / 2.0 ** 30, 1),
                          "sharding": "read block b on rank b mod N, target replicated, no collective; a step = one pass over all blocks (total work fixed)"},
               "value_note": "whole-job rate with every read block resident in HBM when the timed region starts (the bench contract); value_with_h2d: reads on the "
                             "host, block i+1 uploaded by a second thread on the context's upload stream while block i is aligned",
               "mix": {"word_rules": int(sum(a["n_word"] for a in all_acc)), "byte_rules": int(sum(a["n_byte"] for a in all_acc))},
               "phases_ms_per_step": {k2: round(sum(a[k1] for a in all_acc) / world / args.steps, 3) for k1, k2 in
                                      (("fill_ms", "fill"), ("locate_ms", "locate"), ("trace_ms", "trace"), ("reduce_ms", "reduce_and_copies"))},
               "roofline": {"bound": "valu-issue", "kernel": tm["fill_kernel"] if tm else None, "achieved": round(real / 1e12, 3), "peak": round(VALU_PEAK_LANEOPS / 1e12, 2),
                            "unit": "T lane-op/s per GPU", "frac": round(real / VALU_PEAK_LANEOPS, 4),
                            "frac_of_isa_ideal": round(real / VALU_PEAK_LANEOPS * (CYCLES_PER_PAIR_ROW_ISA_IDEAL / CYCLES_PER_PAIR_ROW_4CYCLE if ops == 6.5 else 1.0), 4),
                            "ops_per_pair_row": ops, "launch_ms": round(launch_ms, 3), "launches": int(launches), "traffic": None,
                            "peak_note": "VALU issue of the fill kernel's recurrence on readLen x refLen cells, as on the default line (DESIGN.md 4); traffic: see the "
                                         "default line's roofline_hbm (same kernel, PMC passes in profiles/)"}}
        if streamed is not None:
            out["value_with_h2d"] = round(streamed, 2)
        if all_checked:
            out["parity"] = {"sample": int(sum(x["sample"] for x in all_checked)), "mismatching_alignments": int(sum(x["mismatching_alignments"] for x in all_checked)),
                             "blocks_checked": len(all_checked), "blocks": nblocks,
                             "fields": "score1 score2 ref_end1 read_end1 ref_end2 (+ begins -1, no CIGAR) vs the flag-2 reference records" if stated else "all record fields",
                             "against": "tests/golden/full/config3_blocks_sample.npz (blocks 0..7, 2 000 sampled reads each) and config3_blocks_sample_8_49.npz "
                                        "(blocks 8..49, 500 each): unmodified reference (oracle/_ref) on the same seeded reads" if stated else
                                        "the unmodified reference (oracle/_ref) on the first reads of every block",
                             "per_block": all_checked if not stated else [x for x in all_checked if x["mismatching_alignments"]]}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if not args.quiet:
            print(json.dumps(out))
            sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    T.free(); ctx.close()
    return out, results


# ====================================================================================================== config 5
def bench_db(args, world, rank, local_rank, dist):
    """BASELINE config 5: every protein query against every DB entry, BLOSUM50, gaps 3/1, score only; the 5e8 records of the
    full size are streamed back in target chunks (ssw_gpu_search_db) and folded into order-independent checksums as they pass."""
    import ssw_amd
    import workloads as W
    from sswutil import ref_lib, _ptr, i8p, i32p, i64p
    lib = ssw_amd.load(args.lib)
    if lib.ssw_gpu_device_count() < 1:
        raise RuntimeError("bench.py: no HIP device visible; libssw.so has no CPU path")
    ctx = ssw_amd.Context(local_rank % max(1, lib.ssw_gpu_device_count()), lib)
    if own_device(args, world, lib):
        ctx.set_exclusive()
    nq = args.reads if args.reads is not None else 50_000
    nt = args.db_targets if args.db_targets is not None else 10_000
    db, qs, mat = W.protein_config(rank, queries=nq, db_entries=nt)
    is_preset = nt == 10_000 and nq == 50_000       # (the generators' streams depend on the counts: the fixtures cover the stated size only)
    Q = ctx.upload(qs); T = ctx.upload(db)
    keep = min(nq, 2048)                         # rows kept for the per-query parity check
    blk = 2048
    nblk = -(-nq // blk)
    state = {}

    def fresh():
        state["kept"] = np.zeros((keep, nt), dtype=ssw_amd.HIT_DTYPE)
        state["sums"] = [(0, 0, 0)] * nblk

    def on_chunk_timed(tfirst, hits):
        # the records are on the host (page-locked buffer of the library) when this is called: the timed steps only keep the rows
        # of the first queries; folding all 5e8 records into checksums is verification work and runs in one extra, untimed step
        state["kept"][:, tfirst:tfirst + hits.shape[1]] = hits[:keep]
        return 0

    def on_chunk_verify(tfirst, hits):
        state["kept"][:, tfirst:tfirst + hits.shape[1]] = hits[:keep]
        w = np.ascontiguousarray(hits).view("<u8").reshape(hits.shape[0], hits.shape[1], 2)
        for b in range(nblk):
            sl = w[b * blk:(b + 1) * blk]
            state["sums"][b] = W.combine_checksums(state["sums"][b], W.words_checksum(sl[..., 0], sl[..., 1]))
        return 0

    def step(cb=on_chunk_timed):
        fresh()
        ctx.search_db(Q, T, mat, 24, args.gap_open, args.gap_extend, -1, 2, args.db_chunk, cb)

    # allocation warm-up, outside every timer: the library page-locks its two host buffers for nq x chunk records on the first search of a
    # context (of the order of a second) and keeps them -- one chunk of DB entries makes that happen before the first measured step
    Tw = ctx.upload(db[:min(nt, args.db_chunk)])
    ctx.search_db(Q, Tw, mat, 24, args.gap_open, args.gap_extend, -1, 2, args.db_chunk, lambda tfirst, hits: 0)
    Tw.free()
    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    fill_ms = 0.0; launches = 0; fill_cells = 0
    for _ in range(args.steps):
        step()
        tm = ctx.timing()
        fill_ms += tm["fill_ms"]; launches += tm["fill_launches"]; fill_cells += tm["fill_cells"]
    if dist is not None:
        dist.barrier()
    dt = max_over_ranks(dist, time.perf_counter() - t0)
    tm = ctx.timing()
    cells = float(tm["cells"])
    out = None
    if rank == 0 and world == 1:
        step(on_chunk_verify)          # untimed: the same search once more, every record folded into the parity checksums
    # ranks other than 0 have no committed fixture for their query block: their first queries x all entries through the unmodified reference
    rank_par = None
    if rank > 0 and args.cpu_sample != 0:
        Rr = ref_lib()
        if Rr is not None and hasattr(Rr, "refwrap_bench_db"):
            kq = min(nq, keep, 4)
            tc_, to_ = W.pack(db); qc_, qo_ = W.pack(qs[:kq])
            r5 = np.zeros((kq, nt, 5), dtype=np.int32)
            Rr.refwrap_bench_db(_ptr(qc_, i8p), _ptr(qo_, i64p), kq, _ptr(tc_, i8p), _ptr(to_, i64p), nt, _ptr(mat, i8p), 24, args.gap_open, args.gap_extend, -1, 2, _ptr(r5, i32p))
            kept = state["kept"][:kq]
            got = np.stack([kept["score1"], kept["score2"], kept["ref_end1"], kept["read_end1"], kept["ref_end2"]], axis=2).astype(np.int32)
            rank_par = {"rank": rank, "sample": kq * nt, "mismatching_alignments": int((got != r5).any(axis=2).sum()),
                        "against": "the unmodified reference (oracle/_ref) run by this rank on its first %d queries x all %d entries" % (kq, nt)}
    per_rank = gather_objects(dist, world, rank_par)
    if rank == 0:
        qsum = float(sum(len(x) for x in qs)); tsum = float(sum(len(x) for x in db))
        out = {"metric": "GCUPS", "value": round(cells * args.steps * world / dt / 1e9, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int16x2 in a column frame (value + phi(column)); reference u8/int16 semantics" if "frame" in tm["fill_kernel"] else "int16x2 (packed; reference u8/int16 semantics)",
               "data": "synthetic",
               "config": {"workload": "BASELINE config 5: %d protein queries (~300 aa) x %d DB entries per GPU, BLOSUM50, 3/1, score only, "
                                      "results streamed in chunks of %d entries" % (nq, nt, args.db_chunk), "baseline_config": 5,
                          "sharding": "query block r on rank r, DB replicated, no collective"},
               "alignments_per_step": nq * nt,
               "mix": {"word_rules": int(tm["n_word"]), "byte_rules": int(tm["n_byte"])},
               "phases_ms_per_step": {"fill(k_filldb, includes its fused reduction)": round(fill_ms / args.steps, 3), "other": round((dt * 1e3 - fill_ms) / args.steps, 3)}}
        launch_ms = fill_ms / max(1, launches)
        # algorithmic HBM bytes per alignment of the fused kernel: the target's codes, the query's codes, the matrix, a 16-byte record
        # (the two column-maximum streams the chain writes and re-reads for score2 are L2-resident scratch: 8 x target length more if counted)
        aln_per_launch = nq * nt * args.steps / max(1, launches)
        bytes_per_aln = tsum / nt + qsum / nq + 24 * 24 + 16
        achieved = aln_per_launch * bytes_per_aln / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        tr = measured_traffic("config5") if is_preset else None
        traffic = round(tr["hbm_bytes_per_alignment"] * aln_per_launch / (launch_ms * 1e-3) / 1e9, 2) if tr and launch_ms > 0 else None
        out["roofline_hbm"] = {"bound": "hbm", "kernel": tm["fill_kernel"] + " (largest size class; a launch = all size classes of one chunk of DB entries, side by side on 4 streams)",
                               "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "launch_ms": round(launch_ms, 3), "launches": int(launches), "algorithmic_bytes_per_alignment": round(bytes_per_aln, 1),
                               "note": "HBM is not the binding resource of this path (see roofline)"}
        hbm_traffic = traffic
        ops = tm["fill_ops_per_row"]
        achieved_valu = fill_cells * ops / 2.0 / (fill_ms * 1e-3) if fill_ms > 0 else 0.0
        real = cells * args.steps * ops / 2.0 / (fill_ms * 1e-3) if fill_ms > 0 else 0.0
        probe = measured_valu_probe(ctx) if args.lib is None else 0.0
        ideal = (CYCLES_PER_PAIR_ROW_ISA_IDEAL / CYCLES_PER_PAIR_ROW_4CYCLE) if ops == 6.5 else 1.0
        out["roofline"] = {"bound": "valu-issue", "kernel": tm["fill_kernel"] + " (largest size class)", "achieved": round(real / 1e12, 3), "peak": round(VALU_PEAK_LANEOPS / 1e12, 2),
                           "unit": "T lane-op/s", "frac": round(real / VALU_PEAK_LANEOPS, 4), "frac_with_padding_and_halo": round(achieved_valu / VALU_PEAK_LANEOPS, 4),
                           "frac_of_isa_ideal": round(real / VALU_PEAK_LANEOPS * ideal, 4),
                           "measured_peak_probe": round(probe / 1e12, 2), "ops_per_pair_row": ops,
                           "peak_note": "recurrence instructions only (%.1f per row of a query pair, each a 4-cycle issue slot in this mix: profiles/round3_mix_issue_probe.txt; "
                                        "ISA ideal %.1f cycles per pair-row); best-cell tracking and the fused reduction are overhead on top; `achieved` counts readLen x refLen cells, "
                                        "`frac_with_padding_and_halo` every evaluated cell" % (ops, CYCLES_PER_PAIR_ROW_ISA_IDEAL),
                           "fill_gcups_padded": round(fill_cells / (fill_ms * 1e-3) / 1e9, 1) if fill_ms > 0 else 0.0,
                           "counters": measured_issue("config5") if is_preset else None,
                           "traffic": hbm_traffic, "traffic_unit": "GB/s of HBM traffic (FETCH_SIZE + WRITE_SIZE), see roofline_hbm"}
        if True:
            par = {}
            fix = os.path.join(FULL, "config5_block0.npz")
            if is_preset and os.path.exists(fix):      # (the rows of the first queries are kept in the timed steps too: rank 0 of any N)
                z = np.load(fix)
                k = min(keep, int(z["nq"]))
                kept = state["kept"][:k]
                rows = np.stack([kept["score1"], kept["score2"], kept["ref_end1"], kept["read_end1"], kept["ref_end2"]], axis=2).astype(np.int32).reshape(k, -1)
                bad_rows = int((W.row_checksums(rows) != z["row_checksum"][:k]).sum())
                k16 = min(k, 16)
                bad16 = int((rows[:k16].reshape(k16, nt, 5) != z["first16"][:k16]).any(axis=2).sum())
                par = {"sample": k * nt, "mismatching_alignments": bad16, "queries_with_wrong_checksum": bad_rows,
                       "against": "tests/golden/full/config5_block0.npz: unmodified reference on the first %d queries x all %d entries "
                                  "(one checksum per query, full records of the first 16)" % (k, nt)}
            fixf = os.path.join(FULL, "config5_full_block0.npz")
            if is_preset and world == 1 and os.path.exists(fixf):      # (needs the extra verification search: N = 1 only)
                z = np.load(fixf)
                nb = min(nblk, int(z["done"]) // int(z["block"]))
                if nb > 0:
                    wrong = sum(1 for b in range(nb) if state["sums"][b] != (int(z["xor0"][b]), int(z["xor1"][b]), int(z["sums"][b])))
                    par["full_size"] = {"alignments": nb * blk * nt, "query_blocks_checked": nb, "query_blocks_with_wrong_checksum": wrong,
                                        "against": "tests/golden/full/config5_full_block0.npz: order-independent checksums (XOR + weighted sum of the "
                                                   "16-byte records) per 2048-query block over all entries, unmodified reference"}
            R = ref_lib()
            if args.cpu_sample != 0 and R is not None and hasattr(R, "refwrap_bench_db"):
                cores = usable_cores()
                tc, to = W.pack(db)

                def cpu_run(kq):
                    qc, qo = W.pack(qs[:kq])
                    r5 = np.zeros((kq, nt, 5), dtype=np.int32)
                    secs = R.refwrap_bench_db(_ptr(qc, i8p), _ptr(qo, i64p), kq, _ptr(tc, i8p), _ptr(to, i64p), nt, _ptr(mat, i8p), 24,
                                              args.gap_open, args.gap_extend, -1, cores, _ptr(r5, i32p))
                    return secs, r5, float(qo[-1]) * tsum
                if args.cpu_sample > 0:
                    kq = min(args.cpu_sample, nq, keep)
                else:
                    pilot = min(nq, cores, keep)
                    s0, _, _ = cpu_run(pilot)
                    kq = int(min(nq, keep, max(pilot, pilot / max(s0, 1e-3) * 12.0)))
                secs, r5, ccells = cpu_run(kq)
                out["cpu_baseline"] = {"value": round(ccells / secs / 1e9, 2), "unit": "GCUPS", "cores": cores, "kind": "reference",
                                       "per_core": round(ccells / secs / 1e9 / cores, 2),
                                       "sample": "first %d queries x all %d entries, one ssw_init per query + ssw_align per entry through the reference C API "
                                                 "(the loop of src/main.c), one thread per core, %.1f s" % (kq, nt, secs)}
                kept = state["kept"][:kq]
                got = np.stack([kept["score1"], kept["score2"], kept["ref_end1"], kept["read_end1"], kept["ref_end2"]], axis=2).astype(np.int32)
                par.setdefault("sample", kq * nt)
                par["cpu_sample"] = {"alignments": kq * nt, "mismatching_alignments": int((got != r5).any(axis=2).sum())}
                par.setdefault("mismatching_alignments", par["cpu_sample"]["mismatching_alignments"])
            if world > 1 and par:
                par["rank"] = 0
                got_par = [par] + [x for x in per_rank if x]
                par = {"sample": int(sum(x["sample"] for x in got_par)), "mismatching_alignments": int(sum(x["mismatching_alignments"] for x in got_par)),
                       "ranks_checked": len(got_par), "per_rank": got_par}
            if par:
                out["parity"] = par
        if not args.quiet:
            print(json.dumps(out))
            sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    Q.free(); T.free(); ctx.close()
    return out, state.get("kept")


def attach_also(args, out, world):
    """The driver-run line is BASELINE config 2; the other configs (3: 5 Mb target, 4: long reads with traceback, 5: protein database
    search) run for a few steps each AFTER the timed region -- `value` / `ms_per_step` above are not touched -- and their rate,
    roofline fraction and parity against the committed full-size fixtures go into `also`."""
    which = args.also
    if which is None:
        which = "2u8,3,4,5,6" if (args.config == 2 and world == 1 and not args.custom and args.pool == 0 and args.lib is None and
                            all(getattr(args, k) is None for k in ("reads", "flag", "mask_len"))) else "none"
    if which in ("none", ""):
        return
    also = {}
    t0 = time.perf_counter()
    for tok in which.split(","):
        u8 = tok.endswith("u8")        # "2u8": config 2 under the pure 8-bit scoring of SURVEY 8d (ii), 1/-3/5/2
        c = int(tok[:-2] if u8 else tok)
        scoring = ["--match", "1", "--mismatch", "3", "--gap-open", "5", "--gap-extend", "2"] if u8 else []
        sub = parse_args(["--config", str(c), "--steps", "1" if c == 5 else "5" if c == 6 else "2", "--warmup", "0" if c == 5 else "1", "--cpu-sample", "0"] + scoring +
                         (["--lib", args.lib] if args.lib else []))
        if args.lib:        # (tests on the emulator: small shapes)
            sub = parse_args(["--config", str(c), "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--lib", args.lib] + scoring +
                             (["--reads", "24", "--db-targets", "9", "--db-chunk", "4"] if c == 5 else
                              ["--reads", "6", "--ref-len", "3000"] if c == 6 else
                              ["--reads", "4", "--ref-len", "3000", "--read-len", "700" if c == 4 else "90"]))
        sub.quiet = True
        try:
            o, _ = bench_db(sub, 1, 0, 0, None) if c == 5 else bench_dna(sub, 1, 0, 0, None)
            also["config" + tok] = {"value": o["value"], "unit": "GCUPS", "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"],
                                    "workload": o["config"]["workload"], "dtype": o["dtype"],
                                    "fill_kernel": o.get("roofline", {}).get("kernel"),
                                    "roofline_frac": o.get("roofline", {}).get("frac"),
                                    "roofline_frac_with_padding_and_halo": o.get("roofline", {}).get("frac_with_padding_and_halo"),
                                    "roofline_hbm_frac": o.get("roofline_hbm", {}).get("frac"), "mix": o.get("mix"),
                                    "phases_ms_per_step": o.get("phases_ms_per_step"), "parity": o.get("parity")}
        except Exception as e:      # the metric's line must survive a failure here
            also["config" + tok] = {"error": "%s: %s" % (type(e).__name__, e)}
    also["seconds"] = round(time.perf_counter() - t0, 1)
    also["note"] = "run after the timed region of the metric's config, same process and device; parity = the GPU results of these runs against tests/golden/full"
    out["also"] = also


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU under
    torch.distributed.run on 127.0.0.1, exactly what the driver's command line does) and pass rank 0's line through.  With fewer
    visible devices than ranks, rank r uses device r mod devices (the ranks then share GPUs: a plumbing check, not a scaling number)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def main(argv=None):
    args = parse_args(argv)
    world, rank, local_rank, dist = init_dist()
    if args.gpus != world and world == 1 and args.gpus > 1 and args.pool == 0:
        self_launch(args, argv)
    if args.config == 5:
        return bench_db(args, world, rank, local_rank, dist)
    if args.full:
        return bench_config3_full(args, world, rank, local_rank, dist)
    return bench_dna(args, world, rank, local_rank, dist)


if __name__ == "__main__":
    main()
