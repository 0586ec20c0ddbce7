#!/usr/bin/env python3
"""bench.py -- GCUPS of the Smith-Waterman hot path (BASELINE.json metric) on N MI355X of one node.

A "step" is one pass of the whole hot path (forward fill + reduction + read_end1 location) over one batch of
synthetic reads that is already resident in HBM:  BASELINE config 2 -- 100k x 150 bp DNA reads (sampled from the
target, ~3% substitutions, ~1% indels, 5% random reads) against a 1 Mb random target, match 2 / mismatch -2 /
gap open 3 / gap extension 1, score_size 2 (8-bit rules with 16-bit fallback), score-only (flag 0) like the
reference CLI's default run.  With N > 1 every rank aligns its own 100k-read shard against a replicated target
(no collective on the data path): weak scaling.  GCUPS counts readLen x refLen of the forward matrix only.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline      HBM view of the dominant kernel (k_fill): algorithmic bytes per launch / measured launch time
  roofline_valu the binding roofline of this integer max-plus recurrence: packed-int16 VALU issue rate
  cpu_baseline  the unmodified reference (oracle/_ref, its SSE2 path) timed on this host's cores on a bounded sample
  parity        the same sample compared bit-exactly with the GPU results of the timed batch
"""
import argparse
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
                                            # = 39.3e12 packed lane-instr/s = 78.6e12 16-bit values/s, the same datapath rate as v_fma_f32 (profiles/round1_valu_rate_probe.txt)


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


def make_reads_fast(ref, nreads, length, seed, sub=0.03, ins=0.005, dele=0.005, frac_random=0.05):
    """vectorised version of tests/sswutil.sample_reads (same model, different stream): [nreads, length] int8"""
    rng = np.random.default_rng(seed)
    span = length + 32
    off = rng.integers(0, len(ref) - span, size=nreads)
    is_ins = rng.random((nreads, length)) < ins
    is_del = (rng.random((nreads, length)) < dele) & ~is_ins
    consumed = np.cumsum(~is_ins, axis=1) - 1
    deleted = np.cumsum(is_del, axis=1)
    src = off[:, None] + np.clip(consumed + deleted, 0, span - 1)
    reads = ref[src]
    rnd = rng.integers(0, 4, size=(nreads, length), dtype=np.int8)
    reads = np.where(is_ins, rnd, reads)
    do_sub = rng.random((nreads, length)) < sub
    reads = np.where(do_sub, (reads + 1 + rng.integers(0, 3, size=(nreads, length))) % 4, reads)
    whole = rng.random(nreads) < frac_random
    reads[whole] = rnd[whole]
    return np.ascontiguousarray(reads, dtype=np.int8)


def bench_db(args, rank, world, local_rank, dist):
    """BASELINE config 5 shape: every protein query (len ~ N(300, 60) clipped to [50, 1000], background residue
    frequencies, 1 % planted homologs) against every DB entry, BLOSUM50, gaps 3/1, score-only (fused k_filldb path)."""
    import ctypes as C
    import ssw_amd
    from sswutil import blosum50, mutate, ref_lib, _ptr, i8p, i32p, i64p
    lib = ssw_amd.load(args.lib)
    ctx = ssw_amd.Context(local_rank % max(1, lib.ssw_gpu_device_count()), lib)
    rng = np.random.default_rng(4 + rank)
    freq = np.array([8.3, 5.5, 4.1, 5.5, 1.4, 3.9, 6.8, 7.1, 2.3, 5.9, 9.7, 5.8, 2.4, 3.9, 4.7, 6.6, 5.3, 1.1, 2.9, 6.9]); freq /= freq.sum()
    def seqs(count):
        lens = np.clip(rng.normal(300, 60, size=count), 50, 1000).astype(np.int64)
        return [rng.choice(20, size=int(L), p=freq).astype(np.int8) for L in lens]
    db = seqs(args.db_targets)
    qs = seqs(args.reads)
    for i in range(0, args.reads, 100):          # planted homologs: 1 % of the queries are mutated copies of a DB entry
        src = db[int(rng.integers(0, len(db)))]
        m = mutate(src, rng, 0.15, 0.02, 0.02, 20)
        if len(m) >= 50:
            qs[i] = m[:1000]
    mat = blosum50()
    Q = ctx.upload(qs); T = ctx.upload(db)
    res_buf = ctx.result_array(len(qs), len(db))     # page-locked: nq x nt records come back at PCIe rate (include/ssw_gpu.h)
    def step():
        return ctx.align_batch(Q, T, mat, 24, 3, 1, 0, 0, 0, -1, 2, want_cigar=False, out=res_buf)
    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    fill_ms = 0.0
    for _ in range(args.steps):
        res, _ = step()
        fill_ms += ctx.timing()["fill_ms"]
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    tm = ctx.timing()
    if dist is not None:
        import torch
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    cells = float(tm["cells"])
    out = None
    if rank == 0:
        out = {"metric": "GCUPS", "value": round(cells * args.steps * world / dt / 1e9, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int16x2 (packed; reference u8/int16 semantics)", "data": "synthetic",
               "config": {"workload": "BASELINE config 5 shape: %d protein queries (~300 aa) x %d DB entries per GPU, BLOSUM50, 3/1, score-only"
                                      % (args.reads, args.db_targets)},
               "alignments_per_step": args.reads * args.db_targets,
               "mix": {"word_rules": int(tm["n_word"]), "byte_rules": int(tm["n_byte"])},
               "phases_ms_per_step": {"fill(k_filldb/k_chainx)": round(fill_ms / args.steps, 3), "other": round((dt * 1e3 - fill_ms) / args.steps, 3)},
               "fill_gcups_padded": round(tm["fill_cells"] / (tm["fill_ms"] * 1e-3) / 1e9, 1) if tm["fill_ms"] > 0 else 0.0}
        R = ref_lib()
        if world == 1 and args.cpu_sample != 0 and R is not None:
            cores = usable_cores()
            ns = min(args.reads, max(cores, 512)); ntc = min(args.db_targets, 8)
            codes = np.concatenate(qs[:ns]).astype(np.int8); off = np.zeros(ns + 1, dtype=np.int64)
            off[1:] = np.cumsum([len(x) for x in qs[:ns]])
            secs = 0.0; mism = 0; ccells = 0
            for t in range(ntc):
                cres = np.zeros((ns, 10), dtype=np.int32)
                tg = np.ascontiguousarray(db[t])
                secs += R.refwrap_bench(_ptr(codes, i8p), _ptr(off, i64p), ns, _ptr(tg, i8p), len(tg), _ptr(mat, i8p), 24, 3, 1, 0, 0, 0, -1, cores,
                                        _ptr(cres, i32p))
                g = res[:ns, t]
                got = np.stack([g["score1"], g["score2"], g["ref_begin1"], g["ref_end1"], g["read_begin1"], g["read_end1"], g["ref_end2"]], axis=1).astype(np.int32)
                mism += int((got != cres[:, :7]).any(axis=1).sum())
                ccells += int(off[ns]) * len(tg)
            out["cpu_baseline"] = {"value": round(ccells / secs / 1e9, 2), "unit": "GCUPS", "cores": cores, "kind": "reference",
                                   "sample": "first %d queries x first %d DB entries through the reference C API, one thread per core, %.2f s" % (ns, ntc, secs)}
            out["parity"] = {"sample": ns * ntc, "mismatching_alignments": mism}
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    Q.free(); T.free(); ctx.close()
    return out, res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000, help="reads per GPU and step (config 2: 100k)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--ref-len", type=int, default=1_000_000)
    ap.add_argument("--match", type=int, default=2, help="match score (ssw_test -m)")
    ap.add_argument("--mismatch", type=int, default=2, help="mismatch penalty (ssw_test -x)")
    ap.add_argument("--gap-open", type=int, default=3, help="gap opening penalty (ssw_test -o)")
    ap.add_argument("--gap-extend", type=int, default=1, help="gap extension penalty (ssw_test -e)")
    ap.add_argument("--flag", type=int, default=0, help="ssw_align flag (0 = scores + end positions; 2 = + begin + CIGAR)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads in the CPU-baseline sample (-1: ~20 s of work; 0: skip)")
    ap.add_argument("--sub", type=float, default=0.03, help="substitution rate of the synthetic reads")
    ap.add_argument("--indel", type=float, default=0.005, help="insertion rate = deletion rate of the synthetic reads")
    ap.add_argument("--mask-len", type=int, default=-1, help="maskLen (-1: readLen/2 per read, like the reference CLI)")
    ap.add_argument("--db-targets", type=int, default=0,
                    help="> 0: BASELINE config 5 shape instead -- --reads protein queries (~300 aa) against this many DB entries, BLOSUM50")
    ap.add_argument("--lib", default=None, help=argparse.SUPPRESS)   # tests point this at the emulated library
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:
        # One process per GPU.  The data path has no collective (reads are sharded, the target is replicated), so the
        # only cross-rank traffic is the timing barrier + MAX: it runs over gloo on CPU tensors, which keeps torch's own
        # HIP runtime out of the processes' data path (libssw.so drives its GPU through its own stream).
        # SSW_BENCH_BACKEND=nccl switches the barrier to RCCL.
        import torch
        import torch.distributed as dist_mod
        backend = os.environ.get("SSW_BENCH_BACKEND", "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist_mod.init_process_group(backend=backend)
        dist = dist_mod
    ngpus = world
    if args.gpus != ngpus and world == 1 and args.gpus > 1:
        print("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus),
              file=sys.stderr)
        sys.exit(2)

    import ssw_amd
    from sswutil import dna_matrix, random_ref
    if args.db_targets > 0:
        return bench_db(args, rank, world, local_rank, dist)
    lib = ssw_amd.load(args.lib)
    if lib.ssw_gpu_device_count() < 1:
        raise RuntimeError("bench.py: no HIP device visible; libssw.so has no CPU path")
    ndev = lib.ssw_gpu_device_count()
    ctx = ssw_amd.Context(local_rank % ndev, lib)

    mat = dna_matrix(args.match, args.mismatch)
    ref = random_ref(args.ref_len, 1, 4)                                   # seed 1: BASELINE config 2
    reads = make_reads_fast(ref, args.reads, args.read_len, seed=1000 + rank, sub=args.sub, ins=args.indel, dele=args.indel)
    # upload through the packed form directly (a Python list of 100k arrays is slow to concatenate)
    import ctypes as C
    off = (np.arange(args.reads + 1, dtype=np.int64) * args.read_len)
    qh = lib.ssw_gpu_seqs_upload(ctx.h, reads.ctypes.data_as(C.POINTER(C.c_int8)), off.ctypes.data_as(C.POINTER(C.c_int64)), args.reads)
    if not qh:
        raise RuntimeError(ctx.error())
    Q = ssw_amd.Seqs.__new__(ssw_amd.Seqs); Q.ctx = ctx; Q.count = args.reads; Q.h = qh
    T = ctx.upload([ref])

    def step():
        return ctx.align_batch(Q, T, mat, 5, args.gap_open, args.gap_extend, args.flag, 0, 0, args.mask_len, 2, want_cigar=(args.flag & 7) != 0)

    def sync_all():
        # align_batch() returns only after its stream is synchronised and the results are on the host, so every rank is
        # idle here; the barrier aligns the ranks' clocks around the timed region.
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    fill_ms = 0.0; fill_launches = 0; fill_cells = 0; phase = {"reduce_ms": 0.0, "locate_ms": 0.0, "trace_ms": 0.0, "total_ms": 0.0}
    res = None
    for _ in range(args.steps):
        res, cig = step()            # align_batch returns with results on the host (stream synchronised inside)
        tm = ctx.timing()
        fill_ms += tm["fill_ms"]; fill_launches += tm["fill_launches"]; fill_cells += tm["fill_cells"]
        for k in phase:
            phase[k] += tm[k]
    sync_all()
    dt = time.perf_counter() - t0
    tm = ctx.timing()
    if dist is not None:
        import torch
        dev = "cuda" if os.environ.get("SSW_BENCH_BACKEND", "gloo") == "nccl" else "cpu"
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    cells_per_step = float(args.reads) * args.read_len * args.ref_len
    value = cells_per_step * args.steps * ngpus / dt / 1e9

    out = None
    if rank == 0:
        launch_ms = fill_ms / max(1, fill_launches)
        # algorithmic HBM bytes of one k_fill launch (SURVEY 8d / DESIGN.md): per alignment refLen target codes read,
        # readLen + n^2 query/matrix bytes, 4*refLen column-maximum bytes written (2 rule sets x u16), 40 B result
        aln_per_launch = args.reads * args.steps / max(1, fill_launches)
        bytes_per_aln = args.ref_len + args.read_len + 25 + 40 + 4 * args.ref_len
        if args.read_len <= 384:
            _top = 16 * ((args.read_len + 15) // 16) * max(args.match, 0)
            _on = os.environ.get("SSW_GPU_FILL_F16", "1") != "0"
            fill_kernel = "k_fill<%d, %s>" % ((args.read_len + 15) // 16, "f16" if (_on and _top <= 2047) else "int16+max3" if (_on and _top < 31744) else "int16")
        else:   # long queries: row strips of 64 x R rows (csrc/ssw_host.c); boundary records of 16 B per column and pair between strips
            p16 = (args.read_len + 15) // 16 * 16
            strips = (p16 + 64 * 12 - 1) // (64 * 12)
            fill_kernel = "k_chainx<%d, false, 64> x %d strips" % ((p16 + 64 * strips - 1) // (64 * strips), strips)
            bytes_per_aln += 16 * args.ref_len * (strips - 1)      # written once, read once, shared by the two queries of a pair
        achieved_gbs = aln_per_launch * bytes_per_aln / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        # VALU view: 9 packed int16 instructions per (row, column) for two queries -> 4.5 lane-ops per evaluated cell; 7.5 in the
        # f16 form that csrc/ssw_host.c selects when no score of the bucket can reach 2048 (short reads, small match scores)
        top = 16 * ((args.read_len + 15) // 16) * max(args.match, 0)          # no cell of the batch scores more
        forms_on = os.environ.get("SSW_GPU_FILL_F16", "1") != "0"
        f16_form = args.read_len <= 384 and top <= 2047 and forms_on
        cm3_form = args.read_len <= 384 and not f16_form and top < 31744 and forms_on
        ops_per_pair_cell = 7.5 if f16_form else 8.5 if cm3_form else 9     # csrc/ssw_kernels.hip k_fill<R, FORM>
        valu_ops = fill_cells * ops_per_pair_cell / 2.0
        achieved_valu = valu_ops / (fill_ms * 1e-3) if fill_ms > 0 else 0.0
        probe = ctx.valu_probe(8192, 4000) if args.lib is None else 0.0   # (skipped on the test emulator)
        traffic = None
        try:   # HBM bytes per launch from the committed PMC passes (profiles/), scaled to this run's pairs per launch
            with open(os.path.join(ROOT, "profiles", "round1_traffic.json")) as f:
                tj = json.load(f)
            if args.read_len == 150 and args.ref_len == 1_000_000:
                traffic = round(tj["hbm_bytes_per_pair"] * (aln_per_launch / 2.0) / (launch_ms * 1e-3) / 1e9, 2)   # GB/s, like `achieved`
        except Exception:
            traffic = None
        out = {
            "metric": "GCUPS", "value": round(value, 2), "unit": "GCUPS", "n_gpus": ngpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f16x2 holding exact integers (scores/2048); reference u8/int16 semantics" if f16_form else
                      "int16x2 (packed; reference u8/int16 semantics)"), "data": "synthetic",
            "config": {"workload": "%s: %d x %d bp DNA reads vs %.1f Mb target per GPU, %d/-%d/%d/%d, score_size 2, flag %d"
                                   % ("BASELINE config 2" if (args.reads, args.read_len, args.ref_len) == (100000, 150, 1000000) else "custom",
                                      args.reads, args.read_len, args.ref_len / 1e6, args.match, args.mismatch, args.gap_open, args.gap_extend, args.flag),
                       "reads_per_gpu": args.reads, "read_len": args.read_len, "ref_len": args.ref_len,
                       "sharding": "reads sharded across ranks, target replicated, no collective"},
            "mix": {"word_rules": int(tm["n_word"]), "byte_rules": int(tm["n_byte"])},
            "phases_ms_per_step": {"fill": round(fill_ms / args.steps, 3), "locate": round(phase["locate_ms"] / args.steps, 3),
                                   "trace": round(phase["trace_ms"] / args.steps, 3),
                                   "reduce_and_copies": round(phase["reduce_ms"] / args.steps, 3)},
            "roofline": {"bound": "hbm", "kernel": fill_kernel, "achieved": round(achieved_gbs, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_note": "GB/s from rocprofv3 FETCH_SIZE(x2)+WRITE_SIZE of this kernel (profiles/round1_traffic.json), per launch",
                         "launch_ms": round(launch_ms, 3), "launches": int(fill_launches),
                         "note": "integer max-plus recurrence: HBM is not the binding resource, see roofline_valu"},
            "roofline_valu": {"bound": "valu-packed16", "achieved": round(achieved_valu / 1e12, 3), "peak": round(VALU_PEAK_LANEOPS / 1e12, 2),
                              "unit": "T lane-op/s", "frac": round(achieved_valu / VALU_PEAK_LANEOPS, 4),
                              "measured_peak_probe": round(probe / 1e12, 2),
                              "ops_per_pair_cell": ops_per_pair_cell,
                              "note": "packed 16-bit (VOP3P) instruction rate: 16 lanes/clk/SIMD (peak = 256 CU x 4 SIMD x 16 x 2.4 GHz); the probe is the int16 mix measured on this device",
                              "fill_gcups_padded": round(fill_cells / (fill_ms * 1e-3) / 1e9, 1) if fill_ms > 0 else 0.0},
        }
        # ---- CPU baseline + parity on a bounded sample (rank 0, N = 1 only) ----
        if ngpus == 1 and args.cpu_sample != 0:
            from sswutil import ref_lib, oracle_align, _ptr, i8p, i32p, i64p
            cores = usable_cores()
            R = ref_lib()
            if R is not None:
                def cpu_run(k):
                    sample = np.ascontiguousarray(reads[:k])
                    soff = np.arange(k + 1, dtype=np.int64) * args.read_len
                    cres = np.zeros((k, 10), dtype=np.int32)
                    secs = R.refwrap_bench(_ptr(sample, i8p), _ptr(soff, i64p), k, _ptr(ref, i8p), len(ref), _ptr(mat, i8p), 5, args.gap_open, args.gap_extend,
                                           args.flag, 0, 0, args.mask_len, cores, _ptr(cres, i32p))
                    return secs, cres
                if args.cpu_sample > 0:
                    ns = min(args.cpu_sample, args.reads)
                else:   # pilot of one read per core, then a sample sized for ~15 s of wall-clock on all cores
                    pilot = min(args.reads, cores)
                    s0, _ = cpu_run(pilot)
                    ns = int(min(args.reads, max(pilot, pilot / max(s0, 1e-3) * 15.0)))
                secs, cres = cpu_run(ns)
                cpu_gcups = ns * args.read_len * args.ref_len / secs / 1e9
                g = res[:ns, 0]
                got = np.stack([g["score1"], g["score2"], g["ref_begin1"], g["ref_end1"], g["read_begin1"], g["read_end1"], g["ref_end2"],
                                g["cigarLen"], g["flag"]], axis=1).astype(np.int32)
                mism = int((got != cres[:, :9]).any(axis=1).sum())
                out["cpu_baseline"] = {"value": round(cpu_gcups, 2), "unit": "GCUPS", "cores": cores, "kind": "reference",
                                       "per_core": round(cpu_gcups / cores, 2), "host_logical_cpus": os.cpu_count(),
                                       "cores_note": "threads = CPUs this container may use (affinity capped by the cgroup cpu.max quota)",
                                       "sample": "first %d reads of the batch vs the same target, ssw_init(...,2)+ssw_align through the C API, "
                                                 "reference ssw.c built -O2 (oracle/_ref), one thread per core, %.1f s" % (ns, secs)}
                out["parity"] = {"sample": ns, "mismatching_alignments": mism, "fields": "score1 score2 ref/read begin/end ref_end2 cigarLen flag"}
            else:
                ns = args.cpu_sample if args.cpu_sample > 0 else 2
                t1 = time.perf_counter()
                mism = 0
                for i in range(ns):
                    d, _ = oracle_align(reads[i], mat, 5, ref, args.gap_open, args.gap_extend, args.flag, 0, 0, args.read_len // 2, 2, 0)
                    g = res[i, 0]
                    mism += any(int(g[k]) != d[k] for k in ("score1", "score2", "ref_end1", "read_end1", "ref_end2"))
                secs = time.perf_counter() - t1
                out["cpu_baseline"] = {"value": round(ns * args.read_len * args.ref_len / secs / 1e9, 3), "unit": "GCUPS", "cores": 1,
                                       "kind": "port", "sample": "%d reads, scalar lane-model oracle (oracle/_ref not shipped)" % ns}
                out["parity"] = {"sample": ns, "mismatching_alignments": mism}
        print(json.dumps(out))
        sys.stdout.flush()
    dump = os.environ.get("SSW_BENCH_DUMP")
    if dump:   # tests: keep every rank's shard and results for an independent check
        np.savez(os.path.join(dump, "rank%d.npz" % rank), reads=reads, ref=ref, res=res)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    Q.free(); T.free(); ctx.close()
    return out, res


if __name__ == "__main__":
    main()
