"""Concurrency contract (include/ssw_gpu.h "Threads"):

* the single-pair ABI is re-entrant like the reference (src/ssw.c has no mutable global state): several threads call
  ssw_align at once, also on ONE shared profile, and each gets the reference's answer;
* a pool of workers (one context + host thread per device) pulls read blocks from a shared queue and produces exactly the
  records -- order, fields, CIGAR words -- of a single-context batch call.

CPU: the real host code on the SIMT emulator with SSW_EMU_DEVICES pretend-devices (every host thread runs its own
fibres).  GPU: the same through libssw.so (two workers share the one device of the test box)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import ssw_amd
from parity import compare_batch, make_reads
from sswutil import RES_FIELDS, dna_matrix, random_ref, ref_align, sample_reads

i8p = C.POINTER(C.c_int8)


def _threaded_single_pair(lib, nthreads=4, per_thread=6, reflen=500):
    rng = np.random.default_rng(31)
    mat = dna_matrix(2, 2)
    refs = [random_ref(reflen + 37 * i, 70 + i, 4, 0.01) for i in range(3)]
    reads = make_reads(rng, refs[0], nthreads * per_thread, rng.integers(20, 200, size=nthreads * per_thread), 4)
    shared_read = reads[0]
    shared = lib.ssw_init(shared_read.ctypes.data_as(i8p), len(shared_read), mat.ctypes.data_as(i8p), 5, 2)   # one profile, all threads
    errors = []

    def check(a, read, ref, flag, mask):
        exp, ecig = ref_align(read, mat, 5, ref, 3, 1, flag, 0, 0, mask)
        s = a.contents
        got = dict(score1=s.nScore, score2=s.nScore2, ref_begin1=s.nRefBeg, ref_end1=s.nRefEnd, read_begin1=s.nQryBeg, read_end1=s.nQryEnd,
                   ref_end2=s.nRefEnd2, cigarLen=s.nCigarLen, flag=s.nFlag)
        gc = [int(s.sCigar[i]) for i in range(s.nCigarLen)]
        if got != {k: exp[k] for k in RES_FIELDS} or gc != ecig:
            errors.append((got, exp))

    def worker(t):
        try:
            for k in range(per_thread):
                read = reads[t * per_thread + k]
                ref = refs[(t + k) % len(refs)]          # targets alternate: the per-thread target cache is hit and missed
                flag = (0, 1, 2)[k % 3]
                mask = max(15, len(read) // 2)
                p = lib.ssw_init(read.ctypes.data_as(i8p), len(read), mat.ctypes.data_as(i8p), 5, 2)
                for rep in range(2):                     # second call: same target bytes -> cached
                    a = lib.ssw_align(p, ref.ctypes.data_as(i8p), len(ref), 3, 1, flag, 0, 0, mask)
                    assert a
                    check(a, read, ref, flag, mask)
                    lib.align_destroy(a)
                lib.init_destroy(p)
                a = lib.ssw_align(shared, ref.ctypes.data_as(i8p), len(ref), 3, 1, 1, 0, 0, max(15, len(shared_read) // 2))
                assert a
                check(a, shared_read, ref, 1, max(15, len(shared_read) // 2))
                lib.align_destroy(a)
        except Exception as e:     # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    lib.init_destroy(shared)
    assert not errors, errors[:3]


def _pool_vs_single(lib, devices, nreads, reflen, block):
    rng = np.random.default_rng(57)
    mat = dna_matrix(2, 2)
    refs = [random_ref(reflen, 91, 4, 0.005), random_ref(reflen // 3, 92, 4)]
    lens = list(rng.integers(12, 180, size=nreads - 3)) + [400, 0, 150]          # a long read and an empty one ride along
    reads = make_reads(rng, refs[0], nreads, lens, 4)
    reads[nreads - 2] = np.zeros(0, dtype=np.int8)
    pool = ssw_amd.Pool(devices, lib)
    ctx = ssw_amd.Context(0, lib)
    try:
        pool.set_targets(refs)
        Q = ctx.upload(reads); T = ctx.upload(refs)
        for flag, mm in ((0, False), (2, False), (2, True)):
            res_p, cig_p = pool.align(reads, mat, 5, 3, 1, flag, block=block, mark_mismatch=mm)
            res_s, cig_s = ctx.align_batch(Q, T, mat, 5, 3, 1, flag, mark_mismatch=mm)
            for f in ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2", "cigarLen", "edit_distance", "flag", "status"):
                assert (res_p[f] == res_s[f]).all(), (flag, f)
            for q in range(nreads):
                for t in range(len(refs)):
                    a, b = res_p[q, t], res_s[q, t]
                    n = int(a["cigarLen"])
                    if n:
                        assert (cig_p[int(a["cigar_off"]):int(a["cigar_off"]) + n] == cig_s[int(b["cigar_off"]):int(b["cigar_off"]) + n]).all(), (flag, q, t)
                    else:
                        assert int(a["cigar_off"]) == -1
            st = pool.stats()
            assert sum(s["blocks"] for s in st) == -(-nreads // block) and sum(s["queries"] for s in st) == nreads
            if not mm:
                bad = compare_batch(res_p, cig_p, reads, refs, mat, 5, 3, 1, flag, 0, 0, -1, 2)
                assert not bad, "\n".join(bad)
        Q.free(); T.free()
        return pool.stats()
    finally:
        ctx.close(); pool.close()


def test_single_pair_abi_from_four_threads_emulated(emu_lib_path, monkeypatch):
    monkeypatch.setenv("SSW_EMU_DEVICES", "3")       # implicit contexts are spread round-robin over 3 pretend-devices
    _threaded_single_pair(ssw_amd.load(emu_lib_path), nthreads=4, per_thread=4, reflen=300)


def test_pool_two_workers_emulated(emu_lib_path, monkeypatch):
    monkeypatch.setenv("SSW_EMU_DEVICES", "2")
    st = _pool_vs_single(ssw_amd.load(emu_lib_path), None, nreads=23, reflen=600, block=4)
    assert len(st) == 2 and [s["device"] for s in st] == [0, 1]
    assert all(s["blocks"] > 0 for s in st)          # both queues did work


def test_pool_eight_workers_emulated(emu_lib_path, monkeypatch):
    """the per-GPU work queues at the node's size: 8 workers on 8 pretend-devices (an 8 x MI355X node), 36 blocks of reads"""
    monkeypatch.setenv("SSW_EMU_DEVICES", "8")
    st = _pool_vs_single(ssw_amd.load(emu_lib_path), None, nreads=36, reflen=300, block=1)
    assert len(st) == 8 and [s["device"] for s in st] == list(range(8))
    assert sum(1 for s in st if s["blocks"] > 0) >= 4      # the queues really spread the blocks (which worker wins a block is a race)


def test_pool_rejects_bad_use(emu_lib_path):
    lib = ssw_amd.load(emu_lib_path)
    pool = ssw_amd.Pool([0], lib)
    with pytest.raises(RuntimeError, match="no target set"):
        pool.align([np.zeros(5, dtype=np.int8)], dna_matrix(2, 2), 5)
    pool.close()


@pytest.mark.gpu
def test_single_pair_abi_from_four_threads_gpu(product_lib_path):
    _threaded_single_pair(ssw_amd.load(product_lib_path), nthreads=4, per_thread=8, reflen=20000)


@pytest.mark.gpu
def test_pool_two_workers_one_gpu(product_lib_path):
    st = _pool_vs_single(ssw_amd.load(product_lib_path), [0, 0], nreads=1500, reflen=30000, block=128)
    assert len(st) == 2 and sum(s["queries"] for s in st) == 1500


# ---- return codes, the sized timing record, per-context budgets ----
def test_busy_context_returns_its_own_code_and_timing_is_sized(emu_lib_path):
    """a second thread entering a busy context gets SSW_GPU_BUSY (-2, named by ssw_gpu_strerror; no stale message is reported);
    ssw_gpu_last_timing_sized writes no more than the caller's sizeof; a chunk function's negative return value comes back as it is"""
    import ctypes as C
    import threading
    lib = ssw_amd.load(emu_lib_path)
    assert lib.ssw_gpu_strerror(-2).decode().startswith("the context is inside another call")
    ctx = ssw_amd.Context(0, lib)
    ref = random_ref(4000, 3, 4)
    reads = sample_reads(ref, 6, 100, seed=4)
    Q = ctx.upload(reads); T = ctx.upload([ref, ref[:900].copy(), ref[1000:2500].copy(), ref[50:700].copy()])
    seen = []

    def worker():
        try:
            ctx.align_batch(Q, T, dna_matrix(2, 2), 5, 3, 1, 2, 0, 0, -1, 2)
            seen.append("done")
        except RuntimeError as e:
            seen.append(str(e))
    th = [threading.Thread(target=worker) for _ in range(3)]
    for t in th: t.start()
    for t in th: t.join()
    assert "done" in seen and all(s == "done" or "inside another call" in s for s in seen), seen
    # sized timing: 24 bytes = total_ms, fill_ms, fill_launches; the rest of a larger caller buffer is left alone
    buf = (C.c_ubyte * 64)(*([0xEE] * 64))
    assert lib.ssw_gpu_last_timing_sized(ctx.h, buf, 24) == 0
    assert bytes(buf[24:]) == b"\xee" * 40 and C.cast(buf, C.POINTER(C.c_double))[0] > 0
    # a negative stop value of the caller's chunk function is returned, not raised
    assert ctx.search_db(Q, T, dna_matrix(2, 2), 5, 3, 1, -1, 2, 2, lambda t0, h: -7) == -7
    # budget setter
    b0 = lib.ssw_gpu_get_budget(ctx.h)
    assert lib.ssw_gpu_set_budget(ctx.h, 64 << 20) == 0 and lib.ssw_gpu_get_budget(ctx.h) == 64 << 20
    assert lib.ssw_gpu_set_budget(ctx.h, 0) == 0 and lib.ssw_gpu_get_budget(ctx.h) == b0
    Q.free(); T.free(); ctx.close()


def test_pool_workers_on_one_device_share_its_budget(emu_lib_path, monkeypatch):
    """every context sizes its scratch budget from the HBM that is free when IT is opened; a pool divides that by the workers it put
    on the same device (an explicit SSW_GPU_CM_BUDGET_MB is per context and kept as it is)"""
    monkeypatch.setenv("SSW_EMU_DEVICES", "2")
    monkeypatch.delenv("SSW_GPU_CM_BUDGET_MB", raising=False)
    lib = ssw_amd.load(emu_lib_path)
    solo = ssw_amd.Context(0, lib); full = lib.ssw_gpu_get_budget(solo.h); solo.close()
    pool = ssw_amd.Pool([0, 0, 1], lib)
    try:
        b = [lib.ssw_gpu_pool_budget(pool.h, i) for i in range(3)]
        assert b[0] == full // 2 and b[1] == full // 2 and b[2] == full, (b, full)
    finally:
        pool.close()
    monkeypatch.setenv("SSW_GPU_CM_BUDGET_MB", "32")
    pool = ssw_amd.Pool([0, 0], lib)
    try:
        assert [lib.ssw_gpu_pool_budget(pool.h, i) for i in range(2)] == [32 << 20, 32 << 20]
    finally:
        pool.close()
