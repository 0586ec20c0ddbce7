"""ctypes binding of libssw.so (the MI355X-native Smith-Waterman library).

Two layers, both thin:

* ``CSsw`` mirrors the reference's own ctypes wrapper (reference src/ssw_lib.py:94-197: class
  ``CSsw`` with ``ssw_init`` / ``ssw_align`` / ``init_destroy`` / ``align_destroy`` and the
  ``CAlignRes`` field layout of src/ssw_lib.py:61-69) so that code written against it runs unchanged.
* ``Context`` / ``Seqs`` / ``align_batch`` bind the batch ABI of include/ssw_gpu.h.

This module never computes alignments itself: if the shared library (and through it the HIP device)
is missing, loading or calling fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libssw.so")

_i8p = C.POINTER(C.c_int8)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)


class CAlignRes(C.Structure):
    """s_align (include/ssw.h; reference src/ssw.h:55-66)."""
    _fields_ = [("nScore", C.c_uint16), ("nScore2", C.c_uint16), ("nRefBeg", C.c_int32), ("nRefEnd", C.c_int32),
                ("nQryBeg", C.c_int32), ("nQryEnd", C.c_int32), ("nRefEnd2", C.c_int32), ("sCigar", _u32p),
                ("nCigarLen", C.c_int32), ("nFlag", C.c_uint16)]


class Params(C.Structure):
    """ssw_gpu_params (include/ssw_gpu.h)."""
    _fields_ = [("mat", _i8p), ("n", C.c_int32), ("gapO", C.c_uint8), ("gapE", C.c_uint8), ("flag", C.c_uint8),
                ("filters", C.c_uint16), ("filterd", C.c_int32), ("maskLen", C.c_int32), ("score_size", C.c_int8),
                ("mark_mismatch", C.c_int8)]


class Result(C.Structure):
    """ssw_gpu_result (include/ssw_gpu.h)."""
    _fields_ = [("score1", C.c_uint16), ("score2", C.c_uint16), ("ref_begin1", C.c_int32), ("ref_end1", C.c_int32),
                ("read_begin1", C.c_int32), ("read_end1", C.c_int32), ("ref_end2", C.c_int32), ("cigarLen", C.c_int32),
                ("edit_distance", C.c_int32), ("cigar_off", C.c_int64), ("flag", C.c_uint16), ("status", C.c_uint16)]


class Timing(C.Structure):
    """ssw_gpu_timing (include/ssw_gpu.h)."""
    _fields_ = [("total_ms", C.c_double), ("fill_ms", C.c_double), ("fill_launches", C.c_int64),
                ("fill_cells", C.c_int64), ("cells", C.c_int64), ("reduce_ms", C.c_double), ("locate_ms", C.c_double),
                ("trace_ms", C.c_double), ("n_word", C.c_int64), ("n_byte", C.c_int64), ("fill_kernel", C.c_char * 48),
                ("fill_ops_per_row", C.c_double), ("fill_rows_per_lane", C.c_int32), ("fill_strips", C.c_int32),
                ("db_repeats", C.c_int64), ("fill_pipelined", C.c_int64)]


HIT_DTYPE = np.dtype([("score1", "<u2"), ("score2", "<u2"), ("ref_end1", "<i4"), ("read_end1", "<i4"), ("ref_end2", "<i4")], align=True)
HITS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p)     # ssw_gpu_hits_fn (include/ssw_gpu.h)


class PoolStat(C.Structure):
    """ssw_gpu_pool_stat (include/ssw_gpu.h)."""
    _fields_ = [("device", C.c_int32), ("blocks", C.c_int64), ("queries", C.c_int64), ("cells", C.c_int64), ("busy_ms", C.c_double)]


RESULT_DTYPE = np.dtype([("score1", "<u2"), ("score2", "<u2"), ("ref_begin1", "<i4"), ("ref_end1", "<i4"),
                         ("read_begin1", "<i4"), ("read_end1", "<i4"), ("ref_end2", "<i4"), ("cigarLen", "<i4"),
                         ("edit_distance", "<i4"), ("cigar_off", "<i8"), ("flag", "<u2"), ("status", "<u2")], align=True)
assert RESULT_DTYPE.itemsize == C.sizeof(Result)


def load(path=None):
    path = path or os.environ.get("SSW_LIB", DEFAULT_LIB)
    if not os.path.exists(path):
        raise OSError("libssw.so not built: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
    L = C.CDLL(path)
    L.ssw_init.argtypes = [_i8p, C.c_int32, _i8p, C.c_int32, C.c_int8]
    L.ssw_init.restype = C.c_void_p
    L.init_destroy.argtypes = [C.c_void_p]
    L.init_destroy.restype = None
    L.ssw_align.argtypes = [C.c_void_p, _i8p, C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int32,
                            C.c_int32]
    L.ssw_align.restype = C.POINTER(CAlignRes)
    L.align_destroy.argtypes = [C.POINTER(CAlignRes)]
    L.align_destroy.restype = None
    L.mark_mismatch.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i8p, _i8p, C.c_int32, C.POINTER(_u32p),
                                C.POINTER(C.c_int32)]
    L.mark_mismatch.restype = C.c_int32
    L.ssw_gpu_device_count.restype = C.c_int
    L.ssw_gpu_open.argtypes = [C.c_int]
    L.ssw_gpu_open.restype = C.c_void_p
    L.ssw_gpu_close.argtypes = [C.c_void_p]
    L.ssw_gpu_close.restype = None
    L.ssw_gpu_last_error.argtypes = [C.c_void_p]
    L.ssw_gpu_last_error.restype = C.c_char_p
    L.ssw_gpu_seqs_upload.argtypes = [C.c_void_p, _i8p, _i64p, C.c_int32]
    L.ssw_gpu_seqs_upload.restype = C.c_void_p
    L.ssw_gpu_seqs_upload_ascii.argtypes = [C.c_void_p, C.c_char_p, _i64p, C.c_int32, _i8p]
    L.ssw_gpu_seqs_upload_ascii.restype = C.c_void_p
    L.ssw_gpu_seqs_revcomp.argtypes = [C.c_void_p, C.c_void_p]
    L.ssw_gpu_seqs_revcomp.restype = C.c_void_p
    L.ssw_gpu_seqs_download.argtypes = [C.c_void_p, C.c_void_p, _i8p]
    L.ssw_gpu_seqs_download.restype = C.c_int
    L.ssw_gpu_seqs_free.argtypes = [C.c_void_p]
    L.ssw_gpu_seqs_free.restype = None
    L.ssw_gpu_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Params),
                                      C.c_void_p, C.POINTER(_u32p), _i64p]
    L.ssw_gpu_align_batch.restype = C.c_int
    L.ssw_gpu_last_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
    L.ssw_gpu_last_timing.restype = C.c_int
    L.ssw_gpu_last_timing_sized.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ssw_gpu_last_timing_sized.restype = C.c_int
    L.ssw_gpu_strerror.argtypes = [C.c_int]
    L.ssw_gpu_strerror.restype = C.c_char_p
    L.ssw_gpu_set_budget.argtypes = [C.c_void_p, C.c_size_t]
    L.ssw_gpu_set_budget.restype = C.c_int
    L.ssw_gpu_get_budget.argtypes = [C.c_void_p]
    L.ssw_gpu_get_budget.restype = C.c_size_t
    L.ssw_gpu_set_budget_exclusive.argtypes = [C.c_void_p]
    L.ssw_gpu_set_budget_exclusive.restype = C.c_int
    L.ssw_gpu_pool_budget.argtypes = [C.c_void_p, C.c_int]
    L.ssw_gpu_pool_budget.restype = C.c_size_t
    L.ssw_gpu_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
    L.ssw_gpu_host_alloc.restype = C.c_void_p
    L.ssw_gpu_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.ssw_gpu_host_free.restype = None
    if hasattr(L, "ssw_gpu_valu_probe"):      # diagnostics (include/ssw_gpu_diag.h): libssw_hooks.so and the test emulator only
        L.ssw_gpu_selftest_lanes.argtypes = [C.c_void_p, _u32p]
        L.ssw_gpu_selftest_lanes.restype = C.c_int
        L.ssw_gpu_valu_probe.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.ssw_gpu_valu_probe.restype = C.c_double
    L.ssw_gpu_release_parked.restype = C.c_int
    L.ssw_gpu_search_db.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_int32, HITS_FN, C.c_void_p]
    L.ssw_gpu_search_db.restype = C.c_int
    L.ssw_gpu_pool_open.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.ssw_gpu_pool_open.restype = C.c_void_p
    L.ssw_gpu_pool_close.argtypes = [C.c_void_p]
    L.ssw_gpu_pool_close.restype = None
    L.ssw_gpu_pool_size.argtypes = [C.c_void_p]
    L.ssw_gpu_pool_last_error.argtypes = [C.c_void_p]
    L.ssw_gpu_pool_last_error.restype = C.c_char_p
    L.ssw_gpu_pool_set_targets.argtypes = [C.c_void_p, _i8p, _i64p, C.c_int32]
    L.ssw_gpu_pool_align.argtypes = [C.c_void_p, _i8p, _i64p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Params),
                                     C.c_void_p, C.POINTER(_u32p), _i64p]
    L.ssw_gpu_pool_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(PoolStat)]
    return L


class CSsw(object):
    """Same surface as the reference's src/ssw_lib.py CSsw (sLibPath = directory holding libssw.so)."""

    def __init__(self, sLibPath=None):
        path = os.path.join(sLibPath, "libssw.so") if sLibPath else None
        self.ssw = load(path)
        self.ssw_init = self.ssw.ssw_init
        self.init_destroy = self.ssw.init_destroy
        self.ssw_align = self.ssw.ssw_align
        self.align_destroy = self.ssw.align_destroy
        self._ctx = None

    def align_many(self, lQueries, lTargets, mat, nOpen, nExt, nFlag, nMaskLen, nFilterScore=0, nFilterDist=0, device=0):
        """The loop of the reference's pyssw.py (src/pyssw.py:236-275: for every query ssw_init, for every target align_one) as ONE
        batch call.  lQueries / lTargets: number arrays as pyssw.to_int builds them (ctypes c_int8 arrays, numpy int8 arrays or lists);
        mat: the flat score matrix (lScore); nMaskLen < 0: len(query) / 2 per query (pyssw.py:243).  Returns [query][target] of the
        tuples align_one returns: (nScore, nScore2, nRefBeg, nRefEnd, nQryBeg, nQryEnd, nRefEnd2, nCigarLen, lCigar)."""
        if self._ctx is None:
            self._ctx = Context(device, self.ssw)
        ctx = self._ctx
        qs = [np.frombuffer(bytes(bytearray(x)), dtype=np.int8) if not isinstance(x, np.ndarray) else np.ascontiguousarray(x, dtype=np.int8) for x in
              ([(int(v) & 0xff) for v in q] if not isinstance(q, np.ndarray) else q for q in lQueries)]
        ts = [np.frombuffer(bytes(bytearray(x)), dtype=np.int8) if not isinstance(x, np.ndarray) else np.ascontiguousarray(x, dtype=np.int8) for x in
              ([(int(v) & 0xff) for v in t] if not isinstance(t, np.ndarray) else t for t in lTargets)]
        m = np.ascontiguousarray(np.array(list(mat), dtype=np.int8))
        n = int(round(len(m) ** 0.5))
        Q = ctx.upload(qs); T = ctx.upload(ts)
        try:
            res, cig = ctx.align_batch(Q, T, m, n, nOpen, nExt, nFlag, nFilterScore, nFilterDist, nMaskLen, 2)
        finally:
            Q.free(); T.free()
        out = []
        for qi in range(len(qs)):
            row = []
            for ti in range(len(ts)):
                g = res[qi, ti]
                k, o = int(g["cigarLen"]), int(g["cigar_off"])
                row.append((int(g["score1"]), int(g["score2"]), int(g["ref_begin1"]), int(g["ref_end1"]), int(g["read_begin1"]), int(g["read_end1"]),
                            int(g["ref_end2"]), k, [int(x) for x in cig[o:o + k]] if k > 0 else []))
            out.append(row)
        return out

    def close(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


class Seqs(object):
    def __init__(self, ctx, seqs):
        self.ctx = ctx
        self.count = len(seqs)
        off = np.zeros(self.count + 1, dtype=np.int64)
        for i, s in enumerate(seqs):
            off[i + 1] = off[i] + len(s)
        codes = (np.concatenate([np.asarray(s, dtype=np.int8) for s in seqs]) if self.count else
                 np.zeros(0, dtype=np.int8))
        codes = np.ascontiguousarray(codes, dtype=np.int8)
        self.lengths = np.diff(off)
        self.h = ctx.lib.ssw_gpu_seqs_upload(ctx.h, codes.ctypes.data_as(_i8p), off.ctypes.data_as(_i64p), self.count)
        if not self.h:
            raise RuntimeError("ssw_gpu_seqs_upload: " + ctx.error())

    def free(self):
        if self.h:
            self.ctx.lib.ssw_gpu_seqs_free(self.h)
            self.h = None


class Context(object):
    """One GPU (include/ssw_gpu.h ssw_gpu_ctx)."""

    def __init__(self, device=0, lib=None):
        self.lib = lib if lib is not None and not isinstance(lib, str) else load(lib)
        self._pinned = []
        self.device = device
        self.h = self.lib.ssw_gpu_open(device)
        if not self.h:
            raise RuntimeError("ssw_gpu_open: " + self.lib.ssw_gpu_last_error(None).decode())

    def error(self):
        return self.lib.ssw_gpu_last_error(self.h).decode()

    def set_exclusive(self):
        """this context has its device to itself: scratch budget sized for the whole HBM (ssw_gpu_set_budget_exclusive) -> bytes"""
        self.lib.ssw_gpu_set_budget_exclusive(self.h)
        return int(self.lib.ssw_gpu_get_budget(self.h))

    def upload(self, seqs):
        return Seqs(self, seqs)

    def align_batch(self, queries, targets, mat, n, gapO=3, gapE=1, flag=0, filters=0, filterd=0, maskLen=-1,
                    score_size=2, target_first=0, target_count=None, want_cigar=True, mark_mismatch=False, out=None):
        """-> (numpy record array [nq, nt] of RESULT_DTYPE, numpy uint32 CIGAR pool).  `out`: a result array to fill
        (e.g. page-locked, from result_array(): large database searches download at PCIe rate only into pinned pages)."""
        if target_count is None:
            target_count = targets.count - target_first
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        p = Params(mat.ctypes.data_as(_i8p), n, gapO, gapE, flag, filters, filterd, maskLen, score_size, 1 if mark_mismatch else 0)
        if out is None:
            res = np.zeros((queries.count, target_count), dtype=RESULT_DTYPE)
        else:
            res = out
            if res.dtype != RESULT_DTYPE or res.shape != (queries.count, target_count) or not res.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a C-contiguous [nq, nt] array of RESULT_DTYPE")
        pool = _u32p()
        words = C.c_int64(0)
        rc = self.lib.ssw_gpu_align_batch(self.h, queries.h, targets.h, target_first, target_count, C.byref(p),
                                          res.ctypes.data_as(C.c_void_p), C.byref(pool) if want_cigar else None,
                                          C.byref(words))
        if rc != 0:
            raise RuntimeError("ssw_gpu_align_batch: " + (self.lib.ssw_gpu_strerror(rc).decode() if rc == -2 else self.error()))
        if want_cigar and words.value > 0:
            cig = np.ctypeslib.as_array(pool, shape=(words.value,)).copy()
        else:
            cig = np.zeros(0, dtype=np.uint32)
        if want_cigar and pool:
            C.CDLL(None).free(pool)
        return res, cig

    def search_db(self, queries, targets, mat, n, gapO=3, gapE=1, maskLen=-1, score_size=2, chunk=0, on_chunk=None):
        """streamed database search (ssw_gpu_search_db): on_chunk(target_first, hits[nq, target_count] of HIT_DTYPE) is called for
        every chunk of targets while the device works on the next one (the array is only valid during the call; return a
        non-zero int to stop).  Without on_chunk the chunks are assembled: -> hits [nq, nt]."""
        assert HIT_DTYPE.itemsize == 16
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        p = Params(mat.ctypes.data_as(_i8p), n, gapO, gapE, 0, 0, 0, maskLen, score_size, 0)
        nq = queries.count
        whole = None if on_chunk is not None else np.zeros((nq, targets.count), dtype=HIT_DTYPE)
        err = []
        stopped = []        # what the caller's function returned to stop the search (any non-zero int, negative ones included)

        def cb(_user, tfirst, tcount, ptr):
            try:
                buf = (C.c_char * (nq * tcount * 16)).from_address(ptr)
                hits = np.frombuffer(buf, dtype=HIT_DTYPE).reshape(nq, tcount)
                if on_chunk is not None:
                    r = int(on_chunk(tfirst, hits) or 0)
                    if r:
                        stopped.append(r)
                    return r
                whole[:, tfirst:tfirst + tcount] = hits
                return 0
            except Exception as e:     # noqa: BLE001 -- an exception must not unwind through the C caller
                err.append(e)
                return -99

        rc = self.lib.ssw_gpu_search_db(self.h, queries.h, targets.h, C.byref(p), chunk, HITS_FN(cb), None)
        if err:
            raise err[0]
        if stopped and rc == stopped[0]:
            return rc               # "non-zero to stop, that value is returned" (include/ssw_gpu.h): not a library failure
        if rc == -2:
            raise RuntimeError("ssw_gpu_search_db: " + self.lib.ssw_gpu_strerror(rc).decode())
        if rc < 0:
            raise RuntimeError("ssw_gpu_search_db: " + self.error())
        return whole if on_chunk is None else rc

    def result_array(self, nq, nt):
        """[nq, nt] result records in page-locked host memory (freed with the context)"""
        nbytes = int(nq) * int(nt) * RESULT_DTYPE.itemsize
        p = self.lib.ssw_gpu_host_alloc(self.h, nbytes)
        if not p:
            raise RuntimeError("ssw_gpu_host_alloc: " + self.error())
        self._pinned.append(p)
        buf = (C.c_char * nbytes).from_address(p)
        return np.frombuffer(buf, dtype=RESULT_DTYPE).reshape(int(nq), int(nt))

    def timing(self):
        t = Timing()
        self.lib.ssw_gpu_last_timing_sized(self.h, C.byref(t), C.sizeof(t))     # sized: this mirror may be older than the library
        d = {k: getattr(t, k) for k, _ in Timing._fields_}
        d["fill_kernel"] = d["fill_kernel"].decode()
        return d

    def selftest_lanes(self):
        """diagnostic of libssw_hooks.so (include/ssw_gpu_diag.h); the product library does not export it"""
        out = np.zeros((16, 64), dtype=np.uint32)
        if self.lib.ssw_gpu_selftest_lanes(self.h, out.ctypes.data_as(_u32p)) != 0:
            raise RuntimeError("ssw_gpu_selftest_lanes: " + self.error())
        return out

    def valu_probe(self, blocks=4096, iters=2000):
        return float(self.lib.ssw_gpu_valu_probe(self.h, blocks, iters))

    def close(self):
        if self.h:
            for p in self._pinned:
                self.lib.ssw_gpu_host_free(self.h, p)
            self._pinned = []
            self.lib.ssw_gpu_close(self.h)
            self.h = None


def _pack(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    for i, x in enumerate(seqs):
        off[i + 1] = off[i] + len(x)
    codes = np.concatenate([np.asarray(x, dtype=np.int8) for x in seqs]) if len(seqs) else np.zeros(0, dtype=np.int8)
    return np.ascontiguousarray(codes, dtype=np.int8), off


class Pool(object):
    """Several devices, one batch (include/ssw_gpu.h ssw_gpu_pool): per-GPU work queues over host-resident reads."""

    def __init__(self, devices=None, lib=None):
        self.lib = lib if lib is not None and not isinstance(lib, str) else load(lib)
        if devices is None:
            self.h = self.lib.ssw_gpu_pool_open(None, 0)
        else:
            arr = (C.c_int * len(devices))(*devices)
            self.h = self.lib.ssw_gpu_pool_open(arr, len(devices))
        if not self.h:
            raise RuntimeError("ssw_gpu_pool_open: " + self.lib.ssw_gpu_last_error(None).decode())
        self.ntargets = 0

    def size(self):
        return int(self.lib.ssw_gpu_pool_size(self.h))

    def error(self):
        return self.lib.ssw_gpu_pool_last_error(self.h).decode()

    def set_targets(self, seqs):
        codes, off = _pack(seqs)
        if self.lib.ssw_gpu_pool_set_targets(self.h, codes.ctypes.data_as(_i8p), off.ctypes.data_as(_i64p), len(seqs)) != 0:
            raise RuntimeError("ssw_gpu_pool_set_targets: " + self.error())
        self.ntargets = len(seqs)

    def align(self, reads, mat, n, gapO=3, gapE=1, flag=0, filters=0, filterd=0, maskLen=-1, score_size=2, block=0,
              want_cigar=True, mark_mismatch=False, packed=None):
        """reads: list of code arrays, or packed=(codes int8, offsets int64).  -> (records [nq, nt], CIGAR pool)"""
        codes, off = packed if packed is not None else _pack(reads)
        nq = len(off) - 1
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        p = Params(mat.ctypes.data_as(_i8p), n, gapO, gapE, flag, filters, filterd, maskLen, score_size, 1 if mark_mismatch else 0)
        res = np.zeros((nq, self.ntargets), dtype=RESULT_DTYPE)
        pool = _u32p()
        words = C.c_int64(0)
        rc = self.lib.ssw_gpu_pool_align(self.h, codes.ctypes.data_as(_i8p), off.ctypes.data_as(_i64p), nq, block, 0, self.ntargets,
                                         C.byref(p), res.ctypes.data_as(C.c_void_p), C.byref(pool) if want_cigar else None, C.byref(words))
        if rc != 0:
            raise RuntimeError("ssw_gpu_pool_align: " + self.error())
        cig = np.ctypeslib.as_array(pool, shape=(words.value,)).copy() if want_cigar and words.value > 0 else np.zeros(0, dtype=np.uint32)
        if want_cigar and pool:
            C.CDLL(None).free(pool)
        return res, cig

    def stats(self):
        out = []
        for w in range(self.size()):
            st = PoolStat()
            self.lib.ssw_gpu_pool_stats(self.h, w, C.byref(st))
            out.append({k: getattr(st, k) for k, _ in PoolStat._fields_})
        return out

    def close(self):
        if self.h:
            self.lib.ssw_gpu_pool_close(self.h)
            self.h = None
