"""Shared helpers for the test-suite, bench.py and __graft_entry__.smoke().

Loads the TEST-ONLY checkers (oracle/libssw_oracle.so = this repo's CPU restatement,
oracle/_ref/libssw_ref.so = the unmodified reference built by oracle/Makefile) through
ctypes, and provides seeded synthetic workloads shaped like BASELINE.json's configs.
Nothing here is imported by the product library.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_SRC = "/root/reference/src"

i8p = C.POINTER(C.c_int8)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
i64p = C.POINTER(C.c_int64)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True, stdout=subprocess.DEVNULL)


class OrcResult(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2",
                 "cigarLen", "flag", "is_null", "used_word")]


class OrcEnd(C.Structure):
    _fields_ = [("score", C.c_int32), ("ref", C.c_int32), ("read", C.c_int32)]


# mirrors the reference's s_align (ssw.h:55-66)
class SAlign(C.Structure):
    _fields_ = [("score1", C.c_uint16), ("score2", C.c_uint16), ("ref_begin1", C.c_int32),
                ("ref_end1", C.c_int32), ("read_begin1", C.c_int32), ("read_end1", C.c_int32),
                ("ref_end2", C.c_int32), ("cigar", u32p), ("cigarLen", C.c_int32), ("flag", C.c_uint16)]


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "libssw_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.orc_align.argtypes = [C.c_int, i8p, C.c_int32, i8p, C.c_int32, C.c_int, i8p, C.c_int32, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int32, C.c_int32, C.POINTER(OrcResult), u32p, C.c_int32]
        L.orc_align.restype = None
        L.orc_striped.argtypes = [C.c_int, i8p, C.c_int, C.c_int32, i8p, C.c_int32, i8p, C.c_int32, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int32, C.POINTER(OrcEnd), i32p]
        L.orc_striped.restype = None
        L.orc_plain_fill.argtypes = [i8p, C.c_int32, C.c_int32, C.c_int, i8p, C.c_int32, C.c_int32, i8p, C.c_int32,
                                     C.c_int, C.c_int, C.c_int, i32p, C.POINTER(OrcEnd)]
        L.orc_plain_fill.restype = C.c_int32
        L.orc_plain_halo.argtypes = [C.c_int32, i8p, C.c_int32, C.c_int]
        L.orc_plain_halo.restype = C.c_int32
        L.orc_plain_colmax_tiled.argtypes = [i8p, C.c_int32, i8p, C.c_int32, C.c_int32, i8p, C.c_int32, C.c_int,
                                             C.c_int, C.c_int32, C.c_int32, i32p]
        L.orc_plain_colmax_tiled.restype = None
        L.orc_banded.argtypes = [i8p, i8p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_int32, i8p,
                                 C.c_int32, u32p, C.c_int32]
        L.orc_banded.restype = C.c_int32
        L.orc_mark_mismatch.argtypes = [C.c_int32, C.c_int32, C.c_int32, i8p, i8p, C.c_int32, u32p, C.c_int32, u32p,
                                        i32p]
        L.orc_mark_mismatch.restype = C.c_int32
        _oracle = L
    return _oracle


def ref_lib(required=False):
    """The unmodified reference (None when oracle/_ref was never built and cannot be built here)."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libssw_ref.so")
        if not os.path.exists(path) and os.path.exists(os.path.join(REF_SRC, "ssw.c")):
            build_oracle()
        if not os.path.exists(path):
            if required:
                raise RuntimeError("oracle/_ref/libssw_ref.so missing")
            return None
        L = C.CDLL(path)
        L.ssw_init.argtypes = [i8p, C.c_int32, i8p, C.c_int32, C.c_int8]
        L.ssw_init.restype = C.c_void_p
        L.init_destroy.argtypes = [C.c_void_p]
        L.ssw_align.argtypes = [C.c_void_p, i8p, C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int32,
                                C.c_int32]
        L.ssw_align.restype = C.POINTER(SAlign)
        L.align_destroy.argtypes = [C.POINTER(SAlign)]
        L.mark_mismatch.argtypes = [C.c_int32, C.c_int32, C.c_int32, i8p, i8p, C.c_int32, C.POINTER(u32p), i32p]
        L.mark_mismatch.restype = C.c_int32
        L.refwrap_sw_byte.argtypes = [i8p, C.c_int8, C.c_int32, i8p, C.c_int32, i8p, C.c_int32, C.c_uint8, C.c_uint8,
                                      C.c_uint8, C.c_uint8, C.c_int32, i32p]
        L.refwrap_sw_word.argtypes = [i8p, C.c_int8, C.c_int32, i8p, C.c_int32, i8p, C.c_int32, C.c_uint8, C.c_uint8,
                                      C.c_uint16, C.c_int32, i32p]
        L.refwrap_banded_sw.argtypes = [i8p, i8p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_int32,
                                        i8p, C.c_int32, u32p, C.c_int32]
        L.refwrap_banded_sw.restype = C.c_int32
        L.refwrap_bench.argtypes = [i8p, i64p, C.c_int32, i8p, C.c_int32, i8p, C.c_int32, C.c_uint8, C.c_uint8,
                                    C.c_uint8, C.c_uint16, C.c_int32, C.c_int32, C.c_int32, i32p]
        L.refwrap_bench.restype = C.c_double
        if hasattr(L, "refwrap_bench_hash"):     # (a prebuilt oracle/_ref of an older round lacks them)
            L.refwrap_bench_hash.argtypes = L.refwrap_bench.argtypes + [u32p]
            L.refwrap_bench_hash.restype = C.c_double
            L.refwrap_bench_db.argtypes = [i8p, i64p, C.c_int32, i8p, i64p, C.c_int32, i8p, C.c_int32, C.c_uint8, C.c_uint8,
                                           C.c_int32, C.c_int32, i32p]
            L.refwrap_bench_db.restype = C.c_double
        if hasattr(L, "refwrap_bench_dbx"):
            L.refwrap_bench_dbx.argtypes = [i8p, i64p, C.c_int32, i8p, i64p, C.c_int32, i8p, C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16,
                                            C.c_int32, C.c_int32, C.c_int32, i32p, u32p]
            L.refwrap_bench_dbx.restype = C.c_double
        _ref = L
    return _ref


# ---------------------------------------------------------------- scoring
def dna_matrix(match=2, mismatch=2):
    """5x5 matrix built as the reference CLI does (main.c:326-334): N scores 0."""
    m = np.zeros((5, 5), dtype=np.int8)
    for i in range(4):
        for j in range(4):
            m[i, j] = match if i == j else -mismatch
    return m.reshape(-1).copy()


# BLOSUM50 in the reference's 24-letter order ARNDCQEGHILKMFPSTWYVBZX* (main.c:43-69); regenerated
# here from the standard BLOSUM50 table, not copied: tests/test_oracle_vs_ref.py checks the values
# against the protein demo alignment of the reference build.
_B50_ROWS = """
 5 -2 -1 -2 -1 -1 -1  0 -2 -1 -2 -1 -1 -3 -1  1  0 -3 -2  0 -2 -1 -1 -5
-2  7 -1 -2 -4  1  0 -3  0 -4 -3  3 -2 -3 -3 -1 -1 -3 -1 -3 -1  0 -1 -5
-1 -1  7  2 -2  0  0  0  1 -3 -4  0 -2 -4 -2  1  0 -4 -2 -3  5  0 -1 -5
-2 -2  2  8 -4  0  2 -1 -1 -4 -4 -1 -4 -5 -1  0 -1 -5 -3 -4  6  1 -1 -5
-1 -4 -2 -4 13 -3 -3 -3 -3 -2 -2 -3 -2 -2 -4 -1 -1 -5 -3 -1 -3 -3 -1 -5
-1  1  0  0 -3  7  2 -2  1 -3 -2  2  0 -4 -1  0 -1 -1 -1 -3  0  4 -1 -5
-1  0  0  2 -3  2  6 -3  0 -4 -3  1 -2 -3 -1 -1 -1 -3 -2 -3  1  5 -1 -5
 0 -3  0 -1 -3 -2 -3  8 -2 -4 -4 -2 -3 -4 -2  0 -2 -3 -3 -4 -1 -2 -1 -5
-2  0  1 -1 -3  1  0 -2 10 -4 -3  0 -1 -1 -2 -1 -2 -3  2 -4  0  0 -1 -5
-1 -4 -3 -4 -2 -3 -4 -4 -4  5  2 -3  2  0 -3 -3 -1 -3 -1  4 -4 -3 -1 -5
-2 -3 -4 -4 -2 -2 -3 -4 -3  2  5 -3  3  1 -4 -3 -1 -2 -1  1 -4 -3 -1 -5
-1  3  0 -1 -3  2  1 -2  0 -3 -3  6 -2 -4 -1  0 -1 -3 -2 -3  0  1 -1 -5
-1 -2 -2 -4 -2  0 -2 -3 -1  2  3 -2  7  0 -3 -2 -1 -1  0  1 -3 -1 -1 -5
-3 -3 -4 -5 -2 -4 -3 -4 -1  0  1 -4  0  8 -4 -3 -2  1  4 -1 -4 -4 -1 -5
-1 -3 -2 -1 -4 -1 -1 -2 -2 -3 -4 -1 -3 -4 10 -1 -1 -4 -3 -3 -2 -1 -1 -5
 1 -1  1  0 -1  0 -1  0 -1 -3 -3  0 -2 -3 -1  5  2 -4 -2 -2  0  0 -1 -5
 0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  2  5 -3 -2  0  0 -1 -1 -5
-3 -3 -4 -5 -5 -1 -3 -3 -3 -3 -2 -3 -1  1 -4 -4 -3 15  2 -3 -5 -2 -1 -5
-2 -1 -2 -3 -3 -1 -2 -3  2 -1 -1 -2  0  4 -3 -2 -2  2  8 -1 -3 -2 -1 -5
 0 -3 -3 -4 -1 -3 -3 -4 -4  4  1 -3  1 -1 -3 -2  0 -3 -1  5 -3 -3 -1 -5
-2 -1  5  6 -3  0  1 -1  0 -4 -4  0 -3 -4 -2  0  0 -5 -3 -3  6  1 -1 -5
-1  0  0  1 -3  4  5 -2  0 -3 -3  1 -1 -4 -1  0 -1 -2 -2 -3  1  5 -1 -5
-1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -5
-5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5  1
"""


def blosum50():
    return np.array(_B50_ROWS.split(), dtype=np.int8)


AA_ORDER = "ARNDCQEGHILKMFPSTWYVBZX*"


def encode_dna(s):
    t = np.full(128, 4, dtype=np.int8)
    for ch, v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 3)):
        t[ord(ch)] = v
        t[ord(ch.lower())] = v
    return t[np.frombuffer(s.encode(), dtype=np.uint8)].copy()


def encode_aa(s):
    t = np.full(128, 23, dtype=np.int8)
    for i, ch in enumerate(AA_ORDER[:23]):
        t[ord(ch)] = i
        t[ord(ch.lower())] = i
    return t[np.frombuffer(s.encode(), dtype=np.uint8)].copy()


# ---------------------------------------------------------------- synthetic workloads
def random_ref(length, seed, n_codes=4, n_frac=0.0):
    rng = np.random.default_rng(seed)
    r = rng.integers(0, n_codes, size=length, dtype=np.int8)
    if n_frac > 0:
        r[rng.random(length) < n_frac] = 4
    return r


def mutate(seq, rng, sub=0.03, ins=0.005, dele=0.005, n_codes=4):
    out = []
    for b in seq:
        u = rng.random()
        if u < dele:
            continue
        if u < dele + ins:
            out.append(int(rng.integers(0, n_codes)))
        if rng.random() < sub:
            out.append(int((b + 1 + rng.integers(0, n_codes - 1)) % n_codes))
        else:
            out.append(int(b))
    return np.array(out, dtype=np.int8)


def sample_reads(ref, nreads, length, seed, sub=0.03, ins=0.005, dele=0.005, frac_random=0.05, n_codes=4,
                 fixed_len=True):
    """Reads sampled from `ref` at uniform offsets, mutated; `frac_random` of them fully random."""
    rng = np.random.default_rng(seed)
    reads = []
    for _ in range(nreads):
        if rng.random() < frac_random:
            reads.append(rng.integers(0, n_codes, size=length, dtype=np.int8))
            continue
        span = length + 16
        off = int(rng.integers(0, max(1, len(ref) - span)))
        r = mutate(ref[off:off + span], rng, sub, ins, dele, n_codes)
        if fixed_len:
            r = r[:length]
            if len(r) < length:
                r = np.concatenate([r, rng.integers(0, n_codes, size=length - len(r), dtype=np.int8)])
        reads.append(r.astype(np.int8))
    return reads


def pack_seqs(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    for i, s in enumerate(seqs):
        off[i + 1] = off[i] + len(s)
    codes = np.concatenate(seqs).astype(np.int8) if len(seqs) else np.zeros(0, dtype=np.int8)
    return np.ascontiguousarray(codes), off


# ---------------------------------------------------------------- checker front-ends
RES_FIELDS = ("score1", "score2", "ref_begin1", "ref_end1", "read_begin1", "read_end1", "ref_end2", "cigarLen", "flag")


def oracle_align(read, mat, n, ref, gapO, gapE, flag, filters, filterd, maskLen, score_size=2, model=0):
    """-> (dict of fields | None when the reference returns NULL, cigar list)"""
    L = oracle_lib()
    read = np.ascontiguousarray(read, dtype=np.int8)
    ref = np.ascontiguousarray(ref, dtype=np.int8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    r = OrcResult()
    cap = len(read) + len(ref) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    L.orc_align(model, _ptr(read, i8p), len(read), _ptr(mat, i8p), n, score_size, _ptr(ref, i8p), len(ref), gapO, gapE,
                flag, filters, filterd, maskLen, C.byref(r), _ptr(cig, u32p), cap)
    if r.is_null:
        return None, []
    d = {k: getattr(r, k) for k in RES_FIELDS}
    d["used_word"] = r.used_word
    return d, [int(x) for x in cig[:r.cigarLen]]


def ref_align(read, mat, n, ref, gapO, gapE, flag, filters, filterd, maskLen, score_size=2):
    L = ref_lib(required=True)
    read = np.ascontiguousarray(read, dtype=np.int8)
    ref = np.ascontiguousarray(ref, dtype=np.int8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    p = L.ssw_init(_ptr(read, i8p), len(read), _ptr(mat, i8p), n, score_size)
    a = L.ssw_align(p, _ptr(ref, i8p), len(ref), gapO, gapE, flag, filters, filterd, maskLen)
    if not a:
        L.init_destroy(p)
        return None, []
    s = a.contents
    d = {k: int(getattr(s, k)) for k in RES_FIELDS}
    cig = [int(s.cigar[i]) for i in range(s.cigarLen)] if s.cigarLen > 0 and s.cigar else []
    L.align_destroy(a)
    L.init_destroy(p)
    return d, cig


def cigar_str(cig):
    return "".join("%d%s" % (c >> 4, "MIDNSHP=X"[c & 0xf] if (c & 0xf) <= 8 else "M") for c in cig)
