"""The "next" rows of SURVEY 8f that run on the device, checked directly against the reference (not only through the CLI):

  8f-2  k_prep   ASCII -> residue codes through the caller's table, reverse complement (reference src/main.c:84-116, 476-481, 504)
  8f-3  k_mark   mark_mismatch on the device (reference src/ssw.c:1019-1074): randomized against the compiled reference's own
                 mark_mismatch -- leading / trailing soft clips, I / D next to mismatches, all-mismatch runs
  8f-4  the Python front-end: ssw_amd.CSsw keeps the surface of the reference's src/ssw_lib.py CSsw; the flow of
        src/pyssw.py (to_int -> ssw_init -> align_one -> align_destroy) is executed through it; in the build container the
        reference's ssw_lib.py itself is imported in place and drives the (emulated) library.

Same helpers for the CPU emulator (small, `not gpu`) and the MI355X (`gpu`, >= 1000 CIGARs)."""
import ctypes as C
import os

import numpy as np
import pytest

import ssw_amd
from parity import make_reads
from sswutil import _ptr, dna_matrix, encode_dna, i8p, i32p, random_ref, ref_align, ref_lib, u32p, RES_FIELDS

HERE = os.path.dirname(os.path.abspath(__file__))
CLI_DIR = os.path.join(HERE, "golden", "cli")


@pytest.fixture(scope="module")
def ectx(emu_lib_path):
    ctx = ssw_amd.Context(0, ssw_amd.load(emu_lib_path))
    yield ctx
    ctx.close()


# ------------------------------------------------------------------------------------------------ 8f-3
def _mark_case(ctx, nreads, reflen, seed):
    R = ref_lib(required=True)
    rng = np.random.default_rng(seed)
    ref = random_ref(reflen, seed + 1, 4, 0.01)
    reads = make_reads(rng, ref, nreads, rng.integers(20, 260, size=nreads), 4, sub=0.10, ins=0.03, dele=0.03, frac_random=0.05)
    for i in range(nreads):      # clips on either side for two thirds of the reads; runs of mismatches in some
        lead = rng.integers(0, 4, size=int(rng.integers(1, 12)), dtype=np.int8) if i % 3 != 0 else np.zeros(0, dtype=np.int8)
        tail = rng.integers(0, 4, size=int(rng.integers(1, 12)), dtype=np.int8) if i % 3 != 1 else np.zeros(0, dtype=np.int8)
        r = reads[i].copy()
        if i % 5 == 0 and len(r) > 40:
            k = int(rng.integers(10, len(r) - 20)); r[k:k + 4] = (r[k:k + 4] + 2) % 4
        reads[i] = np.ascontiguousarray(np.concatenate([lead, r, tail]), dtype=np.int8)
    mat = dna_matrix(3, 2)
    Q = ctx.upload(reads); T = ctx.upload([ref])
    raw, rcig = ctx.align_batch(Q, T, mat, 5, 4, 1, 2, 0, 0, -1, 2)
    res, cig = ctx.align_batch(Q, T, mat, 5, 4, 1, 2, 0, 0, -1, 2, mark_mismatch=True)
    Q.free(); T.free()
    checked = ops = 0
    for i, rd in enumerate(reads):
        a, b = raw[i, 0], res[i, 0]
        n = int(a["cigarLen"])
        if n <= 0:
            assert int(b["cigarLen"]) <= 0
            continue
        # the reference's own function on the raw CIGAR (it frees and replaces a malloc'ed array)
        libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.free.argtypes = [C.c_void_p]
        buf = libc.malloc(4 * n)
        rawc = np.ascontiguousarray(rcig[int(a["cigar_off"]):int(a["cigar_off"]) + n], dtype=np.uint32)      # (kept in a name: a temporary's buffer may be gone before memmove reads it)
        C.memmove(buf, rawc.ctypes.data, 4 * n)
        pc = C.cast(buf, u32p); cl = C.c_int32(n)
        nm = R.mark_mismatch(int(a["ref_begin1"]), int(a["read_begin1"]), int(a["read_end1"]), _ptr(ref, i8p), _ptr(rd, i8p), len(rd), C.byref(pc), C.byref(cl))
        want = [int(pc[k]) for k in range(cl.value)]
        libc.free(C.cast(pc, C.c_void_p))
        got = [int(x) for x in cig[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigarLen"])]]
        assert int(b["edit_distance"]) == nm and got == want, (i, got, want)
        checked += 1; ops += len(want)
    return checked, ops


def test_device_mark_mismatch_vs_reference_emulated(ectx):
    checked, _ = _mark_case(ectx, 24, 700, 5)
    assert checked >= 18


@pytest.mark.gpu
def test_device_mark_mismatch_vs_reference_gpu(gpu_ctx):
    checked, ops = _mark_case(gpu_ctx, 1400, 40000, 6)
    assert checked >= 1000 and ops > 10 * checked


# ------------------------------------------------------------------------------------------------ 8f-2
def _prep_case(ctx, nseq, seed):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGTNacgtnUuRYKM-*", dtype=np.uint8)
    texts = [bytes(rng.choice(alphabet, size=int(L))) for L in rng.integers(0, 300, size=nseq)]
    off = np.zeros(nseq + 1, dtype=np.int64); off[1:] = np.cumsum([len(t) for t in texts])
    blob = b"".join(texts)
    table = np.full(128, 4, dtype=np.int8)
    for ch, v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 3)):       # the reference's nt_table (src/main.c:72-82)
        table[ord(ch)] = v; table[ord(ch.lower())] = v
    lib = ctx.lib
    h = lib.ssw_gpu_seqs_upload_ascii(ctx.h, blob, _ptr(off, C.POINTER(C.c_int64)), nseq, _ptr(table, i8p))
    assert h, ctx.error()
    got = np.zeros(len(blob), dtype=np.int8)
    assert lib.ssw_gpu_seqs_download(ctx.h, h, _ptr(got, i8p)) == 0
    want = table[np.frombuffer(blob, dtype=np.uint8) & 127]
    assert (got == want).all()
    # reverse complement: what src/main.c:95-116 + nt_table produce: reversed, A<->T, C<->G, everything else code 4
    hr = lib.ssw_gpu_seqs_revcomp(ctx.h, h)
    assert hr, ctx.error()
    rc = np.zeros(len(blob), dtype=np.int8)
    assert lib.ssw_gpu_seqs_download(ctx.h, hr, _ptr(rc, i8p)) == 0
    for i in range(nseq):
        seg = want[off[i]:off[i + 1]][::-1]
        exp = np.where(seg < 4, 3 - seg, seg)
        assert (rc[off[i]:off[i + 1]] == exp).all(), i
    hrr = lib.ssw_gpu_seqs_revcomp(ctx.h, hr)
    back = np.zeros(len(blob), dtype=np.int8)
    assert lib.ssw_gpu_seqs_download(ctx.h, hrr, _ptr(back, i8p)) == 0 and (back == want).all()      # involution
    for x in (h, hr, hrr):
        lib.ssw_gpu_seqs_free(x)


def test_device_sequence_preparation_emulated(ectx):
    _prep_case(ectx, 40, 1)


@pytest.mark.gpu
def test_device_sequence_preparation_gpu(gpu_ctx):
    _prep_case(gpu_ctx, 5000, 2)


# ------------------------------------------------------------------------------------------------ 8f-4
def _pyssw_flow(w, ct_mod, pairs, mat, n):
    """the calls of reference src/pyssw.py main(): to_int, ssw_init, align_one (flag 1, maskLen len/2 or 15), destroy"""
    out = []
    for read, ref in pairs:
        q = (ct_mod.c_int8 * len(read))(*[int(x) for x in read])
        r = (ct_mod.c_int8 * len(ref))(*[int(x) for x in ref])
        m = (ct_mod.c_int8 * len(mat))(*[int(x) for x in mat])
        prof = w.ssw_init(q, ct_mod.c_int32(len(read)), m, n, 2)
        mask = len(read) // 2 if len(read) > 30 else 15
        res = w.ssw_align(prof, r, ct_mod.c_int32(len(ref)), 3, 1, 1, 0, 0, mask)
        c = res.contents
        out.append(((c.nScore, c.nScore2, c.nRefBeg, c.nRefEnd, c.nQryBeg, c.nQryEnd, c.nRefEnd2, c.nCigarLen),
                    [c.sCigar[i] for i in range(c.nCigarLen)], mask))
        w.align_destroy(res)
        w.init_destroy(prof)
    return out


def _fastq_seqs(path):
    lines = open(path).read().split("\n")
    return [lines[i + 1] for i in range(0, len(lines) - 3, 4) if lines[i].startswith("@")]


def _wrapper_pairs():
    pairs = [(encode_dna("CTGAGCCGGTAAATC"), encode_dna("CAGCCTTTCTGACCCGGAAATCAAAATAGGCACAACAAA"))]      # example.c's pair
    targets = [encode_dna(s) for s in _fastq_seqs(os.path.join(CLI_DIR, "target.fastq"))]
    queries = [encode_dna(s) for s in _fastq_seqs(os.path.join(CLI_DIR, "query.fastq"))]
    pairs += [(q, t) for q in queries for t in targets]                                                    # BASELINE config 1
    return pairs


def _check_wrapper(w, ct_mod):
    mat = dna_matrix(2, 2)
    pairs = _wrapper_pairs()
    got = _pyssw_flow(w, ct_mod, pairs, mat, 5)
    assert got[0][0] == (21, 8, 8, 21, 0, 14, 4, 3) and got[0][1] == [144, 17, 80]
    if ref_lib() is not None:
        for (read, ref), (fields, cig, mask) in zip(pairs, got):
            exp, ecig = ref_align(read, mat, 5, ref, 3, 1, 1, 0, 0, mask)
            assert fields == tuple(exp[k] for k in RES_FIELDS[:8]) and cig == ecig
    return len(pairs)


def test_reference_python_wrapper_runs_alignments_on_the_emulated_library(emu_lib_path, tmp_path):
    """the reference's src/ssw_lib.py, imported from where it lies (build container only), loads `libssw.so` from a directory:
    give it one holding a link to the emulated library and run pyssw.py's call sequence through it"""
    import importlib.util
    src = "/root/reference/src/ssw_lib.py"
    if not os.path.exists(src):
        pytest.skip("reference sources not present")
    os.symlink(emu_lib_path, os.path.join(tmp_path, "libssw.so"))
    spec = importlib.util.spec_from_file_location("ref_ssw_lib_run", src)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    w = mod.CSsw(str(tmp_path))
    assert _check_wrapper(w, C) >= 5


def test_our_python_wrapper_keeps_the_reference_surface_emulated(emu_lib_path):
    w = ssw_amd.CSsw(os.path.dirname(emu_lib_path)) if os.path.basename(emu_lib_path) == "libssw.so" else None
    if w is None:      # the emulated library has another file name: bind the class by hand
        w = ssw_amd.CSsw.__new__(ssw_amd.CSsw)
        w.ssw = ssw_amd.load(emu_lib_path)
        w.ssw_init, w.init_destroy, w.ssw_align, w.align_destroy = w.ssw.ssw_init, w.ssw.init_destroy, w.ssw.ssw_align, w.ssw.align_destroy
    assert _check_wrapper(w, C) >= 5


@pytest.mark.gpu
def test_python_wrapper_alignments_gpu(product_lib_path):
    """example pair + BASELINE config 1 through the Python front-end (ssw_amd.CSsw = the reference's CSsw surface) on the MI355X"""
    w = ssw_amd.CSsw(os.path.dirname(product_lib_path))
    assert _check_wrapper(w, C) >= 5
