"""SURVEY 8f-4, wrapper parity: the reference's OWN front-ends -- ssw_test (src/main.c), example.c and the C++ wrapper
(ssw_cpp.cpp + example.cpp) -- are compiled from where they lie under /root/reference by oracle/Makefile (`dropin`) and
linked against OUR libssw.so; nothing of them is copied into the repository (oracle/_ref is git-ignored and travels to
the GPU box prebuilt).  On the GPU they must print, byte for byte, what the all-reference builds print
(tests/golden/cli/*.stdout and *.example_stdout, produced by tests/golden/make_golden.py).  This is BASELINE config 1
(`ssw_test -c demo/target.fastq demo/query.fastq`) executed by the unmodified reference CLI on the MI355X path."""
import glob
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFDIR = os.path.join(ROOT, "oracle", "_ref")
CLI_DIR = os.path.join(HERE, "golden", "cli")
CASES = sorted(os.path.basename(p)[:-len(".args")] for p in glob.glob(os.path.join(CLI_DIR, "*.args")))
FRONT_ENDS = ("ssw_test_dropin", "example_c_dropin", "example_cpp_dropin")


def _exe(name):
    p = os.path.join(REFDIR, name)
    if not os.path.exists(p):
        pytest.skip("%s not built (oracle/Makefile dropin needs /root/reference and libssw.so)" % name)
    return p


def test_reference_front_ends_link_against_our_library(product_lib_path):
    """they resolve libssw.so to the in-tree product library and nothing of the reference's ssw.c is linked in"""
    for name in FRONT_ENDS:
        exe = _exe(name)
        out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if "libssw.so" in l]
        assert line and os.path.realpath(line[0].split("=>")[1].split("(")[0].strip()) == os.path.realpath(product_lib_path), out
        syms = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
        assert "ssw_align" in syms and "ssw_init" in syms          # imported, not compiled in


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_reference_ssw_test_on_our_library(name):
    exe = _exe("ssw_test_dropin")
    args = open(os.path.join(CLI_DIR, name + ".args")).read().split()
    r = subprocess.run([exe] + args, cwd=CLI_DIR, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == open(os.path.join(CLI_DIR, name + ".stdout")).read(), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["example_c", "example_cpp"])
def test_reference_examples_on_our_library(name):
    exe = _exe(name + "_dropin")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == open(os.path.join(CLI_DIR, name + ".example_stdout")).read()


def test_reference_python_wrapper_binds_our_library(product_lib_path):
    """ssw_lib.py (the reference's ctypes wrapper, used by pyssw.py) is imported from where it lies -- build container
    only -- and pointed at the directory of OUR libssw.so: every function it declares resolves, and its result struct has
    the layout of include/ssw.h's s_align.  (No alignment is run here: that needs a GPU and the file cannot travel.)"""
    import ctypes as C
    import importlib.util
    src = "/root/reference/src/ssw_lib.py"
    if not os.path.exists(src):
        pytest.skip("reference sources not present")
    spec = importlib.util.spec_from_file_location("ref_ssw_lib", src)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    w = mod.CSsw(os.path.dirname(product_lib_path))
    for fn in ("ssw_init", "init_destroy", "ssw_align", "align_destroy"):
        assert getattr(w, fn) is not None
    import ssw_amd
    ours = [(n, getattr(ssw_amd.CAlignRes, n).offset) for n, _ in ssw_amd.CAlignRes._fields_]
    theirs = [(n, getattr(mod.CAlignRes, n).offset) for n, _ in mod.CAlignRes._fields_]
    assert [o for _, o in ours[:len(theirs)]] == [o for _, o in theirs]
    assert C.sizeof(mod.CAlignRes) <= C.sizeof(ssw_amd.CAlignRes)


# ---- batched bindings: what the wrappers' per-read loops compute, as one batch call ----
def _pyssw_batch_check(lib_path, nreads, rlen, reflen, flags=(0, 2, 1)):
    """ssw_amd.CSsw.align_many against the call sequence of the reference's pyssw.py (to_int, ssw_init, align_one = ssw_align +
    field reads + align_destroy) executed pair by pair on the SAME library, and against the reference itself where oracle/_ref is there"""
    import ctypes as C
    import numpy as np
    import ssw_amd
    from sswutil import dna_matrix, random_ref, sample_reads, ref_align
    w = ssw_amd.CSsw(os.path.dirname(lib_path)) if os.path.basename(lib_path) == "libssw.so" else ssw_amd.CSsw()
    if os.path.basename(lib_path) != "libssw.so":
        w.ssw = ssw_amd.load(lib_path); w.ssw_init = w.ssw.ssw_init; w.init_destroy = w.ssw.init_destroy; w.ssw_align = w.ssw.ssw_align; w.align_destroy = w.ssw.align_destroy
    ref = random_ref(reflen, 5, 4, 0.002)
    reads = sample_reads(ref, nreads, rlen, seed=9)
    refs = [ref, np.ascontiguousarray(ref[reflen // 3:reflen // 3 + reflen // 2])]
    mat = dna_matrix(2, 2)
    for nFlag in flags:
        got = w.align_many(reads, refs, [int(x) for x in mat], 3, 1, nFlag, -1)
        for qi, rd in enumerate(reads):
            qNum = (C.c_int8 * len(rd))(*[int(x) for x in rd])
            mNum = (C.c_int8 * 25)(*[int(x) for x in mat])
            prof = w.ssw_init(qNum, C.c_int32(len(rd)), mNum, 5, 2)
            nMask = len(rd) // 2
            for ti, rf in enumerate(refs):
                rNum = (C.c_int8 * len(rf))(*[int(x) for x in rf])
                res = w.ssw_align(prof, rNum, C.c_int32(len(rf)), 3, 1, nFlag, 0, 0, nMask)
                c = res.contents
                one = (c.nScore, c.nScore2, c.nRefBeg, c.nRefEnd, c.nQryBeg, c.nQryEnd, c.nRefEnd2, c.nCigarLen, [c.sCigar[i] for i in range(c.nCigarLen)])
                w.align_destroy(res)
                assert got[qi][ti] == one, (nFlag, qi, ti, got[qi][ti], one)
                if qi < 6:
                    exp, ecig = ref_align(rd, mat, 5, rf, 3, 1, nFlag, 0, 0, nMask, 2) if _have_ref() else (None, None)
                    if exp is not None:
                        assert one[:7] == (exp["score1"], exp["score2"], exp["ref_begin1"], exp["ref_end1"], exp["read_begin1"], exp["read_end1"], exp["ref_end2"]) and one[8] == ecig
            w.init_destroy(prof)
    w.close()


def _have_ref():
    from sswutil import ref_lib
    return ref_lib() is not None


def test_pyssw_style_batch_equals_the_per_pair_loop_emulated(emu_lib_path):
    _pyssw_batch_check(emu_lib_path, 4, 50, 260, flags=(2,))


@pytest.mark.gpu
def test_pyssw_style_batch_equals_the_per_pair_loop_gpu(product_lib_path):
    _pyssw_batch_check(product_lib_path, 60, 150, 20000)


def test_cpp_batch_aligner_equals_reference_aligner_emulated(emu_lib_path, tmp_path):
    """include/ssw_gpu_cpp.h BatchAligner vs the reference's StripedSmithWaterman::Aligner (its ssw_cpp.cpp compiled from where it lies),
    both on the emulated library: every Alignment field, cigar vector / string, return flag, three filter settings"""
    if not os.path.exists("/root/reference/src/ssw_cpp.cpp"):
        pytest.skip("reference sources not present")
    exe = str(tmp_path / "batch_cpp_check_emu")
    emu_dir = os.path.dirname(emu_lib_path)
    subprocess.run(["g++", "-O2", "-std=c++11", "-include", os.path.join(ROOT, "include", "ssw.h"), "-I/root/reference/src", "-I" + os.path.join(ROOT, "include"),
                    "/root/reference/src/ssw_cpp.cpp", os.path.join(HERE, "cpp", "batch_check.cpp"), "-o", exe, "-L" + emu_dir, "-lssw_emu", "-lm",
                    "-Wl,-rpath," + emu_dir], check=True)
    r = subprocess.run([exe, "16", "70", "600"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_cpp_batch_aligner_equals_reference_aligner_gpu():
    exe = _exe("batch_cpp_check")
    r = subprocess.run([exe, "3000", "150", "200000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok 9000"), r.stdout[-2000:] + r.stderr[-2000:]
