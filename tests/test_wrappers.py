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
