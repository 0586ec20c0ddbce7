import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "complete-striped-smith-waterman-library_amd")
sys.path.insert(0, HERE)
sys.path.insert(0, PKG)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import sswutil
    return sswutil.oracle_lib()


@pytest.fixture(scope="session")
def reflib():
    import sswutil
    lib = sswutil.ref_lib()
    if lib is None:
        pytest.skip("oracle/_ref/libssw_ref.so not available (reference sources absent and no prebuilt copy)")
    return lib


@pytest.fixture(scope="session")
def emu_lib_path():
    """The REAL host driver + REAL kernel source on the CPU SIMT emulator (tests/emu) -- test-only."""
    import fcntl
    with open(os.path.join(HERE, "emu", ".build.lock"), "w") as lk:      # pytest-xdist workers build once, not against each other
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.run(["make", "-C", os.path.join(HERE, "emu"), "-s", "libssw_emu.so", "ssw_test_emu"], check=True)
    return os.path.join(HERE, "emu", "libssw_emu.so")


@pytest.fixture(scope="session")
def product_lib_path():
    path = os.path.join(PKG, "libssw.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", PKG, "-s"], check=True)
    return path


@pytest.fixture(scope="session")
def hooks_lib_path(product_lib_path):
    """libssw_hooks.so: the product's kernels object and host source, with the form-switching SSW_GPU_* test hooks compiled in
    (-DSSW_GPU_TEST_HOOKS).  Tests that force a kernel form through the environment load THIS library; libssw.so ignores those hooks."""
    path = os.path.join(PKG, "libssw_hooks.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", PKG, "-s", "libssw_hooks.so"], check=True)
    return path


@pytest.fixture(scope="session")
def gpu_hctx(hooks_lib_path):
    import ssw_amd
    lib = ssw_amd.load(hooks_lib_path)
    assert lib.ssw_gpu_has_test_hooks() == 1
    assert lib.ssw_gpu_device_count() > 0, "no HIP device: the GPU tests need a real MI355X"
    ctx = ssw_amd.Context(0, lib)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_ctx(product_lib_path):
    import ssw_amd
    lib = ssw_amd.load(product_lib_path)
    assert lib.ssw_gpu_device_count() > 0, "no HIP device: the GPU tests need a real MI355X"
    ctx = ssw_amd.Context(0, lib)
    yield ctx
    ctx.close()
