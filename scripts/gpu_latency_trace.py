import ctypes as C, os, sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'complete-striped-smith-waterman-library_amd')
import ssw_amd
from sswutil import SAlign, dna_matrix, i8p, random_ref, sample_reads
lib = ssw_amd.load()
lib.ssw_init.argtypes = [i8p, C.c_int32, i8p, C.c_int32, C.c_int8]; lib.ssw_init.restype = C.c_void_p
lib.ssw_align.argtypes = [C.c_void_p, i8p, C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int32, C.c_int32]; lib.ssw_align.restype = C.POINTER(SAlign)
lib.align_destroy.argtypes = [C.POINTER(SAlign)]; lib.init_destroy.argtypes = [C.c_void_p]
mat = dna_matrix(2, 2)
ref = random_ref(1_000_000, 1, 4)
reads = sample_reads(ref, 12, 150, seed=5)
for flag in (0, 2):
    for i, r in enumerate(reads):
        if i == 8: os.environ["SSW_GPU_CALL_TRACE"] = "1"
        r = np.ascontiguousarray(r)
        p = lib.ssw_init(r.ctypes.data_as(i8p), len(r), mat.ctypes.data_as(i8p), 5, 2)
        a = lib.ssw_align(p, ref.ctypes.data_as(i8p), len(ref), 3, 1, flag, 0, 0, 75)
        lib.align_destroy(a); lib.init_destroy(p)
    os.environ.pop("SSW_GPU_CALL_TRACE", None)
    print("---- flag", flag, file=sys.stderr)
