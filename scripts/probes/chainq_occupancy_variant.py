#!/usr/bin/env python3
"""MEASUREMENT ONLY (wrong results): what would k_chainq gain from a third wavefront per SIMD?

The queue kernel runs two wavefronts per SIMD: a strip's score profile -- (n + 1) residues x C x 1 KiB, 18 KiB for DNA at 12 rows per
lane -- lets 8 one-wavefront workgroups share a CU's LDS, and hipcc takes 175 registers.  This builds a copy of csrc/ssw_kernels.hip whose
strip profile has FOUR entries (target code & 3: no N, no null columns -- wrong scores at the edges, the same instruction stream) and
whose queue kernel is held to 168 registers (amdgpu_waves_per_eu(3, 8)): 11 workgroups per CU.  Run config 4 with --flag 0 under it and
compare the fill phase (scripts/gpu_round3_r.sh).  The committed kernel source is not touched: the copy goes to build/, the library to
variants/libssw_chainq3.so.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "complete-striped-smith-waterman-library_amd")
src = open(os.path.join(PKG, "csrc", "ssw_kernels.hip")).read()

def sub(old, new, count=1):
    global src
    assert src.count(old) == count, (src.count(old), old)
    src = src.replace(old, new)

sub("	const int total = (n + 1) * C * GL * 4;\n	for (int w = first; w < total; w += nthreads) {\n		const int b = w / (C * GL * 4), rem = w - b * (C * GL * 4);\n		const int c = rem / (GL * 4), l",
    "	const int total = 4 * C * GL * 4;\n	for (int w = first; w < total; w += nthreads) {\n		const int b = w / (C * GL * 4), rem = w - b * (C * GL * 4);\n		const int c = rem / (GL * 4), l")
sub("	return (u32)code * (u32)PS;\n}", "	return (u32)(code & 3) * (u32)PS;\n}")
sub("""template <int R, bool CAPTURE, int FORM>
__global__ void __launch_bounds__(64) k_chainq(ssw_chainx_args a)""", """template <int R, bool CAPTURE, int FORM>
__global__ void __launch_bounds__(64) SSW_WAVES_PER_EU(3, 8) k_chainq(ssw_chainx_args a)""")
sub("""	const int tid = (int)threadIdx.x, l16 = tid;
	const u32 prof_bytes = (u32)(a.n + 1) * G::PSTRIDE;
	StripCtx x;
	x.prof = 0; x.ring = prof_bytes;""", """	const int tid = (int)threadIdx.x, l16 = tid;
	const u32 prof_bytes = 4u * G::PSTRIDE;
	StripCtx x;
	x.prof = 0; x.ring = prof_bytes;""")
sub("x.bout = x.bin + BND_RING_BYTES; x.nulloff = (u32)a.n * G::PSTRIDE; x.bmask = 31u;", "x.bout = x.bin + BND_RING_BYTES; x.nulloff = 3u * G::PSTRIDE; x.bmask = 31u;")
sub("#define X(r) case r: { const size_t ldsb = (size_t)(args.n + 1) * StripGeom<r, 64>::PSTRIDE + (capture ? QueueGeom<r, true>::EXTRA : QueueGeom<r, false>::EXTRA); \\",
    "#define X(r) case r: { const size_t ldsb = (size_t)4 * StripGeom<r, 64>::PSTRIDE + (capture ? QueueGeom<r, true>::EXTRA : QueueGeom<r, false>::EXTRA); \\")
sub("#define X(r) case r: { const size_t ldsb = (size_t)(n + 1) * StripGeom<r, 64>::PSTRIDE + (capture ? QueueGeom<r, true>::EXTRA : QueueGeom<r, false>::EXTRA); \\",
    "#define X(r) case r: { const size_t ldsb = (size_t)4 * StripGeom<r, 64>::PSTRIDE + (capture ? QueueGeom<r, true>::EXTRA : QueueGeom<r, false>::EXTRA); \\")
sub("		const size_t ldsb = (size_t)(n + 1) * ((size_t)((R + 3) / 4) * 1024) + (capture ? 2176 : 1856);",
    "		const size_t ldsb = (size_t)4 * ((size_t)((R + 3) / 4) * 1024) + (capture ? 2176 : 1856);")

os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
os.makedirs(os.path.join(PKG, "variants"), exist_ok=True)
out = os.path.join(PKG, "build", "ssw_kernels_chainq3.hip")
open(out, "w").write(src)
hipcc = "/opt/rocm/bin/hipcc"
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../include", "-Icsrc", "-c", out, "-o", "build/ssw_kernels_chainq3.o"] + sys.argv[1:],
               cwd=PKG, check=True)
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "variants/libssw_chainq3.so", "build/ssw_kernels_chainq3.o",
                "build/ssw_host.o", "build/ssw_pool.o", "build/ssw_cigar.o", "-lpthread"], cwd=PKG, check=True)
print("built variants/libssw_chainq3.so")
