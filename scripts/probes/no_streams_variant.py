#!/usr/bin/env python3
"""MEASUREMENT ONLY (wrong results): a copy of csrc/ssw_kernels.hip in which k_fill keeps no per-column maxima.

Each lane accumulates the maxima of its own rows over the 16 steps of a block (two accumulators, for the two padding rules, that follow
the column frame with one add per step) instead of handing the column maximum down the chain; nothing is parked in LDS, and per 16
steps the 16 lanes' accumulators are reduced to one value per stream that lane 0 stores.  This is the instruction stream a design
without per-column streams ("skewed block maxima" + window passes that regenerate the few columns the reduction needs)
would run in the fill; the durations of k_fill under rocprofv3 --kernel-trace bound what that design could gain.  The committed kernel
source is not touched: the patched copy goes to build/ and is linked into variants/libssw_nostreams.so.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "complete-striped-smith-waterman-library_amd")
src = open(os.path.join(PKG, "csrc", "ssw_kernels.hip")).read()

def sub(old, new, count=1):
    global src
    assert src.count(old) == count, (src.count(old), old)
    src = src.replace(old, new)

# 1. the step loop of k_fill: the column maximum no longer travels down the chain, nothing is parked
sub("""			u32 cm = xl_row_shr1_zero(cmout);      /* this column's maximum of the rows above */
			chain_rows<R, true, FORM>(sc, H, E, hsave, f, cm, ck, gO, gE, fl);
			hsave = hin; Hlast = H[R - 1]; Fout = f; cmout = cm;
			if (G::TAP == 15) {
				if (l16 == 15) { lds_st32(lds, ob16 + 4u * j, cm); lds_st32(lds, ob8 + 4u * j, ck); }
			} else {
				if (l16 == 15) lds_st32(lds, ob16 + 4u * j, cm);
				if (l16 == G::TAP) lds_st32(lds, ob8 + 4u * j, ck);
			}
""", """			accA += gE; accB += gE;
			chain_rows_fr<R, 0, G::K8>(sc, H, E, hsave2 = hsave, f, accA, gO, gE, fl);
			chain_rows_fr<R, G::K8, R>(sc, H, E, hsave2, f, accB, gO, gE, fl);
			hsave = hin; Hlast = H[R - 1]; Fout = f;
""")
sub("	u32 Hlast = zero0, Fout = zero0, cmout = zero0, ck = 0, hsave = zero0;\n	const u32 lane_prof = (u32)l16 * 16u;\n	const u32 gO = FR ? a.gapO2 - a.gapE2 : a.gapO2;      /* frame form: gapO - gapE */",
    "	u32 Hlast = zero0, Fout = zero0, cmout = zero0, ck = 0, hsave = zero0, hsave2, accA = zero0, accB = zero0;\n	const u32 lane_prof = (u32)l16 * 16u;\n	const u32 gO = FR ? a.gapO2 - a.gapE2 : a.gapO2;")
# 2. per 16 steps: the lanes' accumulators -> one value per stream
sub("		if (s0 >= 32) fill_flush16<R, FORM>(lds, out16, out8, s0 - 32, l16, store_from, ncols, o16, o8, g16, g8, a.fr_base, a.fr_kmask, gapEi);   /* columns [s0-32, s0-16) are complete in the out rings */",
    """		if (s0 >= 16) {
			const u32 ph = fl - gE;
			u32 m16 = pk_max(accA, accB) - ph, m8 = accA - ph;
			m16 = pk_max(m16, xl_row_ror<1>(m16)); m8 = pk_max(m8, xl_row_ror<1>(m8));
			m16 = pk_max(m16, xl_row_ror<2>(m16)); m8 = pk_max(m8, xl_row_ror<2>(m8));
			m16 = pk_max(m16, xl_row_ror<4>(m16)); m8 = pk_max(m8, xl_row_ror<4>(m8));
			m16 = pk_max(m16, xl_row_ror<8>(m16)); m8 = pk_max(m8, xl_row_ror<8>(m8));
			if (l16 == 0 && s0 - 16 < ncols) { o16[(s0 - 16) >> 4] = m16; o8[(s0 - 16) >> 4] = m8; }
			accA = ph; accB = ph;
		}""")
sub("""	for (int base = nsteps - 32; base < nsteps; base += 16)
		if (base >= 0) fill_flush16<R, FORM>(lds, out16, out8, base, l16, store_from, ncols, o16, o8, g16, g8, a.fr_base, a.fr_kmask, gapEi);
}""", """	if (l16 == 0) { o16[nsteps >> 4] = accA; o8[nsteps >> 4] = accB; }
}""")
# the renormalisation also moves the accumulators
sub("			Hlast -= k; Fout -= k; cmout -= k; hsave -= k; fl -= k;\n		}\n		{   /* stage target columns",
    "			Hlast -= k; Fout -= k; cmout -= k; hsave -= k; fl -= k; accA -= k; accB -= k;\n		}\n		{   /* stage target columns")

os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
os.makedirs(os.path.join(PKG, "variants"), exist_ok=True)
out = os.path.join(PKG, "build", "ssw_kernels_nostreams.hip")
open(out, "w").write(src)
hipcc = "/opt/rocm/bin/hipcc"
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../include", "-Icsrc", "-c", out, "-o", "build/ssw_kernels_nostreams.o"] + sys.argv[1:],
               cwd=PKG, check=True)
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "variants/libssw_nostreams.so", "build/ssw_kernels_nostreams.o",
                "build/ssw_host.o", "build/ssw_pool.o", "build/ssw_cigar.o", "-lpthread"], cwd=PKG, check=True)
print("built variants/libssw_nostreams.so")
