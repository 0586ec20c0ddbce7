// frame_form_probe.hip -- round 3: VALU cost of one chain step (10 rows x 2 packed queries) in
//   A  the f16 form of rounds 1-2 (7.5 VOP3P instructions per row, all 4-cycle)
//   B  the "column frame" int16 form: values carry + floor(column), so E needs no decrement and the three add/sub of a
//      row are plain 32-bit adds on the packed pair (no carry can cross the halves) -- 2-cycle VOP2 -- next to 3.5 packed maxima
// plus the issue cost of the individual instructions involved.  Build: hipcc --offload-arch=gfx950 -O3 frame_form_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define DEV __device__ __forceinline__
DEV u32 max3nn(u32 a, u32 b, u32 c) { return __builtin_bit_cast(u32, __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)), __builtin_bit_cast(f16x2, c))); }
DEV u32 pkmax(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
DEV u32 shr1(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x111, 0xf, 0xf, true); }
DEV void pkf_cell2(u32 d0, u32 s0, u32 d1, u32 s1, u32& E0, u32& E1, u32& f, u32& cm, u32& h0, u32& h1, u32 nO, u32 nE)
{
	u32 t;
	asm("v_pk_add_f16 %[h0], %[d0], %[s0] clamp\n\t"
	    "v_pk_maximum3_f16 %[h0], %[h0], %[E0], %[f]\n\t"
	    "v_pk_add_f16 %[t], %[h0], %[nO] clamp\n\t"
	    "v_pk_add_f16 %[E0], %[E0], %[nE] clamp\n\t"
	    "v_pk_add_f16 %[f], %[f], %[nE] clamp\n\t"
	    "v_pk_max_f16 %[E0], %[E0], %[t]\n\t"
	    "v_pk_max_f16 %[f], %[f], %[t]\n\t"
	    "v_pk_add_f16 %[h1], %[d1], %[s1] clamp\n\t"
	    "v_pk_maximum3_f16 %[h1], %[h1], %[E1], %[f]\n\t"
	    "v_pk_add_f16 %[t], %[h1], %[nO] clamp\n\t"
	    "v_pk_add_f16 %[E1], %[E1], %[nE] clamp\n\t"
	    "v_pk_add_f16 %[f], %[f], %[nE] clamp\n\t"
	    "v_pk_max_f16 %[E1], %[E1], %[t]\n\t"
	    "v_pk_max_f16 %[f], %[f], %[t]\n\t"
	    "v_pk_maximum3_f16 %[cm], %[cm], %[h0], %[h1]"
	    : [h0] "=&v"(h0), [h1] "=&v"(h1), [t] "=&v"(t), [E0] "+v"(E0), [E1] "+v"(E1), [f] "+v"(f), [cm] "+v"(cm)
	    : [d0] "v"(d0), [s0] "v"(s0), [d1] "v"(d1), [s1] "v"(s1), [nO] "v"(nO), [nE] "v"(nE));
}
constexpr int R = 10;
// A: f16 form step
__global__ void __launch_bounds__(256) k_formA(u32* sink, int steps, u32 seed)
{
	const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
	u32 H[R], E[R], S[R];
	for (int r = 0; r < R; ++r) { H[r] = 0; E[r] = 0; S[r] = ((gid * 2654435761u + r * 40503u) & 0x03ff03ffu) ^ seed; }
	u32 Hlast = 0, Fout = 0, cmout = 0, hsave = 0, acc = 0;
	const u32 nO = 0x96009600u, nE = 0x90009000u;
	for (int s = 0; s < steps; ++s) {
		const u32 hin = shr1(Hlast);
		u32 f = shr1(Fout), cm = shr1(cmout), d = hsave;
#pragma unroll
		for (int r = 0; r < R; r += 2) {
			const u32 d1 = H[r], hold = H[r + 1];
			u32 h0, h1;
			pkf_cell2(d, S[r], d1, S[r + 1], E[r], E[r + 1], f, cm, h0, h1, nO, nE);
			H[r] = h0; H[r + 1] = h1; d = hold;
		}
		hsave = hin; Hlast = H[R - 1]; Fout = f; cmout = cm; acc ^= cm;
	}
	sink[gid] = acc ^ Hlast;
}
// B: column-frame int16 form step.  ADDS 0: the add/sub of a row are 32-bit adds on the packed pair; 1: v_pk_add_u16 / v_pk_sub_u16;
//    2: as 0, one asm block per row pair (the order is fixed by hand)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <int ADDS> DEV u32 padd(u32 a, u32 b) { return ADDS == 1 ? __builtin_bit_cast(u32, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)) : a + b; }
template <int ADDS> DEV u32 psub(u32 a, u32 b) { return ADDS == 1 ? __builtin_bit_cast(u32, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)) : a - b; }
DEV void cell2_asm(u32 d0, u32 s0, u32 d1, u32 s1, u32& E0, u32& E1, u32& f, u32& cm, u32& h0, u32& h1, u32 c1, u32 gE, u32 fl)
{
	u32 t;
	asm("v_add_u32 %[h0], %[d0], %[s0]\n\t"
	    "v_add_u32 %[h1], %[d1], %[s1]\n\t"
	    "v_pk_maximum3_f16 %[h0], %[h0], %[E0], %[f]\n\t"
	    "v_sub_u32 %[t], %[h0], %[c1]\n\t"
	    "v_pk_max_i16 %[f], %[f], %[t]\n\t"
	    "v_pk_maximum3_f16 %[E0], %[E0], %[t], %[fl]\n\t"
	    "v_sub_u32 %[f], %[f], %[gE]\n\t"
	    "v_pk_maximum3_f16 %[h1], %[h1], %[E1], %[f]\n\t"
	    "v_sub_u32 %[t], %[h1], %[c1]\n\t"
	    "v_pk_max_i16 %[f], %[f], %[t]\n\t"
	    "v_pk_maximum3_f16 %[E1], %[E1], %[t], %[fl]\n\t"
	    "v_sub_u32 %[f], %[f], %[gE]\n\t"
	    "v_pk_maximum3_f16 %[cm], %[cm], %[h0], %[h1]"
	    : [h0] "=&v"(h0), [h1] "=&v"(h1), [t] "=&v"(t), [E0] "+v"(E0), [E1] "+v"(E1), [f] "+v"(f), [cm] "+v"(cm)
	    : [d0] "v"(d0), [s0] "v"(s0), [d1] "v"(d1), [s1] "v"(s1), [c1] "v"(c1), [gE] "v"(gE), [fl] "v"(fl));
}
template <int ADDS> __global__ void __launch_bounds__(256) k_formB(u32* sink, int steps, u32 seed, u32 base, u32 gE, u32 c1)
{
	const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
	u32 H[R], E[R], S[R];
	asm volatile("" : "+v"(base), "+v"(gE), "+v"(c1));
	for (int r = 0; r < R; ++r) { H[r] = base; E[r] = base; S[r] = (((gid * 2654435761u + r * 40503u) & 0x00030003u) ^ (seed & 0x00010001u)); }
	u32 Hlast = base, Fout = base, cmout = 0, hsave = base, acc = 0, fl = base + gE + (threadIdx.x & 15);
#pragma unroll 2
	for (int s = 0; s < steps; ++s) {
		u32 hin = shr1(Hlast);
		hin = hin > fl ? hin : fl;            /* lane 0 of the chain: the zero the DPP move fills in becomes the floor (u32 max is exact here) */
		u32 f = shr1(Fout), cm = shr1(cmout), d = hsave;
		fl = padd<ADDS>(fl, gE);
#pragma unroll
		for (int r = 0; r < R; r += 2) {
			const u32 hold0 = H[r], hold1 = H[r + 1];
			if (ADDS == 2) {
				u32 h0, h1;
				cell2_asm(d, S[r], hold0, S[r + 1], E[r], E[r + 1], f, cm, h0, h1, c1, gE, fl);
				H[r] = h0; H[r + 1] = h1;
			} else {
				const u32 h0 = max3nn(padd<ADDS>(d, S[r]), E[r], f);
				const u32 t0 = psub<ADDS>(h0, c1);
				E[r] = max3nn(E[r], t0, fl);
				f = psub<ADDS>(pkmax(f, t0), gE);
				const u32 h1 = max3nn(padd<ADDS>(hold0, S[r + 1]), E[r + 1], f);
				const u32 t1 = psub<ADDS>(h1, c1);
				E[r + 1] = max3nn(E[r + 1], t1, fl);
				f = psub<ADDS>(pkmax(f, t1), gE);
				cm = max3nn(cm, h0, h1);
				H[r] = h0; H[r + 1] = h1;
			}
			d = hold1;
		}
		hsave = hin; Hlast = H[R - 1]; Fout = f; cmout = cm; acc ^= cm;
		if ((s & 255) == 255) {   /* renormalisation */
			const u32 k = gE << 8;
#pragma unroll
			for (int r = 0; r < R; ++r) { H[r] -= k; E[r] -= k; }
			hsave -= k; Hlast -= k; Fout -= k; cmout -= k; fl -= k;
		}
	}
	sink[gid] = acc ^ Hlast;
}
// dependent mixes, as one asm block: (a) add, max3, sub, max, sub, max3 = one row of form B;  (b) the same with packed adds
template <int MODE> DEV u32 mixop(u32 x, u32 g)
{
	u32 r = x, t, e = g;
	if (MODE == 0) asm volatile("v_add_u32 %0, %0, %3\n v_pk_maximum3_f16 %0, %0, %2, %3\n v_sub_u32 %1, %0, %3\n v_pk_max_i16 %0, %0, %1\n v_sub_u32 %0, %0, %3\n v_pk_maximum3_f16 %2, %2, %1, %3" : "+v"(r), "=&v"(t), "+v"(e) : "v"(g));
	if (MODE == 1) asm volatile("v_pk_add_u16 %0, %0, %3\n v_pk_maximum3_f16 %0, %0, %2, %3\n v_pk_sub_u16 %1, %0, %3\n v_pk_max_i16 %0, %0, %1\n v_pk_sub_u16 %0, %0, %3\n v_pk_maximum3_f16 %2, %2, %1, %3" : "+v"(r), "=&v"(t), "+v"(e) : "v"(g));
	if (MODE == 2) asm volatile("v_add_u32 %0, %0, %3\n v_pk_max_i16 %0, %0, %3\n v_add_u32 %0, %0, %3\n v_pk_max_i16 %0, %0, %3\n v_add_u32 %0, %0, %3\n v_pk_max_i16 %0, %0, %3" : "+v"(r), "=&v"(t), "+v"(e) : "v"(g));
	if (MODE == 3) asm volatile("v_add_u32 %0, %0, %3\n v_add_u32 %0, %0, %3\n v_add_u32 %0, %0, %3\n v_pk_max_i16 %0, %0, %3\n v_pk_max_i16 %0, %0, %3\n v_pk_max_i16 %0, %0, %3" : "+v"(r), "=&v"(t), "+v"(e) : "v"(g));
	return r ^ e;
}
template <int MODE> __global__ void __launch_bounds__(256) k_mix(u32* sink, int iters, u32 g)
{
	const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
	u32 x0 = gid, x1 = gid * 3, x2 = gid * 5, x3 = gid * 7, x4 = gid * 11, x5 = gid * 13, x6 = gid * 17, x7 = gid * 19;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int k2 = 0; k2 < 2; ++k2) { x0 = mixop<MODE>(x0, g); x1 = mixop<MODE>(x1, g); x2 = mixop<MODE>(x2, g); x3 = mixop<MODE>(x3, g); x4 = mixop<MODE>(x4, g); x5 = mixop<MODE>(x5, g); x6 = mixop<MODE>(x6, g); x7 = mixop<MODE>(x7, g); }
	}
	sink[gid] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
// single-instruction issue cost: 8 independent chains x 12 instructions per iteration
template <int MODE> DEV u32 op(u32 x, u32 g)
{
	u32 r = x;
	if (MODE == 0) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 1) asm volatile("v_sub_u32 %0, %0, %1\n v_sub_u32 %0, %0, %1\n v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 2) asm volatile("v_max_u32 %0, %0, %1\n v_max_u32 %0, %0, %1\n v_max_u32 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 3) asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 4) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %1\n v_pk_maximum3_f16 %0, %0, %1, %1\n v_pk_maximum3_f16 %0, %0, %1, %1" : "+v"(r) : "v"(g));
	if (MODE == 5) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1" : "+v"(r) : "v"(g));
	if (MODE == 6) asm volatile("v_max_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n v_max_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n v_max_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1" : "+v"(r) : "v"(g));
	if (MODE == 7) asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %0, %0, %1\n v_max_f32 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 8) asm volatile("v_max_u16 %0, %0, %1\n v_max_u16 %0, %0, %1\n v_max_u16 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 9) asm volatile("v_max_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n v_max_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n v_max_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(r) : "v"(g));
	if (MODE == 10) asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 11) asm volatile("v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 12) asm volatile("v_max_i16 %0, %0, %1\n v_max_i16 %0, %0, %1\n v_max_i16 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 13) asm volatile("v_max_f16 %0, %0, %1\n v_max_f16 %0, %0, %1\n v_max_f16 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 14) asm volatile("v_max3_u16 %0, %0, %1, %1\n v_max3_u16 %0, %0, %1, %1\n v_max3_u16 %0, %0, %1, %1" : "+v"(r) : "v"(g));
	if (MODE == 15) asm volatile("v_and_b32 %0, %0, %1\n v_or_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 16) asm volatile("v_bfi_b32 %0, %0, %1, %1\n v_bfi_b32 %0, %0, %1, %1\n v_bfi_b32 %0, %0, %1, %1" : "+v"(r) : "v"(g));
	if (MODE == 17) asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %0, %0, %1\n v_max_i32 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 18) asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %0, %0, %1\n v_min_u32 %0, %0, %1" : "+v"(r) : "v"(g));
	if (MODE == 19) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(g) : "vcc");
	if (MODE == 20) asm volatile("v_maximum3_f32 %0, %0, %1, %1\n v_maximum3_f32 %0, %0, %1, %1\n v_maximum3_f32 %0, %0, %1, %1" : "+v"(r) : "v"(g));
	if (MODE == 21) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp\n v_pk_sub_u16 %0, %0, %1 clamp\n v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(r) : "v"(g));
	return r;
}
template <int MODE> __global__ void __launch_bounds__(256) k_op(u32* sink, int iters, u32 g)
{
	const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
	u32 x0 = gid, x1 = gid * 3, x2 = gid * 5, x3 = gid * 7, x4 = gid * 11, x5 = gid * 13, x6 = gid * 17, x7 = gid * 19;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int k2 = 0; k2 < 4; ++k2) { x0 = op<MODE>(x0, g); x1 = op<MODE>(x1, g); x2 = op<MODE>(x2, g); x3 = op<MODE>(x3, g); x4 = op<MODE>(x4, g); x5 = op<MODE>(x5, g); x6 = op<MODE>(x6, g); x7 = op<MODE>(x7, g); }
	}
	sink[gid] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
static double clk_ghz = 2.4;
template <int MODE> void run_op(const char* name, u32* sink, int per_iter = 96)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	const int blocks = 8192, iters = 1000;
	k_op<MODE><<<blocks, 256>>>(sink, 10, 0x00030001u); hipDeviceSynchronize();
	float best = 1e9;
	for (int r = 0; r < 3; ++r) { hipEventRecord(a); k_op<MODE><<<blocks, 256>>>(sink, iters, 0x00030001u); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
	const double winstr = (double)blocks * 4 * iters * per_iter;             /* wave-instructions */
	const double cyc = best * 1e-3 * clk_ghz * 1e9 * 1024.0 / winstr;        /* SIMD-cycles per wave-instruction at 1024 SIMDs */
	printf("%-34s %8.3f ms  %6.2f cycles/wave-instr (at %.2f GHz)\n", name, best, cyc, clk_ghz);
}
template <typename K> void run_form(const char* name, K kern, u32* sink)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	const int blocks = 8192, steps = 20000;
	kern<<<blocks, 256>>>(sink, 100, 1u); hipDeviceSynchronize();
	float best = 1e9;
	for (int r = 0; r < 3; ++r) { hipEventRecord(a); kern<<<blocks, 256>>>(sink, steps, 1u); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
	const double wsteps = (double)blocks * 4 * steps;
	const double cyc = best * 1e-3 * clk_ghz * 1e9 * 1024.0 / wsteps;
	const double cells = wsteps * 64.0 * R * 2.0;
	printf("%-34s %8.3f ms  %7.1f SIMD-cycles per step of %d rows  = %.2f TCUPS-equivalent (all rows real)\n", name, best, cyc, R, cells / (best * 1e-3) / 1e12);
}
template <int ADDS> void run_formB(const char* name, u32* sink)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	const int blocks = 8192, steps = 20000;
	k_formB<ADDS><<<blocks, 256>>>(sink, 100, 1u, 0x01000100u, 0x00010001u, 0x00020002u); hipDeviceSynchronize();
	float best = 1e9;
	for (int r = 0; r < 3; ++r) { hipEventRecord(a); k_formB<ADDS><<<blocks, 256>>>(sink, steps, 1u, 0x01000100u, 0x00010001u, 0x00020002u); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
	const double wsteps = (double)blocks * 4 * steps;
	const double cyc = best * 1e-3 * clk_ghz * 1e9 * 1024.0 / wsteps;
	const double cells = wsteps * 64.0 * R * 2.0;
	printf("%-38s %8.3f ms  %7.1f SIMD-cycles per step of %d rows  = %.2f TCUPS-equivalent (all rows real)\n", name, best, cyc, R, cells / (best * 1e-3) / 1e12);
}
template <int MODE> void run_mix(const char* name, u32* sink)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	const int blocks = 8192, iters = 1000;
	k_mix<MODE><<<blocks, 256>>>(sink, 10, 0x00030001u); hipDeviceSynchronize();
	float best = 1e9;
	for (int r = 0; r < 3; ++r) { hipEventRecord(a); k_mix<MODE><<<blocks, 256>>>(sink, iters, 0x00030001u); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
	const double groups = (double)blocks * 4 * iters * 16;
	printf("%-38s %8.3f ms  %6.2f cycles per group of 6 instructions\n", name, best, best * 1e-3 * clk_ghz * 1e9 * 1024.0 / groups);
}
int main()
{
	u32* sink; hipMalloc(&sink, 8192 * 256 * 4);
	run_form("A f16 form (rounds 1-2)", k_formA, sink);
	run_formB<0>("B column-frame int16, u32 adds", sink);
	run_formB<1>("B column-frame int16, packed adds", sink);
	run_formB<2>("B column-frame int16, asm row pairs", sink);
	run_mix<0>("row of B: add max3 sub max sub max3", sink); run_mix<1>("same, packed adds", sink); run_mix<2>("add max add max add max", sink); run_mix<3>("add add add max max max", sink);
	run_op<0>("v_add_u32", sink); run_op<1>("v_sub_u32", sink); run_op<2>("v_max_u32", sink); run_op<17>("v_max_i32", sink); run_op<18>("v_min_u32", sink);
	run_op<7>("v_max_f32", sink); run_op<20>("v_maximum3_f32", sink);
	run_op<3>("v_pk_max_i16", sink); run_op<11>("v_pk_max_u16", sink); run_op<4>("v_pk_maximum3_f16", sink); run_op<10>("v_pk_add_u16", sink); run_op<21>("v_pk_sub_u16 clamp", sink);
	run_op<8>("v_max_u16", sink); run_op<12>("v_max_i16", sink); run_op<13>("v_max_f16", sink); run_op<9>("v_max_u16_sdwa hi", sink); run_op<14>("v_max3_u16", sink);
	run_op<15>("v_and/or/xor_b32", sink); run_op<16>("v_bfi_b32", sink); run_op<19>("v_cmp+v_cndmask", sink, 128);
	run_op<5>("v_mov_b32_dpp (+s_nop 1)", sink); run_op<6>("v_max_u32_dpp (+s_nop 1)", sink);
	return 0;
}
