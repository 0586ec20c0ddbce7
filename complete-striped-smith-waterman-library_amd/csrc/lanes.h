/*
 * lanes.h -- the per-lane primitives the kernels are written in.
 *
 * gfx950 view: a 64-lane wavefront is four DPP "rows" of 16 lanes.  One row is
 * one systolic chain: lane i owns a block of consecutive query rows, and each
 * step hands H/F/column-max to lane i+1 with row_shr / row_ror DPP moves.  Scores
 * are two 16-bit values packed in a 32-bit VGPR (two queries per chain); the hot
 * form keeps them in a "column frame" (below) where the adds are plain 32-bit adds
 * and the maxima are v_pk_maximum3_f16 on the bit patterns; the fallback form uses
 * the packed saturating VOP3P instructions (v_pk_add_i16 clamp, v_pk_sub_u16 clamp,
 * v_pk_max_i16).  gfx950 has no packed 8-bit arithmetic.
 *
 * When SSW_SIMT_EMU is defined (tests/emu only) the same names are provided by a
 * fibre-based SIMT emulator so that the very same kernel source can be executed
 * and checked on a machine without a GPU.  The product library never defines it.
 */
#ifndef SSW_LANES_H
#define SSW_LANES_H

#include <stdint.h>

typedef uint32_t u32;
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#ifdef SSW_SIMT_EMU
#include "simt_emu.h"
#else
#include <hip/hip_runtime.h>
#define SSW_DEV __device__ __forceinline__
#define SSW_DEVM __device__ __forceinline__          /* member functions */
#define SSW_HD __host__ __device__ inline             /* also called by the launchers */
#define SSW_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

/* DPP controls (ISA encodings): row_shr:n = 0x110+n, row_ror:n = 0x120+n */
SSW_DEV u32 xl_row_shr1_zero(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x111, 0xf, 0xf, true); }
SSW_DEV u32 xl_row_shr1_keep(u32 keep, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x111, 0xf, 0xf, false); }
/* Column-frame chains: the hand-off to the next lane FUSED with the arithmetic that follows it -- a DPP modifier on a 32-bit VOP2 costs
   nothing, a separate v_mov_b32_dpp is one of ~75 vector instructions of a step in a kernel that is bound by their issue.  hipcc's DPP
   combiner does this only now and then (never across the basic blocks of a step), hence the asm.  `s_nop 4`: the compiler's hazard
   recogniser does not look into an asm block, so the block pads for the longest DPP hazard itself -- five wait states after a vector write
   of EXEC (two after the vector instruction that wrote the operand) -- whatever code happens to precede it after a compiler update; the
   three extra scalar cycles are filled by the other wavefronts of the SIMD (measured: profiles/round4_bench_default_with_also.json).
   xl_row_shr1_umax: max(previous lane's v, b) as unsigned words; lane 0 of a row: b (the zero row_shr:1 fills in, bound_ctrl).
   xl_row_shr1_sub_keep: dst = previous lane's v - b; lane 0 of a row keeps dst (no bound_ctrl: lanes without a source are disabled). */
SSW_DEV u32 xl_row_shr1_umax(u32 v, u32 b)
{
	u32 r;
	asm("s_nop 4\n\tv_max_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(b));
	return r;
}
SSW_DEV void xl_row_shr1_sub_keep(u32& dst, u32 v, u32 b)
{
	asm("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(dst) : "v"(v), "v"(b));
}
template <int N> SSW_DEV u32 xl_row_ror(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x120 + N, 0xf, 0xf, true); }
/* wave_shr:1 (0x138, GFX9 family only): lane i reads lane i-1 across the whole wavefront; lane 0 keeps `keep` */
SSW_DEV u32 xl_wave_shr1_keep(u32 keep, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x138, 0xf, 0xf, false); }
/* row_shr:N keeping `keep` where the row has no lane N to the left; row_bcast:15 / row_bcast:31 (GFX9 family): lane 15 of
   every row to the lanes of the next row (rows 1 and 3 take it), lane 31 to rows 2 and 3 -- the two cross-row steps of a
   wavefront-wide prefix scan; lanes outside the row mask keep `keep` */
template <int N> SSW_DEV u32 xl_row_shr_keep(u32 keep, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x110 + N, 0xf, 0xf, false); }
/* row_shl:1 (0x101) / wave_shl:1 (0x130): lane i reads lane i+1 inside its row of 16 / across the wavefront; the last lane keeps `keep` */
SSW_DEV u32 xl_row_shl1_keep(u32 keep, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x101, 0xf, 0xf, false); }
SSW_DEV u32 xl_wave_shl1_keep(u32 keep, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x130, 0xf, 0xf, false); }
SSW_DEV u32 xl_row_bcast15_keep(u32 keep, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x142, 0xa, 0xf, false); }
SSW_DEV u32 xl_row_bcast31_keep(u32 keep, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x143, 0xc, 0xf, false); }
/* value of one lane (wavefront-uniform index) in every lane */
SSW_DEV u32 xl_readlane(u32 v, int lane_uniform) { return (u32)__builtin_amdgcn_readlane((int)v, lane_uniform); }
SSW_DEV u32 xl_shfl(u32 v, int src_lane) { return (u32)__shfl((int)v, src_lane, 64); }
SSW_DEV bool wave_any(bool p) { return __any(p) != 0; }
SSW_DEV bool wave_all(bool p) { return __all(p) != 0; }
SSW_DEV unsigned long long wave_ballot(bool p) { return __ballot(p); }
/* register budget of a kernel as wavefronts per SIMD (512 / n registers per lane); the emulator build ignores it */
#define SSW_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
/* `v`, but not before `dep` exists: an opaque data dependency for the instruction scheduler (no instruction is emitted).  Keeps a
   load that refills registers from being hoisted above the last use of their old contents (which doubles the registers). */
SSW_DEV u32 after(u32 v, u32 dep) { asm("" : "+v"(v) : "v"(dep)); return v; }
/* `v` with its origin hidden from the optimiser (no instruction): keeps a bit mask a bit mask (LLVM otherwise turns a mask built from a
   compare back into per-half selects -- three instructions per word where v_bfi_b32 is one) */
SSW_DEV u32 opaque(u32 v) { asm("" : "+v"(v)); return v; }
/* nothing is scheduled across this point (keeps a compare that feeds a LATER scalar branch where it was written) */
SSW_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
/* LDS hand-off between lanes of ONE wavefront (rings, small reductions): LDS operations of a wave execute in issue order,
   so what is needed is (a) that the compiler does not move LDS accesses across this point -- a wavefront-scope fence turned
   out NOT to stop it from hoisting a lane's loads of other lanes' slots above the stores (seen on gfx950, ROCm 7.2) -- and
   (b) nothing more than lgkmcnt(0).  Global loads in flight are not waited for. */
SSW_DEV void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
/* workgroup barrier that orders LDS traffic ONLY: __syncthreads() is also a memory fence, i.e. s_waitcnt vmcnt(0) in front of the s_barrier --
   a kernel that streams direction bytes to HBM and exchanges band rows through LDS then waits for a store round trip at every barrier
   (two per band row in the traceback teams).  The stores stay in flight here; whoever reads them back later fences first. */
SSW_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* LDS accessors with byte offsets into the dynamic segment */
SSW_DEV u32x4 lds_ld128(const unsigned char* lds, u32 off) { return *(const u32x4*)(lds + off); }
SSW_DEV u32x2 lds_ld64(const unsigned char* lds, u32 off) { return *(const u32x2*)(lds + off); }
SSW_DEV u32 lds_ld32(const unsigned char* lds, u32 off) { return *(const u32*)(lds + off); }
SSW_DEV u32 lds_ld16(const unsigned char* lds, u32 off) { return *(const uint16_t*)(lds + off); }
SSW_DEV void lds_st32(unsigned char* lds, u32 off, u32 v) { *(u32*)(lds + off) = v; }
SSW_DEV void lds_st128(unsigned char* lds, u32 off, u32x4 v) { *(u32x4*)(lds + off) = v; }
SSW_DEV int lds_ld8s(const unsigned char* lds, u32 off) { return *(const int8_t*)(lds + off); }
SSW_DEV void lds_st8(unsigned char* lds, u32 off, u32 v) { *(lds + off) = (unsigned char)v; }
/* orders this wavefront's own global-memory traffic (row arrays re-read by other lanes of the same wave) */
SSW_DEV void wg_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
SSW_DEV void dev_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); }
SSW_DEV void lds_st16(unsigned char* lds, u32 off, u32 v) { *(uint16_t*)(lds + off) = (uint16_t)v; }
/* work queue of a persistent launch: one lane draws the next ticket for its wavefront; completion flags in HBM order the
   items of one job (agent scope: the eight XCDs have their own L2s).  dev_flag_set comes after an agent-scope fence that
   every lane executed (its own stores are then visible); dev_flag_wait is followed by one. */
SSW_DEV int dev_ticket(int* counter) { return atomicAdd(counter, 1); }
SSW_DEV void dev_flag_set(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
/* The poll is a RELAXED agent-scope load: an acquire load puts a cache invalidate (buffer_inv sc1) into every iteration, and a
   dozen wavefronts polling that way keep the L2 of their XCD busy invalidating -- the wavefront they are waiting for then
   crawls (seen on MI355X: a test batch with more strips than jobs went from milliseconds to minutes).  One acquire fence after
   the flag has been seen (the caller's dev_fence) is all that is needed.  Bounded: a flag that does not come within 30 seconds
   means a broken queue -- the caller raises the launch's error word and goes on, so that the host fails the call instead
   of the device hanging. */
SSW_DEV bool dev_flag_wait(int* flag)
{
	/* bounded by WALL CLOCK (s_memrealtime: the 100 MHz constant clock), 30 s: a strip over a multi-megabase tile legitimately runs for
	   seconds, longer when ranks or pool workers share the device -- a poll count would fail such calls spuriously */
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	for (;;) {
		for (int spin = 0; spin < 4096; ++spin) {
			if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
			__builtin_amdgcn_s_sleep(64);
		}
		if (__builtin_amdgcn_s_memrealtime() - t0 > 3000000000ull) return false;
	}
}
#endif

/* hand-off to the next lane of a chain of GL lanes (16: one DPP row, 64: the whole wavefront) */
template <int GL> SSW_DEV u32 xl_chain_shr1_keep(u32 keep, u32 v) { return GL == 64 ? xl_wave_shr1_keep(keep, v) : xl_row_shr1_keep(keep, v); }

/* packed 2 x int16 arithmetic (identical source for device and emulation) */
SSW_DEV u32 pk_adds(u32 a, u32 b)   /* v_pk_add_i16 clamp: signed saturating add */
{
	return __builtin_bit_cast(u32, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
SSW_DEV u32 pk_subu(u32 a, u32 b)   /* v_pk_sub_u16 clamp: unsigned saturating subtract */
{
	return __builtin_bit_cast(u32, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
SSW_DEV u32 pk_minu(u32 a, u32 b)   /* v_pk_min_u16 */
{
	return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
SSW_DEV u32 pk_mullo(u32 a, u32 b)  /* v_pk_mul_lo_u16 */
{
	return __builtin_bit_cast(u32, __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b));
}
SSW_DEV u32 bfi32(u32 mask, u32 ins, u32 base) { return (mask & ins) | (~mask & base); }   /* v_bfi_b32 */
SSW_DEV u32 pk_max(u32 a, u32 b)    /* v_pk_max_i16 */
{
	return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
/* v_perm_b32: byte k of the result is byte sel[k] of the 8 bytes { lo (0..3), hi (4..7) }.  PK_LO2 / PK_HI2 gather the low / the
   high 16-bit halves of two registers into one (lo's half first). */
#ifdef SSW_SIMT_EMU
SSW_DEV u32 pk_perm(u32 hi, u32 lo, u32 sel)
{
	const unsigned long long src = ((unsigned long long)hi << 32) | lo;
	u32 r = 0;
	for (int k = 0; k < 4; ++k) r |= (u32)((src >> (8 * ((sel >> (8 * k)) & 7u))) & 0xffu) << (8 * k);
	return r;
}
#else
SSW_DEV u32 pk_perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#endif
#define PK_LO2 0x05040100u
#define PK_HI2 0x07060302u
/* max of three packed NON-NEGATIVE int16 pairs below 0x7C00 (31744) in one instruction: such bit patterns are positive
   finite binary16 numbers (denormals included) whose order is the integer order, and v_pk_maximum3_f16 returns one of its
   operands unchanged (checked on the device against integer max on 5e8 random triples, denormal range included).  Used
   for the running column maximum of two rows where the caller guarantees the range. */
#ifdef SSW_SIMT_EMU
SSW_DEV u32 pk_max3_nonneg(u32 a, u32 b, u32 c)
{
	const u32 lo = (a & 0xffffu) > (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu), hi = (a >> 16) > (b >> 16) ? (a >> 16) : (b >> 16);
	return ((c & 0xffffu) > lo ? (c & 0xffffu) : lo) | (((c >> 16) > hi ? (c >> 16) : hi) << 16);
}
#else
/* written with the builtin (nested llvm.maximum.v2f16 folds into ONE v_pk_maximum3_f16 on gfx950) rather than inline asm: the
   compiler then knows what the instruction is and schedules around it instead of padding every use with s_nop */
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
SSW_DEV u32 pk_max3_nonneg(u32 a, u32 b, u32 c)
{
	return __builtin_bit_cast(u32, __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
	                                                             __builtin_bit_cast(f16x2, c)));
}
#endif
/* ---- column frame: every stored value carries + phi(column), phi growing by gapE per column.  Then E needs no
   decrement, every add / subtract of the recurrence is a PLAIN 32-bit add on the packed pair (no carry can cross the halves: all
   live values are non-negative 15-bit numbers) -- v_add_u32 issues in 2 cycles where every VOP3P takes 4 -- and the maxima are
   v_pk_maximum3_f16 on the bit patterns (non-negative int16 below 0x7C00 are positive finite binary16 numbers in the same order).
   The one operand that may be "negative" is diag + score of a dead row / null column: pattern 0x8000 + value, a negative finite
   binary16 number, which loses against the other two.  The emulation takes the halves as signed 16-bit integers (same outcome:
   at most the first operand has bit 15 set). */
#ifdef SSW_SIMT_EMU
SSW_DEV u32 pk_max3_fr(u32 a, u32 b, u32 c)
{
	for (int h = 0; h < 32; h += 16) {   /* no infinity / NaN pattern in any operand; only the first may be "negative" */
		const u32 x = (a >> h) & 0xffffu, y = (b >> h) & 0xffffu, z = (c >> h) & 0xffffu;
		if ((x & 0x7c00u) == 0x7c00u || y >= 0x7c00u || z >= 0x7c00u) emu::fail("pk_max3_fr: operand outside the range of the frame form");
	}
	const int al = (short)(a & 0xffffu), ah = (short)(a >> 16);
	int lo = al > (int)(b & 0xffffu) ? al : (int)(b & 0xffffu), hi = ah > (int)(b >> 16) ? ah : (int)(b >> 16);
	lo = (int)(c & 0xffffu) > lo ? (int)(c & 0xffffu) : lo; hi = (int)(c >> 16) > hi ? (int)(c >> 16) : hi;
	return (u32)lo | ((u32)hi << 16);
}
#else
SSW_DEV u32 pk_max3_fr(u32 a, u32 b, u32 c) { return pk_max3_nonneg(a, b, c); }
#endif
/* packed profile entry of the frame form: each half is a small signed score (live row) or FR_DEAD (+32768 as an unsigned addend:
   diag + 0x8000 has bit 15 set and never carries).  A negative low half borrows from the high half here and gives the carry back
   at run time (diag + score >= 0 for every live row), so the packed sum is exact in both halves. */
#define FR_DEAD 32768
SSW_DEV u32 fr_pack(int lo, int hi) { return (u32)lo + ((u32)hi << 16); }
SSW_DEV u32 umax32(u32 a, u32 b) { return a > b ? a : b; }
/* 0xffff in every half where a > b (unsigned), else 0: saturating difference, min(., 1), x 0xffff -- three packed instructions.  One asm
   block on the device: written with the builtins, LLVM recognises the idiom at every step (umin(x, 1) -> x != 0, ...) and emits two 16-bit
   compares, two selects and a permute instead, also through empty-asm barriers */
#ifdef SSW_SIMT_EMU
SSW_DEV u32 pk_gt_mask(u32 a, u32 b) { return ((a & 0xffffu) > (b & 0xffffu) ? 0xffffu : 0u) | ((a >> 16) > (b >> 16) ? 0xffff0000u : 0u); }
#else
SSW_DEV u32 pk_gt_mask(u32 a, u32 b)
{
	u32 r;
	asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, %3\n\tv_pk_mul_lo_u16 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(0x00010001u), "v"(0xffffffffu));
	return r;
}
#endif
/* v_pk_add_u16 (wrapping, per half): the frame add where the two halves of a score register come from DIFFERENT profile entries (window
   passes: one target column per query half), so that the packed-sum trick of fr_pack does not apply -- entries are per-half two's complement there */
SSW_DEV u32 pk_addw(u32 a, u32 b) { return __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b))); }
/* phi of the column that lane `lane` of a GL-lane chain works on at step `step`: base + ((step mod K) + GL - lane) * gapE (all
   registers drop by K * gapE when step reaches a multiple of K) */
SSW_DEV int fr_phi(int step, int lane, int GL, int base, int kmask, int gapE) { return base + ((step & kmask) + GL - lane) * gapE; }
SSW_DEV u32 pk_dup(int v) { return ((u32)v & 0xffffu) * 0x10001u; }
SSW_DEV u32 pk_make(int lo, int hi) { return ((u32)lo & 0xffffu) | ((u32)hi << 16); }

#endif /* SSW_LANES_H */
