/*
 * tests/emu/simt_emu.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A small fibre-based SIMT emulator: every GPU thread of one workgroup is a
 * cooperative fibre on a single OS thread; cross-lane operations (DPP moves,
 * shuffles, votes) and __syncthreads() are rendez-vous points.  It lets the
 * unmodified kernel source (csrc/ssw_kernels.hip, compiled as C++ with
 * -DSSW_SIMT_EMU) run in the CPU-only build container so that index math, skew,
 * rings and tile seams are debugged before GPU minutes are spent.  It is slow by
 * design and is never part of the product library.
 *
 * Semantics follow the gfx950 ISA: wave = 64 lanes, DPP rows = 16 lanes,
 * row_shr:n reads lane i-n (out-of-row lanes: 0 with bound_ctrl, else the lane
 * keeps `old`), row_ror:n reads lane (i-n) mod 16.
 */
#ifndef SIMT_EMU_H
#define SIMT_EMU_H

#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define SSW_DEV static inline
#define SSW_DEVM inline
#define SSW_HD static inline

namespace emu {

struct dim3_t { unsigned x, y, z; };

struct Wave {
	int live, arrived, gen;
	unsigned long long finished;   /* lanes whose kernel body has returned: ignored by votes */
	uint32_t slot[2][64];
};

struct Block {
	int live, arrived, gen;
	unsigned char* lds;
	size_t lds_bytes;
};

struct Fiber {
	void* sp;
	unsigned char* stack;
	int tid, lane, done;
	Wave* wave;
};

extern thread_local Fiber* cur;
extern thread_local Block blk;
extern thread_local dim3_t block_idx, block_dim, grid_dim;

void yield();
void wave_sync();
void block_sync();
[[noreturn]] void fail(const char* msg);

inline uint32_t exchange(uint32_t v, int src_lane, bool valid, uint32_t other)
{
	Wave* w = cur->wave;
	int g = w->gen & 1;
	w->slot[g][cur->lane] = v;
	wave_sync();
	return valid ? w->slot[g][src_lane] : other;
}

typedef void (*kernel_thunk)(void* args);
/* runs fn(args) for every thread of every block, blocks one after another */
void launch(kernel_thunk fn, void* args, unsigned grid, unsigned block, size_t lds_bytes);

} // namespace emu

struct emu_tid3 { unsigned x, y, z; };
#define threadIdx (emu_tid3{(unsigned)emu::cur->tid, 0u, 0u})
#define blockIdx (emu::block_idx)
#define blockDim (emu::block_dim)
#define gridDim (emu::grid_dim)
#define __syncthreads() emu::block_sync()
#define SSW_DYN_LDS(name) unsigned char* name = emu::blk.lds

typedef uint32_t u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

SSW_DEV u32 xl_row_shr1_zero(u32 v)
{
	int l = emu::cur->lane;
	return emu::exchange(v, l - 1, (l & 15) != 0, 0u);
}
SSW_DEV u32 xl_row_shr1_keep(u32 keep, u32 v)
{
	int l = emu::cur->lane;
	return emu::exchange(v, l - 1, (l & 15) != 0, keep);
}
SSW_DEV u32 xl_row_shr1_umax(u32 v, u32 b)
{
	const u32 t = xl_row_shr1_zero(v);
	return t > b ? t : b;
}
SSW_DEV void xl_row_shr1_sub_keep(u32& dst, u32 v, u32 b)
{
	int l = emu::cur->lane;
	const u32 t = emu::exchange(v, l - 1, (l & 15) != 0, 0u);
	if ((l & 15) != 0) dst = t - b;
}
SSW_DEV u32 xl_wave_shr1_keep(u32 keep, u32 v)
{
	int l = emu::cur->lane;
	return emu::exchange(v, l - 1, l != 0, keep);
}
SSW_DEV u32 xl_row_shl1_keep(u32 keep, u32 v)
{
	int l = emu::cur->lane;
	return emu::exchange(v, l + 1, (l & 15) != 15, keep);
}
SSW_DEV u32 xl_wave_shl1_keep(u32 keep, u32 v)
{
	int l = emu::cur->lane;
	return emu::exchange(v, l + 1, l != 63, keep);
}
template <int N> SSW_DEV u32 xl_row_ror(u32 v)
{
	int l = emu::cur->lane;
	return emu::exchange(v, (l & ~15) | ((l - N) & 15), true, 0u);
}
template <int N> SSW_DEV u32 xl_row_shr_keep(u32 keep, u32 v)
{
	int l = emu::cur->lane;
	return emu::exchange(v, l - N, (l & 15) >= N, keep);
}
SSW_DEV u32 xl_row_bcast15_keep(u32 keep, u32 v)
{
	int l = emu::cur->lane, row = l >> 4;
	return emu::exchange(v, (row << 4) - 1, row == 1 || row == 3, keep);
}
SSW_DEV u32 xl_row_bcast31_keep(u32 keep, u32 v)
{
	int l = emu::cur->lane;
	return emu::exchange(v, 31, l >= 32, keep);
}
SSW_DEV u32 xl_readlane(u32 v, int lane_uniform) { return emu::exchange(v, lane_uniform & 63, true, 0u); }
SSW_DEV u32 xl_shfl(u32 v, int src_lane) { return emu::exchange(v, src_lane & 63, true, 0u); }
SSW_DEV bool wave_any(bool p)
{
	emu::Wave* w = emu::cur->wave;
	int g = w->gen & 1;
	w->slot[g][emu::cur->lane] = p ? 1u : 0u;
	emu::wave_sync();
	bool r = false;
	for (int i = 0; i < 64; ++i) r = r || ((w->slot[g][i] & 1u) && !((w->finished >> i) & 1ull));
	return r;
}
SSW_DEV bool wave_all(bool p) { return !wave_any(!p); }
SSW_DEV void sched_fence() {}
#define SSW_WAVES_PER_EU(lo, hi)
SSW_DEV u32 after(u32 v, u32 dep) { (void)dep; return v; }
SSW_DEV u32 opaque(u32 v) { return v; }
SSW_DEV unsigned long long wave_ballot(bool p)
{
	emu::Wave* w = emu::cur->wave;
	int g = w->gen & 1;
	w->slot[g][emu::cur->lane] = p ? 1u : 0u;
	emu::wave_sync();
	unsigned long long m = 0;
	for (int i = 0; i < 64; ++i) if ((w->slot[g][i] & 1u) && !((w->finished >> i) & 1ull)) m |= 1ull << i;
	return m;
}
SSW_DEV void wave_lds_fence() { emu::wave_sync(); }
SSW_DEV void lds_barrier() { emu::block_sync(); }

static inline void emu_lds_check(u32 off, u32 bytes, u32 align, const char* what)
{
	if ((off % align) != 0 || (size_t)off + bytes > emu::blk.lds_bytes) {
		char m[160];
		snprintf(m, sizeof m, "LDS %s: offset %u (size %u, align %u) outside/unaligned in %zu-byte segment", what, off, bytes, align, emu::blk.lds_bytes);
		emu::fail(m);
	}
}
SSW_DEV u32x4 lds_ld128(const unsigned char* lds, u32 off) { emu_lds_check(off, 16, 16, "ld128"); u32x4 v; memcpy(&v, lds + off, 16); return v; }
SSW_DEV u32x2 lds_ld64(const unsigned char* lds, u32 off) { emu_lds_check(off, 8, 8, "ld64"); u32x2 v; memcpy(&v, lds + off, 8); return v; }
SSW_DEV u32 lds_ld32(const unsigned char* lds, u32 off) { emu_lds_check(off, 4, 4, "ld32"); u32 v; memcpy(&v, lds + off, 4); return v; }
SSW_DEV u32 lds_ld16(const unsigned char* lds, u32 off) { emu_lds_check(off, 2, 2, "ld16"); uint16_t v; memcpy(&v, lds + off, 2); return v; }
SSW_DEV void lds_st32(unsigned char* lds, u32 off, u32 v) { emu_lds_check(off, 4, 4, "st32"); memcpy(lds + off, &v, 4); }
SSW_DEV void lds_st128(unsigned char* lds, u32 off, u32x4 v) { emu_lds_check(off, 16, 16, "st128"); memcpy(lds + off, &v, 16); }
SSW_DEV int lds_ld8s(const unsigned char* lds, u32 off) { emu_lds_check(off, 1, 1, "ld8"); return (int)(int8_t)lds[off]; }
SSW_DEV void lds_st8(unsigned char* lds, u32 off, u32 v) { emu_lds_check(off, 1, 1, "st8"); lds[off] = (unsigned char)v; }
SSW_DEV void dev_fence() { emu::wave_sync(); }
SSW_DEV void wg_fence() { emu::wave_sync(); }
SSW_DEV void lds_st16(unsigned char* lds, u32 off, u32 v) { emu_lds_check(off, 2, 2, "st16"); uint16_t h = (uint16_t)v; memcpy(lds + off, &h, 2); }

static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
/* work-queue primitives (csrc/lanes.h): blocks run one after the other here, so a workgroup drains the queue in ticket order
   and every item's predecessor is complete when it is drawn */
SSW_DEV int dev_ticket(int* counter) { return atomicAdd(counter, 1); }
SSW_DEV void dev_flag_set(int* flag) { *flag = 1; }
SSW_DEV bool dev_flag_wait(int* flag) { return *flag != 0; }

#endif /* SIMT_EMU_H */
