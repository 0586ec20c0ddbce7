/*
 * tests/emu/simt_emu.cpp -- TEST INFRASTRUCTURE ONLY: scheduler of the SIMT emulator
 * declared in simt_emu.h (x86-64 System V only).
 */
#include "simt_emu.h"
#include <vector>

namespace emu {

/* all scheduler state is per host thread: every thread that launches a kernel runs its own fibres (the library's
   per-thread implicit contexts and the pool's workers launch concurrently) */
thread_local Fiber* cur = 0;
thread_local Block blk;
thread_local dim3_t block_idx, block_dim, grid_dim;

static thread_local void* sched_sp = 0;
static thread_local unsigned long progress = 0;
static thread_local kernel_thunk g_fn = 0;
static thread_local void* g_args = 0;

extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(
	".text\n"
	".globl emu_ctx_switch\n"
	".type emu_ctx_switch,@function\n"
	"emu_ctx_switch:\n"
	"  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
	"  movq %rsp, (%rdi)\n"
	"  movq %rsi, %rsp\n"
	"  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
	"  ret\n"
	".size emu_ctx_switch, .-emu_ctx_switch\n");

void yield() { emu_ctx_switch(&cur->sp, sched_sp); }

[[noreturn]] void fail(const char* msg)
{
	fprintf(stderr, "simt_emu: %s (block %u thread %d)\n", msg, block_idx.x, cur ? cur->tid : -1);
	abort();
}

void wave_sync()
{
	Wave* w = cur->wave;
	int g = w->gen;
	if (++w->arrived >= w->live) { w->arrived = 0; w->gen++; ++progress; }
	else while (w->gen == g) yield();
}

void block_sync()
{
	int g = blk.gen;
	if (++blk.arrived >= blk.live) { blk.arrived = 0; blk.gen++; ++progress; }
	else while (blk.gen == g) yield();
}

static void fiber_main()
{
	g_fn(g_args);
	Fiber* f = cur;
	f->done = 1; ++progress;
	Wave* w = f->wave;
	w->finished |= 1ull << f->lane;   /* its exchange slots stay readable: slower lanes may still be about to read them */
	w->live--;
	if (w->live > 0 && w->arrived >= w->live) { w->arrived = 0; w->gen++; }
	blk.live--;
	if (blk.live > 0 && blk.arrived >= blk.live) { blk.arrived = 0; blk.gen++; }
	yield();
	fail("finished fibre resumed");
}

void launch(kernel_thunk fn, void* args, unsigned grid, unsigned block, size_t lds_bytes)
{
	const size_t STK = 256 * 1024;
	g_fn = fn; g_args = args;
	grid_dim = dim3_t{grid, 1, 1}; block_dim = dim3_t{block, 1, 1};
	unsigned nw = (block + 63) / 64;
	std::vector<Fiber> fib(block);
	std::vector<Wave> waves(nw);
	std::vector<unsigned char> stacks((size_t)block * STK + 64);
	std::vector<unsigned char> lds(lds_bytes + 64);
	unsigned char* lds_base = lds.data() + ((64 - ((uintptr_t)lds.data() & 63)) & 63);
	for (unsigned b = 0; b < grid; ++b) {
		block_idx = dim3_t{b, 0, 0};
		memset(lds_base, 0xCD, lds_bytes);   /* poison: uninitialised LDS must not matter */
		blk.live = (int)block; blk.arrived = 0; blk.gen = 0; blk.lds = lds_base; blk.lds_bytes = lds_bytes;
		for (unsigned w = 0; w < nw; ++w) {
			waves[w].arrived = 0; waves[w].gen = 0; waves[w].finished = 0;
			waves[w].live = (int)((w + 1) * 64 <= block ? 64 : block - w * 64);
			memset(waves[w].slot, 0, sizeof(waves[w].slot));
		}
		for (unsigned t = 0; t < block; ++t) {
			Fiber& f = fib[t];
			f.tid = (int)t; f.lane = (int)(t & 63); f.done = 0; f.wave = &waves[t / 64];
			f.stack = stacks.data() + (size_t)t * STK;
			uintptr_t top = ((uintptr_t)(f.stack + STK)) & ~(uintptr_t)15;
			void** sp = (void**)top;
			*--sp = 0;                       /* fake return address of fiber_main */
			*--sp = (void*)fiber_main;       /* popped by ret in emu_ctx_switch */
			for (int k = 0; k < 6; ++k) *--sp = 0;
			f.sp = (void*)sp;
		}
		int remaining = (int)block;
		while (remaining > 0) {
			unsigned long before = progress;
			for (unsigned t = 0; t < block; ++t) {
				if (fib[t].done) continue;
				cur = &fib[t];
				emu_ctx_switch(&sched_sp, cur->sp);
				if (fib[t].done) --remaining;
			}
			if (progress == before && remaining > 0) { cur = 0; fail("deadlock: threads wait at different barriers (non-uniform control flow)"); }
		}
		cur = 0;
	}
}

} // namespace emu
