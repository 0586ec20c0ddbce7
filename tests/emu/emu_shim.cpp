/*
 * tests/emu/emu_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 * Host-memory implementation of the runtime half of the ssw_dev.h shim, so that the REAL host driver
 * (csrc/ssw_host.c) and the REAL kernel source (csrc/ssw_kernels.hip, compiled with -DSSW_SIMT_EMU on the
 * fibre emulator) can be linked into tests/emu/libssw_emu.so and exercised end-to-end without a GPU.
 * The product library (libssw.so) never contains this file.
 */
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include "ssw_dev.h"

extern "C" {
/* SSW_EMU_DEVICES=<n>: pretend to have n devices (round-robin / pool tests); they all are this process's memory */
int ssw_shim_device_count(void) { const char* e = getenv("SSW_EMU_DEVICES"); const int n = e ? atoi(e) : 1; return n >= 1 && n <= 64 ? n : 1; }
int ssw_shim_set_device(int) { return 0; }
const char* ssw_shim_last_error(void) { return "emulator"; }
void* ssw_shim_stream_create(void) { return (void*)1; }
void* ssw_shim_stream_create_low(void) { return (void*)1; }
void ssw_shim_stream_destroy(void*) {}
int ssw_shim_stream_sync(void*) { return 0; }
/* SSW_EMU_MALLOC_LIMIT_MB=<n>: a single device allocation above n MiB fails (the out-of-memory path of the host driver: SSW_ALLOC_RETRY) */
void* ssw_shim_malloc(size_t bytes)
{
	const char* e = getenv("SSW_EMU_MALLOC_LIMIT_MB");
	if (e && bytes > ((size_t)atoll(e) << 20)) return 0;
	void* p = malloc(bytes ? bytes : 16); if (p) memset(p, 0xEE, bytes); return p;
}
void ssw_shim_free(void* p) { free(p); }
void* ssw_shim_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 16); }
void ssw_shim_host_free(void* p) { free(p); }
int ssw_shim_h2d(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
int ssw_shim_d2h(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
int ssw_shim_memset(void* d, int v, size_t n, void*) { memset(d, v, n); return 0; }
size_t ssw_shim_mem_free_bytes(void) { return (size_t)1 << 30; }
int ssw_shim_device_props(int* compute_units, int* waves_per_cu) { *compute_units = 256; *waves_per_cu = 32; return 0; }
void* ssw_shim_event_create(void) { return calloc(1, sizeof(double)); }
void ssw_shim_event_destroy(void* e) { free(e); }
int ssw_shim_event_record(void* e, void*)
{
	*(double*)e = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
	return 0;
}
int ssw_shim_stream_wait_event(void*, void*) { return 0; }
int ssw_shim_event_sync(void*) { return 0; }
float ssw_shim_event_elapsed_ms(void* a, void* b) { return (float)(*(double*)b - *(double*)a); }
}
