/* GPU box: N caller threads, each looping "ssw_init; ssw_align; align_destroy; init_destroy" over its share of the reads against one target --
   what an unmodified multi-threaded caller of the reference's C API does (src/ssw.h:86-134; the library is re-entrant).  Prints calls per
   second and GCUPS per thread count.   build: gcc -O2 -I include scripts/probes/dropin_threads.c -o /tmp/dropin_threads -L <pkg> -lssw -lpthread
   usage: dropin_threads <reads> <read_len> <ref_len> <flag> <threads...> */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "ssw.h"

static int8_t* ref; static int32_t ref_len, read_len, nreads, flag_; static int8_t* reads; static int8_t mat[25];
typedef struct { int tid, nth; long ok; } job;
static void* work(void* p)
{
	job* j = (job*)p;
	for (int i = j->tid; i < nreads; i += j->nth) {
		s_profile* pr = ssw_init(reads + (size_t)i * read_len, read_len, mat, 5, 2);
		s_align* a = ssw_align(pr, ref, ref_len, 3, 1, (uint8_t)flag_, 0, 0, read_len / 2);
		if (a && a->score1 > 0) ++j->ok;
		align_destroy(a); init_destroy(pr);
	}
	return 0;
}
int main(int argc, char** argv)
{
	if (argc < 6) return 2;
	nreads = atoi(argv[1]); read_len = atoi(argv[2]); ref_len = atoi(argv[3]); flag_ = atoi(argv[4]);
	unsigned s = 12345;
	ref = (int8_t*)malloc(ref_len); reads = (int8_t*)malloc((size_t)nreads * read_len);
	for (int i = 0; i < ref_len; ++i) { s = s * 1664525u + 1013904223u; ref[i] = (int8_t)((s >> 24) & 3); }
	for (int i = 0; i < nreads; ++i) {
		s = s * 1664525u + 1013904223u; const int off = (int)((s >> 8) % (unsigned)(ref_len - read_len));
		for (int k = 0; k < read_len; ++k) { s = s * 1664525u + 1013904223u; reads[(size_t)i * read_len + k] = (s >> 27) == 0 ? (int8_t)((s >> 20) & 3) : ref[off + k]; }
	}
	for (int a = 0; a < 5; ++a) for (int b = 0; b < 5; ++b) mat[a * 5 + b] = a == 4 || b == 4 ? 0 : a == b ? 2 : -2;
	for (int t = 5; t < argc; ++t) {
		const int nth = atoi(argv[t]);
		double best = 1e30; long ok = 0;
		for (int rep = 0; rep < 2; ++rep) {      /* (the first round creates the threads' contexts) */
			pthread_t th[64]; job jb[64];
			struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
			for (int k = 0; k < nth; ++k) { jb[k].tid = k; jb[k].nth = nth; jb[k].ok = 0; pthread_create(&th[k], 0, work, &jb[k]); }
			ok = 0;
			for (int k = 0; k < nth; ++k) { pthread_join(th[k], 0); ok += jb[k].ok; }
			clock_gettime(CLOCK_MONOTONIC, &t1);
			const double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
			if (dt < best) best = dt;
		}
		printf("{\"threads\": %d, \"reads\": %d, \"read_len\": %d, \"ref_len\": %d, \"flag\": %d, \"aligned\": %ld, \"calls_per_s\": %.0f, \"ms_per_call_aggregate\": %.3f, \"gcups\": %.1f}\n",
		       nth, nreads, read_len, ref_len, flag_, ok, nreads / best, best / nreads * 1e3, (double)nreads * read_len * ref_len / best / 1e9);
	}
	return 0;
}
