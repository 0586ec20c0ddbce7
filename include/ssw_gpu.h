/*
 * ssw_gpu.h -- batch entry points of the MI355X-native Smith-Waterman library.
 *
 * The reference library aligns ONE (query, target) pair per synchronous call
 * (reference src/ssw.h:126-134); its callers loop "for each read: ssw_init; for
 * each target: ssw_align" (reference src/main.c:462-526, src/pyssw.py:120-142,
 * src/ssw_cpp.cpp:319-357).  This header is what such a loop binds instead when
 * it wants GPU throughput: the same parameters, applied to a whole batch of
 * device-resident queries x targets, returning one s_align-equivalent record per
 * pair in the caller's loop order (query-major: reads outer, targets inner).
 * ssw.h's single-pair functions are implemented on top of this path.
 *
 * Plain C ABI: pointers and sizes only.  All functions return 0 on success and a
 * negative value on failure (ssw_gpu_last_error() has the text); there is no CPU
 * fallback -- without a usable gfx950 device every entry point fails.
 */
#ifndef SSW_GPU_H
#define SSW_GPU_H

#include <stddef.h>
#include <stdint.h>
#include "ssw.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssw_gpu_ctx ssw_gpu_ctx;    /* one device + its streams/workspaces */
typedef struct ssw_gpu_seqs ssw_gpu_seqs;  /* a set of residue-code sequences resident in HBM */

/* Scoring / reporting parameters: the arguments of reference ssw_init + ssw_align
   (src/ssw.h:86, 126-134) that are not the sequences themselves. */
typedef struct {
	const int8_t* mat;   /* n*n scores, mat[target_code*n + read_code] (host pointer) */
	int32_t n;           /* alphabet size, any n >= 1 like the reference: up to 32 letters on the profile kernels; above that (and whenever gapO <= gapE) on the
	                        lane-model kernel -- the reference's answer at CPU-class speed; beyond 128 only the leading 128 x 128 block is addressable by int8 codes */
	uint8_t gapO;        /* weight_gapO */
	uint8_t gapE;        /* weight_gapE */
	uint8_t flag;        /* as ssw_align */
	uint16_t filters;
	int32_t filterd;
	int32_t maskLen;     /* >= 0: used for every query; < 0: readLen/2 per query (reference src/main.c:464) */
	int8_t score_size;   /* as ssw_init: 0, 1 or 2 */
	int8_t mark_mismatch;/* != 0: every returned CIGAR is rewritten on the device the way mark_mismatch() does it (soft clips,
	                        '=' / 'X' runs) and edit_distance receives its return value (reference src/ssw.c:1019-1074) */
} ssw_gpu_params;

/* One alignment: the fields of s_align (reference src/ssw.h:55-66) with the CIGAR
   held in a shared pool instead of a per-result pointer. */
typedef struct {
	uint16_t score1;
	uint16_t score2;
	int32_t ref_begin1;
	int32_t ref_end1;
	int32_t read_begin1;
	int32_t read_end1;
	int32_t ref_end2;
	int32_t cigarLen;    /* 0: no path */
	int32_t edit_distance; /* mark_mismatch()'s return value when params.mark_mismatch is set, else 0 */
	int64_t cigar_off;   /* first word of this CIGAR in the pool returned by ssw_gpu_align_batch */
	uint16_t flag;       /* as s_align.flag */
	uint16_t status;     /* 0 ok; 1 the reference returns NULL here (8-bit overflow with score_size 0) */
} ssw_gpu_result;

/* Per-phase device time of the last batch call, from HIP events on the library's stream. */
typedef struct {
	double total_ms;       /* first launch .. results on host */
	double fill_ms;        /* sum over forward-fill kernel launches */
	int64_t fill_launches; /* (database search: launch groups -- all size classes of one chunk of targets, side by side on several streams) */
	int64_t fill_cells;    /* DP cells actually evaluated by the fill kernel (padding + halo included) */
	int64_t cells;         /* sum of readLen*refLen over the batch (the GCUPS numerator) */
	double reduce_ms;      /* score1/score2/end-position reduction */
	double locate_ms;      /* read_end1 + reverse (begin position) passes */
	double trace_ms;       /* banded traceback + CIGAR re-score */
	int64_t n_word;        /* alignments decided under 16-bit semantics */
	int64_t n_byte;        /* alignments decided under 8-bit semantics */
	/* the fill kernel that evaluated most cells in the call (for roofline accounting) */
	char fill_kernel[48];  /* e.g. "k_fill<10,frame>", "k_chainq<12,frame> x 14 strips", "k_filldb<19,frame>" */
	double fill_ops_per_row; /* VALU instructions of the recurrence per (row, column) of a query pair in that kernel: 6.5 (column frame) / 7.5 / 8.5 / 9 */
	int32_t fill_rows_per_lane;
	int32_t fill_strips;
	int64_t db_repeats;    /* (rounds 1-2: workgroups of the database search that repeated in the int16 form; always 0 since the column-frame form) */
	int64_t fill_pipelined; /* fill launches of the call that ran as a PIPELINED series (main stream / a second stream alternately, two launches in flight, round 6): they overlap,
	                           fill_ms brackets each series as a whole -- fill_ms / fill_launches is then the series' time per launch, not a kernel's duration */
	/* new fields are only ever appended here; ssw_gpu_last_timing_sized lets a caller built against an older header keep its layout */
} ssw_gpu_timing;

int ssw_gpu_device_count(void);
ssw_gpu_ctx* ssw_gpu_open(int device);          /* NULL on failure */
void ssw_gpu_close(ssw_gpu_ctx* ctx);
const char* ssw_gpu_last_error(const ssw_gpu_ctx* ctx);   /* ctx may be NULL: error of the last failed open */

/* Upload `count` sequences: codes of sequence i are codes[offsets[i] .. offsets[i+1]). */
ssw_gpu_seqs* ssw_gpu_seqs_upload(ssw_gpu_ctx* ctx, const int8_t* codes, const int64_t* offsets, int32_t count);
/* Same, from ASCII residues: the translation through `table128` (e.g. the reference's nt_table / aa_table,
   src/main.c:72-93, applied per read at src/main.c:476, 504) runs on the device. */
ssw_gpu_seqs* ssw_gpu_seqs_upload_ascii(ssw_gpu_ctx* ctx, const char* text, const int64_t* offsets, int32_t count,
                                        const int8_t* table128);
/* Reverse complement of every sequence of a DNA code set (0..3 -> 3 - code, other codes kept), computed on the device:
   what `ssw_test -r` builds per read on the host (reference src/main.c:95-116, 478-481). */
ssw_gpu_seqs* ssw_gpu_seqs_revcomp(ssw_gpu_ctx* ctx, const ssw_gpu_seqs* s);
/* The sequences of `s` followed by their reverse complements: one set of 2 x count sequences (sequence count + i is the reverse
   complement of sequence i) -- `ssw_test -r` as ONE batch call of 2 N queries (reference src/main.c:478-481, 507-519). */
ssw_gpu_seqs* ssw_gpu_seqs_with_revcomp(ssw_gpu_ctx* ctx, const ssw_gpu_seqs* s);
int ssw_gpu_seqs_download(ssw_gpu_ctx* ctx, const ssw_gpu_seqs* s, int8_t* codes_out);
void ssw_gpu_seqs_free(ssw_gpu_seqs* s);
int32_t ssw_gpu_seqs_count(const ssw_gpu_seqs* s);

/*
 * Align every query to targets [target_first, target_first + target_count).
 * results[q * target_count + t] receives the record of (query q, target target_first + t).
 * cigar_pool (optional) receives a malloc()ed array of *cigar_words BAM-packed words (caller frees).
 *
 * Against four or more short targets the call is ONE database search whatever the flag: scores and end positions of all pairs from a
 * fused kernel; with flag != 0 the pairs that pass the reference's gates (src/ssw.c:916: not when flag == 2 and score1 < filters) go
 * through one batched reverse pass and traceback -- the loop of src/main.c:493-506 without a per-target iteration.  Queries of mixed
 * lengths run side by side.  A call with one query and one target is what ssw_align is made of.
 */
int ssw_gpu_align_batch(ssw_gpu_ctx* ctx, const ssw_gpu_seqs* queries, const ssw_gpu_seqs* targets,
                        int32_t target_first, int32_t target_count, const ssw_gpu_params* params,
                        ssw_gpu_result* results, uint32_t** cigar_pool, int64_t* cigar_words);

int ssw_gpu_last_timing(const ssw_gpu_ctx* ctx, ssw_gpu_timing* out);
/* the first min(out_size, sizeof(ssw_gpu_timing)) bytes of the record (the rest of `out`, if any, zeroed): a binary built against
   an earlier, shorter ssw_gpu_timing passes ITS sizeof and is not written past it */
int ssw_gpu_last_timing_sized(const ssw_gpu_ctx* ctx, void* out, size_t out_size);

/* Return codes of the batch calls: 0 ok; -1 failed, ssw_gpu_last_error(ctx) says why; SSW_GPU_BUSY another thread is inside this
   context (no message is written: the running call owns the error text); > 0 from ssw_gpu_search_db: the value the caller's chunk
   function returned to stop the search. */
#define SSW_GPU_BUSY (-2)
const char* ssw_gpu_strerror(int rc);

/* Scratch budget of a context in bytes (column maxima, boundary records, traceback scratch).  Default: min(64 GiB, half of the HBM that
   was free when the context was opened), or SSW_GPU_CM_BUDGET_MB -- it assumes nothing about other users of the device (HIP does not
   refuse another process's over-subscribing allocation; the failure would only show at a kernel launch).  Contexts sharing a device
   (several ranks or pool workers per GPU) should each get their share: ssw_gpu_pool_open does that for its workers.
   ssw_gpu_set_budget(ctx, 0) recomputes the default; ssw_gpu_set_budget_exclusive(ctx) is the caller's statement that the context
   has the device to itself: min(200 GiB, 60 % of the free HBM) -- the 288 GB of an MI355X then hold 3 fill launches per 100 000
   150-bp reads against 1 Mb instead of 6 (+1 %).  UNSAFE on a device that other processes use: their allocations are not refused, a later
   kernel launch of either side fails instead.  Every phase of a call, the traceback included, stays within the budget. */
int ssw_gpu_set_budget(ssw_gpu_ctx* ctx, size_t bytes);
int ssw_gpu_set_budget_exclusive(ssw_gpu_ctx* ctx);
size_t ssw_gpu_get_budget(const ssw_gpu_ctx* ctx);

/*
 * Database search with streamed results.  Every query against every target, scores and end positions only -- what the
 * reference's loop (src/main.c:462-526) computes with flag 0 -- for sizes whose result matrix does not fit anywhere at once
 * (BASELINE config 5: 50 000 x 10 000 = 5e8 alignments).  Targets are processed in chunks of `targets_per_chunk`; for every
 * chunk `fn` receives the compact records hits[q * target_count + t] of targets [target_first, target_first + target_count)
 * (valid only during the call) while the device already works on the next chunk.  fn returns 0 to go on, non-zero to stop
 * (ssw_gpu_search_db then returns that value).  params->flag must be 0.
 */
typedef struct {
	uint16_t score1;     /* as s_align */
	uint16_t score2;
	int32_t ref_end1;
	int32_t read_end1;
	int32_t ref_end2;    /* -2: the reference returns NULL for this pair (8-bit overflow with score_size 0) */
} ssw_gpu_hit;
typedef int (*ssw_gpu_hits_fn)(void* user, int32_t target_first, int32_t target_count, const ssw_gpu_hit* hits);
int ssw_gpu_search_db(ssw_gpu_ctx* ctx, const ssw_gpu_seqs* queries, const ssw_gpu_seqs* targets, const ssw_gpu_params* params,
                      int32_t targets_per_chunk, ssw_gpu_hits_fn fn, void* user);

/*
 * Threads.  A context owns one stream set and its workspaces: ONE call at a time per context (a second thread entering
 * ssw_gpu_align_batch on a busy context gets an error, not a race).  Different contexts -- on the same or on different
 * devices -- are independent and may be driven from different threads concurrently.  The single-pair functions of ssw.h
 * use an implicit context per calling thread (devices assigned round-robin, or SSW_GPU_DEVICE), so concurrent ssw_align
 * calls on one const s_profile* are legal, as with the reference (src/ssw.c has no mutable global state).
 * Streaming callers: while a batch call runs on a context, ONE other thread may prepare the next block on the same context with the
 * ssw_gpu_seqs_* calls (upload, ASCII translation, reverse complement) and free finished sets; those run on the context's upload stream
 * and do not queue behind the batch call's kernels (ssw_test_gpu's three stages, bench.py --config 3 --full).  The context's lazily created
 * streams and its error text are guarded by a lock (round 6): the feeder thread's first upload may race the first batch call.
 */

/*
 * Several devices, one batch: per-GPU work queues (the loop of reference src/main.c:462-526 spread over a node).
 * A pool holds one worker (context + host thread) per entry of `devices`; the target set is replicated on every worker's
 * device, the queries stay on the host and are pulled in blocks of `block` reads from a shared queue, each block is
 * uploaded and aligned by whichever worker is free, and every record lands in its own slot results[q * target_count + t]
 * -- output order does not depend on which device did what.  No inter-device traffic at all.
 */
typedef struct ssw_gpu_pool ssw_gpu_pool;
/* devices == NULL: one worker per visible device (n ignored); else n entries, an index may repeat (several workers on one
   device share it through separate streams). */
ssw_gpu_pool* ssw_gpu_pool_open(const int* devices, int n);      /* NULL on failure: ssw_gpu_last_error(NULL) */
void ssw_gpu_pool_close(ssw_gpu_pool* pool);
int ssw_gpu_pool_size(const ssw_gpu_pool* pool);
size_t ssw_gpu_pool_budget(const ssw_gpu_pool* pool, int worker);   /* scratch budget of that worker's context: workers sharing a device share its HBM */
const char* ssw_gpu_pool_last_error(const ssw_gpu_pool* pool);
int ssw_gpu_pool_set_targets(ssw_gpu_pool* pool, const int8_t* codes, const int64_t* offsets, int32_t count);
int ssw_gpu_pool_align(ssw_gpu_pool* pool, const int8_t* qcodes, const int64_t* qoffsets, int32_t nq, int32_t block,
                       int32_t target_first, int32_t target_count, const ssw_gpu_params* params,
                       ssw_gpu_result* results, uint32_t** cigar_pool, int64_t* cigar_words);
/* per worker, accumulated over the last ssw_gpu_pool_align: blocks taken, DP cells (readLen x refLen), device milliseconds */
typedef struct { int32_t device; int64_t blocks; int64_t queries; int64_t cells; double busy_ms; } ssw_gpu_pool_stat;
int ssw_gpu_pool_stats(const ssw_gpu_pool* pool, int worker, ssw_gpu_pool_stat* out);

/* Page-locked host memory for result arrays (optional): a database search returns nq x nt records, and their download
   runs at PCIe rate only into pinned pages.  Any host pointer is accepted by ssw_gpu_align_batch; this one is faster. */
void* ssw_gpu_host_alloc(ssw_gpu_ctx* ctx, size_t bytes);
void ssw_gpu_host_free(ssw_gpu_ctx* ctx, void* p);

/* (Diagnostics -- lane self-test, VALU issue probe, test-hook query -- are declared in include/ssw_gpu_diag.h and exist only in
   libssw_hooks.so: the product library exports the reference's symbols and the batch ABI above, one by one, in libssw.map.) */

/* Single-pair callers (ssw_align) get an implicit context per calling thread; a thread that ends parks its context for the next new caller
   thread (at most 4 stay parked).  This closes every parked context now -- their streams, scratch pools and resident target copies -- and
   returns how many were closed.  Call it from a live thread after a burst of short-lived caller threads. */
int ssw_gpu_release_parked(void);

/* Convert one batch record into a heap s_align (align_destroy()-compatible), copying its CIGAR. */
s_align* ssw_gpu_result_to_align(const ssw_gpu_result* r, const uint32_t* cigar_pool);

#ifdef __cplusplus
}
#endif
#endif /* SSW_GPU_H */
