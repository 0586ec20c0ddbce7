/*
 * ssw_dev.h -- plain-C structures shared by the C host driver (ssw_host.c) and the HIP
 * kernels (ssw_kernels.hip), plus the thin C shim through which the host reaches HIP.
 * Everything here is POD with fixed-width fields.
 */
#ifndef SSW_DEV_H
#define SSW_DEV_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSW_RMAX 24            /* rows per lane supported by the 16-lane chains: queries up to 16*24 = 384 residues */
#define SSW_MAX_N 32           /* alphabet size limit of the profile kernels (profile residues held in LDS) */
#define SSW_MAX_N_WIDE 128     /* wider alphabets (33 .. 128 letters: every int8 code) take the lane-model kernel with the matrix in LDS and the thread traceback */
#define SSW_LDS_LIMIT (160 * 1024)   /* LDS per workgroup on gfx950 */

/* two queries that share one systolic chain (low / high 16-bit half of every VGPR) */
typedef struct {
	int32_t qa;   /* query index of the low half */
	int32_t qb;   /* query index of the high half, -1: none */
} ssw_pair;

/* per-(query, current target) device record; filled phase by phase */
typedef struct {
	int32_t score1, score2;
	int32_t ref_begin1, ref_end1, read_begin1, read_end1, ref_end2;
	int32_t cigarLen;
	int32_t flag;        /* s_align.flag */
	int32_t status;      /* 0 ok; 1 the reference returns NULL (8-bit overflow, no 16-bit semantics enabled) */
	int32_t word;        /* 1: decided under 16-bit rules */
	int32_t want_begin;  /* phases still to run for this alignment (set by the reduce kernel) */
	int32_t want_cigar;
	int32_t rev_score;   /* best score seen by the reverse pass */
	int32_t loc_done;    /* read_end1 already known (tracked by the fill kernel): no locate pass */
	int32_t nm;          /* edit distance from the device-side mark_mismatch */
	int64_t cigar_off;   /* word offset of this alignment's CIGAR in the device CIGAR pool */
} ssw_dres;

/* Jobs that are (query, target) PAIRS instead of queries against the launch's one target (database search with begin positions /
   CIGARs: the survivors of the score filter, reference src/main.c:493-506 with flag != 0): the "query index" of a job is then a
   virtual id -- the position of the pair in a compact survivor list.  Records, CIGAR slots and resume state are indexed by the
   virtual id, the sequences are reached through the two maps.  vq == NULL: plain query indices, the args' own `tgt`. */
typedef struct {
	const int32_t* vq;       /* virtual id -> query index */
	const int32_t* vt;       /* virtual id -> target index */
	const int8_t* tcodes;    /* all target codes */
	const int64_t* toff;     /* target offsets */
} ssw_vmap;

/* forward fill: column maxima of every (pair, tile) of one target */
typedef struct {
	const int8_t* tgt;       /* codes of the target */
	int32_t refLen;
	const int8_t* qcodes;    /* all query codes */
	const int64_t* qoff;     /* query offsets */
	const ssw_pair* pairs;   /* pairs of this launch */
	int32_t npairs;
	const int8_t* mat;
	int32_t n;
	uint32_t gapO2, gapE2;   /* gap penalties replicated in both 16-bit halves */
	int32_t tile;            /* columns per tile (multiple of 16) */
	int32_t halo;            /* columns recomputed ahead of a tile so that its state is exact */
	int32_t ntiles;
	int32_t bpp;             /* workgroups per pair = ceil(ntiles / 16) */
	uint32_t* cm16;          /* [npairs][cm_stride]: column max over all 16R rows   (8-bit rules) */
	uint32_t* cm8;           /* [npairs][cm_stride]: column max over the first 16R-8 rows (16-bit rules, padded queries) */
	int64_t cm_stride;
	uint32_t* sg16;          /* optional: [npairs][seg_stride] maxima of every aligned group of 16 columns of cm16 ... */
	uint32_t* sg8;           /* ... and of cm8 (k_reduce_seg scans these instead of the columns) */
	int64_t seg_stride;
	int32_t form;            /* form of the recurrence: 3: column frame (ssw_frame_params in ssw_host.c says when), 6.5 instructions per row of
	                            which 3 are 32-bit adds; 0: plain int16 with the reference's saturation, 9 */
	int32_t fr_base, fr_kmask;   /* form 3: phi(column) = fr_base + ((step & fr_kmask) + lanes - lane) * gapE */
} ssw_fill_args;

/* several fill launches (geometry buckets of one register class) as ONE grid: k_fillm */
typedef struct {
	const ssw_fill_args* sub;   /* nsub argument records (device) */
	const int32_t* first_wg;    /* nsub + 1: first workgroup of every sub-launch, last = total */
	const int32_t* subR;        /* rows per lane of every sub-launch */
	int32_t nsub;
} ssw_fillm_args;

/* byte-for-byte the layout of ssw_gpu_result (include/ssw_gpu.h); checked by a static assertion in ssw_host.c */
struct ssw_out_rec {
	uint16_t score1, score2;
	int32_t ref_begin1, ref_end1, read_begin1, read_end1, ref_end2, cigarLen, edit_distance;
	int64_t cigar_off;
	uint16_t flag, status;
};

/* byte-for-byte the layout of ssw_gpu_hit (include/ssw_gpu.h) */
struct ssw_hit_rec {
	uint16_t score1, score2;
	int32_t ref_end1, read_end1, ref_end2;
};

#define SSW_DB_NCH 16     /* chains (targets) per workgroup of k_filldb; 32 -- twice as many share a profile -- measured 12 % slower */

/*
 * database search (many short targets, scores + end positions only): one workgroup = one query pair against 16
 * targets; the chain also tracks the best cell and reduces its own column maxima, so one launch produces final records.
 */
typedef struct {
	const int8_t* tcodes;    /* all target codes */
	const int64_t* toff;     /* target offsets (device) */
	const int32_t* tlist;    /* targets of this launch, sorted by length */
	int32_t ntl;
	int32_t tfirst;          /* res index of target t is (t - tfirst) */
	int32_t res_nt;          /* targets per query in res */
	const int8_t* qcodes;
	const int64_t* qoff;
	const ssw_pair* pairs;
	int32_t npairs;
	const int8_t* mat;
	int32_t n;
	uint32_t gapO2, gapE2;
	uint32_t* cm16;          /* [npairs * ntl][cm_stride] */
	uint32_t* cm8;
	int64_t cm_stride;
	int32_t maskLen, bias, score_size;
	ssw_dres* res;           /* [query][res_nt] (NULL when `out` is used) */
	struct ssw_out_rec* out; /* optional: final ssw_gpu_result-layout records [query][res_nt], downloaded as they are */
	int32_t chain_best;      /* 1: lanes learn the chain's best every 16 steps (fewer best-cell records); 0: lane-local records only (experiments) */
	struct ssw_hit_rec* hits;/* optional (takes precedence): compact 16-byte records [query][res_nt] of the streaming search */
	int32_t* counters;       /* optional: [0] alignments decided under 16-bit rules, [1] under 8-bit rules, [2] workgroups repeated in the int16 form */
	int32_t form;            /* 1: column-frame form of the recurrence (fr_base / fr_kmask as in ssw_fill_args), 0: plain int16 with the two-row maximum */
	int32_t fr_base, fr_kmask;
	int32_t mark_word;       /* `out` records carry SSW_OUT_WORD in their status when the pair was decided under 16-bit rules (k_select reads and clears it) */
} ssw_filldb_args;
#define SSW_OUT_WORD 0x100

/* Flagged database search: which (query, target) pairs of a chunk go on to the reverse pass / traceback (reference src/ssw.c:916: not
   when flag == 0 or (flag == 2 and score1 < filters)), compacted in (bucket-ordered query, target) order -- deterministic, and grouped
   by the geometry bucket whose window kernel they need.  pass 0: survivors per block of 256 pairs; pass 1 (one workgroup): exclusive
   scan of the block counts, total to total[0]; pass 2: survivor records + maps, SSW_OUT_WORD cleared in every record. */
typedef struct {
	struct ssw_out_rec* out; /* [query][nt] */
	const int32_t* order;    /* nk queries in bucket order */
	int32_t nk, nt;
	int32_t tbase;           /* target index of column 0 of `out` */
	int32_t flag, filters;
	int32_t pass;
	int32_t* blk;            /* nblk + 1 ints: counts, then exclusive offsets */
	int32_t nblk;
	const int64_t* bucket_lin; /* nbk + 1 linear indices k * nt at which the buckets start (last = nk * nt) */
	int32_t nbk;
	int32_t* bucket_first;   /* nbk + 1: first survivor of every bucket (last = total) */
	ssw_dres* sres;          /* survivors */
	int32_t* svq; int32_t* svt; int32_t* vlist;   /* maps + the identity list the window / traceback launches take as their job list */
	int32_t cap;             /* survivors the three arrays hold (pass 2 writes no further) */
} ssw_select_args;

/* reduction of the column maxima into score1 / ref_end1 / score2 / ref_end2 */
typedef struct {
	const uint32_t* cm16;
	const uint32_t* cm8;
	int64_t cm_stride;
	int32_t refLen;
	const ssw_pair* pairs;
	int32_t npairs;
	const int64_t* qoff;
	int32_t maskLen;         /* < 0: readLen / 2 */
	int32_t bias;
	int32_t score_size;
	int32_t flag, filters;
	ssw_dres* res;           /* indexed by query */
	const int32_t* cand;     /* optional: best cell tracked by the fill, [pair * ntiles + tile][half][4] = value, column, row, - */
	int32_t tile, ntiles;
	const uint32_t* sg16;    /* optional: group-of-16-columns maxima written by k_fill -> k_reduce_seg */
	const uint32_t* sg8;
	int64_t seg_stride;
} ssw_reduce_args;

/* the reductions of several buckets as ONE grid (one workgroup per pair): k_reducem */
typedef struct {
	const ssw_reduce_args* sub; /* nsub argument records (device), each with sg16 set */
	const int32_t* first_wg;    /* nsub + 1 */
	int32_t nsub;
} ssw_reducem_args;

/* locate (read_end1) and reverse (begin position) passes: one 16-lane chain per alignment */
typedef struct {
	const int8_t* tgt;
	int32_t refLen;
	const int8_t* qcodes;
	const int64_t* qoff;
	const int32_t* qlist;    /* queries of this launch */
	int32_t nq;
	const int8_t* mat;
	int32_t n;
	uint32_t gapO2, gapE2;
	int32_t gapE;
	int32_t maxmat;          /* largest matrix entry (halo bound) */
	int32_t reverse;         /* 0: locate read_end1; 1: reverse pass */
	int32_t flag, filters, filterd;
	ssw_dres* res;
	ssw_vmap vm;
} ssw_capture_args;

/*
 * generic chain kernel (k_chainx): one 16-lane chain per job with its OWN profile; queries of any length are cut
 * into row strips of 16R rows that the chain processes one after the other, handing the bottom boundary
 * (H, F, running column maxima) of a strip to the next through HBM.  mode 0: forward fill of (pair, tile) jobs ->
 * column maxima; mode 1/2: locate / reverse windows of single queries -> best cell (as k_capture).
 */
typedef struct {
	const int8_t* tgt;
	int32_t refLen;
	const int8_t* qcodes;
	const int64_t* qoff;
	const int8_t* mat;
	int32_t n;
	uint32_t gapO2, gapE2;
	int32_t gapE, maxmat;
	int32_t njobs;
	/* fill mode */
	const ssw_pair* pairs;
	int32_t tile, halo, ntiles;
	uint32_t* cm16;
	uint32_t* cm8;
	int64_t cm_stride;
	uint32_t* sg16;          /* optional: [pair][seg_stride] maxima of every aligned group of 16 columns, as in ssw_fill_args */
	uint32_t* sg8;
	int64_t seg_stride;
	/* capture mode */
	const int32_t* qlist;
	int32_t reverse;
	int32_t nlist;           /* capture: entries of qlist (set by the launcher; njobs = pairs of entries) */
	int32_t lanes;           /* lanes per chain: 16 (one DPP row, 4 jobs per wavefront) or 64 (the wavefront is one chain) */
	int32_t window_extra;    /* reverse pass: >= 0 caps the window at rows + rows/4 + window_extra columns (retry uncapped if missed) */
	int32_t* retry_count;    /* incremented for every alignment whose capped window missed */
	int32_t flag, filters, filterd;
	ssw_dres* res;
	/* strip boundary hand-off: njobs regions of bnd_stride records of 4 words (H, F, colmax16, colmax8) */
	uint32_t* bnd;
	int64_t bnd_stride;
	int32_t* cand;           /* fill mode, optional: best cell of every job, [job][half][4] = value, column, row, - */
	/* work-queue form (k_chainq): njobs x strips items drawn from a ticket counter */
	int32_t strips;          /* strips per job of this launch (jobs with fewer strips leave the rest of their items empty) */
	int32_t* queue;          /* [0] ticket counter, [1 + job * strips + strip] completion flags; zeroed before the launch */
	int32_t* err;            /* error word of the call: raised by a wait that timed out, checked by the host before results are handed out */
	int32_t* cand_strip;     /* [job * strips + strip][half][4]: best cell of the job up to and including that strip */
	int32_t form;            /* fill mode: 3 = column frame (fr_base / fr_kmask as in ssw_fill_args), 0 = plain int16 */
	int32_t fr_base, fr_kmask;
	int32_t whole_jobs;      /* 1: a ticket is a whole job (its strips in sequence on one wavefront); 0: a ticket is one strip */
	ssw_vmap vm;             /* capture mode only */
	int32_t banded;          /* capture mode, k_chainq, capped reverse pass: strips walk a diagonal band of the window (accepted only with cap_half_finish's proof) */
	int32_t tail_R;          /* fill mode, k_chainq: > 0: the LAST of the `strips` strips has tail_R (1, 2 or 4) rows per lane instead of R (all jobs of the launch have the same padded length) */
} ssw_chainx_args;

/*
 * literal lane model (k_literal): for gap penalties with gapO <= gapE the reference's answer depends on its striped
 * SIMD layout and on when its lazy-F loop stops, so one DPP row re-enacts one SSE2 register -- 16 unsigned 8-bit
 * lanes (sw_sse2_byte) or 8 signed 16-bit lanes (sw_sse2_word) -- instruction for instruction.
 * pass 0: forward fill (8-bit rules, then 16-bit on overflow) -> score1/ref_end1/read_end1/score2/ref_end2;
 * pass 1: reverse fill with `terminate` -> begin position.
 */
typedef struct {
	const int8_t* tgt;
	int32_t refLen;
	const int8_t* qcodes;
	const int64_t* qoff;
	const int32_t* qlist;
	int32_t nq;
	const int8_t* mat;
	int32_t n;
	int32_t gapO, gapE;
	int32_t pass;
	int32_t maskLen, bias, score_size;
	int32_t flag, filters, filterd;
	ssw_dres* res;
	uint8_t* scratch;        /* nq regions of scratch_stride bytes: [state (state_bytes, unused when it lives in LDS)][maxColumn] */
	int64_t scratch_stride;
	int64_t mc_off;          /* offset of maxColumn in a region (= state_bytes rounded up to 16) */
	int64_t state_bytes;     /* H x2, E, Hmax as [segments][16] int16 + codes, sized for the 16-bit kernel of the longest read */
	int32_t lds_stride;      /* set by the launcher: > 0 = per-alignment state in LDS */
	int32_t* spec_cnt;       /* optional (forward pass, score_size 2): per QUERY a counter, zeroed by the host -- both rule sets run as separate jobs, the later
	                            one writes the record; NULL: the 16-bit kernel follows the 8-bit one where that saturated.  Scratch then holds
	                            ((nq + 3) & ~3) + nq regions */
	int32_t* spec_out;       /* ... and 2 x 8 ints per QUERY: what each of the two jobs found */
} ssw_literal_args;

/* banded traceback */
typedef struct {
	const int8_t* tgt;
	const int8_t* qcodes;
	const int64_t* qoff;
	const int32_t* qlist;
	int32_t nq;
	const int8_t* mat;
	int32_t n;
	int32_t gapO, gapE;
	ssw_dres* res;
	uint8_t* scratch;        /* nq regions of scratch_stride bytes */
	int64_t scratch_stride;
	uint32_t* cigar;         /* nq regions of cigar_stride words */
	int64_t cigar_stride;
	int32_t* need;           /* per job: 0 done, -1 CIGAR slot too small, otherwise scratch that was needed in 4-KiB units;
	                            k_trace_wave also stores the band width that did not fit at need[nq + job] */
	const int64_t* soff;     /* optional: job j owns scratch[soff[j] .. soff[j+1]) instead of a uniform stride */
	int32_t* resume;         /* k_trace_wave: 8 ints per QUERY {band, best, best_i, best_j, stage}; zeroed before round 0 */
	int32_t lds_bytes;       /* k_trace_wave: dynamic LDS per workgroup; bands that fit keep their rows on chip */
	int32_t waves;           /* k_trace_wave: wavefronts working on one alignment (1, 4 or 16): wide bands need the lanes */
	int32_t unblocked;       /* k_trace_wave teams: 1 = one cell per thread and two barriers per 64 x waves cells (the first form; experiments / tests) */
	ssw_vmap vm;
} ssw_trace_args;

/* device-side mark_mismatch: M -> '=' / 'X' runs, soft clips, edit distance */
typedef struct {
	const int8_t* tgt;
	const int8_t* qcodes;
	const int64_t* qoff;
	int32_t nq;
	ssw_dres* res;
	const uint32_t* cigar;   /* slots written by the traceback (res[q].cigar_off) */
	uint32_t* out;           /* nq slots of out_stride words */
	int64_t out_stride;
	ssw_vmap vm;
} ssw_mark_args;

/* compaction of the per-query CIGAR slots into one pool */
typedef struct {
	const uint32_t* src;     /* CIGAR slots (res[q].cigar_off indexes into this) */
	const ssw_dres* res;
	const int64_t* dst_off;  /* per query: first word in dst */
	uint32_t* dst;
	int32_t nq;
} ssw_gather_args;

/* sequence preparation on the device: ASCII -> residue codes, reverse complement of code sequences */
typedef struct {
	const uint8_t* text;     /* mode 0: ASCII residues */
	const int8_t* codes_in;  /* mode 1: codes to reverse-complement */
	const int8_t* table;     /* mode 0: 128-entry translation table */
	const int64_t* off;      /* sequence offsets (count + 1) */
	int32_t count;
	int64_t total;
	int8_t* out;
	int32_t mode;
} ssw_prep_args;

/* device self-test: cross-lane primitive semantics + packed-int16 VALU issue-rate probe */
typedef struct {
	uint32_t* lanes_out;     /* 9 x 64 words, may be NULL */
	uint32_t* sink;          /* one word per thread, may be NULL */
	int32_t iters;           /* each iteration issues 96 packed 16-bit VALU instructions per wavefront */
	uint32_t seed;
} ssw_selftest_args;

/* ---- thin C shim over the HIP runtime + kernel launches (implemented in ssw_kernels.hip) ---- */
int   ssw_shim_device_count(void);
int   ssw_shim_set_device(int dev);
const char* ssw_shim_last_error(void);
void* ssw_shim_stream_create(void);
void* ssw_shim_stream_create_low(void);      /* lowest dispatch priority the device offers */
void  ssw_shim_stream_destroy(void* stream);
int   ssw_shim_stream_sync(void* stream);
void* ssw_shim_malloc(size_t bytes);
void  ssw_shim_free(void* p);
void* ssw_shim_host_alloc(size_t bytes);   /* page-locked host memory */
void  ssw_shim_host_free(void* p);
int   ssw_shim_h2d(void* dst, const void* src, size_t bytes, void* stream);
int   ssw_shim_d2h(void* dst, const void* src, size_t bytes, void* stream);
int   ssw_shim_memset(void* dst, int value, size_t bytes, void* stream);
size_t ssw_shim_mem_free_bytes(void);
int   ssw_shim_device_props(int* compute_units, int* waves_per_cu);   /* of the current device */
void* ssw_shim_event_create(void);
void  ssw_shim_event_destroy(void* ev);
int   ssw_shim_event_record(void* ev, void* stream);
int   ssw_shim_event_sync(void* ev);                         /* host waits for the event */
int   ssw_shim_stream_wait_event(void* stream, void* ev);   /* later work on `stream` waits for `ev` */
float ssw_shim_event_elapsed_ms(void* start, void* stop);   /* both must have completed */

int ssw_shim_launch_fill(int R, const ssw_fill_args* a, void* stream);
int ssw_shim_fill_class(int R);    /* register class of k_fill<R>: sub-launches of one k_fillm grid share it */
int ssw_shim_launch_fillm(const ssw_fillm_args* a, const int32_t* host_R, int n, int form, int64_t total_wgs, void* stream);
int ssw_shim_launch_filldb(int R, const ssw_filldb_args* a, void* stream);
int ssw_shim_launch_reduce(const ssw_reduce_args* a, void* stream);
int ssw_shim_launch_reducem(const ssw_reducem_args* a, int64_t total_pairs, void* stream);
int ssw_shim_launch_capture(int R, const ssw_capture_args* a, void* stream);
int64_t ssw_shim_capture_lds_need(int R, int n);   /* dynamic LDS of one k_capture<R> workgroup */
int ssw_shim_launch_chainx(int R, int capture, const ssw_chainx_args* a, void* stream);
int ssw_shim_launch_chainq(int R, int capture, const ssw_chainx_args* a, int max_workgroups, void* stream);   /* 64-lane chains behind a work queue */
int ssw_shim_chainq_resident(int R, int capture, int n);   /* wavefronts of k_chainq<R> the device holds at once (0: unknown) */
int ssw_shim_launch_literal(const ssw_literal_args* a, void* stream);
int ssw_shim_launch_trace(const ssw_trace_args* a, void* stream);
int ssw_shim_launch_trace_wave(const ssw_trace_args* a, void* stream);
int ssw_shim_launch_trace_diag(int team, const ssw_trace_args* a, void* stream);   /* narrow bands: teams of 16 / 32 lanes, several alignments per wavefront */
int64_t ssw_shim_trace_lds_need(int band_width, int waves);   /* LDS that keeps a band of this width on chip */   /* one wavefront per alignment (long reads) */
int ssw_shim_launch_gather(const ssw_gather_args* a, void* stream);
int ssw_shim_launch_select(const ssw_select_args* a, void* stream);   /* the pass in a->pass */
int ssw_shim_launch_mark(const ssw_mark_args* a, void* stream);
int ssw_shim_launch_prep(const ssw_prep_args* a, void* stream);
int ssw_shim_launch_selftest(const ssw_selftest_args* a, int blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSW_DEV_H */
