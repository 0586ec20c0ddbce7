/*
 * oracle/ref_wrap.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Builds the *unmodified* reference implementation from where it lies under
 * /root/reference/src (path injected by oracle/Makefile as REF_SSW_C) into
 * oracle/_ref/libssw_ref.so.  No reference source is copied into this repo:
 * the translation unit below is a single #include of the reference file plus
 * thin exported shims around its `static` kernels so that tests can do
 * per-kernel differential checks:
 *
 *   refwrap_sw_byte   -> sw_sse2_byte   (reference src/ssw.c:197-386)
 *   refwrap_sw_word   -> sw_sse2_word   (reference src/ssw.c:412-588)
 *   refwrap_banded_sw -> banded_sw      (reference src/ssw.c:590-783)
 *
 * The public ssw.h functions (ssw_init / ssw_align / init_destroy /
 * align_destroy / mark_mismatch) are exported by the included file itself.
 */
#ifndef REF_SSW_C
#error "build through oracle/Makefile (REF_SSW_C must point at the reference ssw.c)"
#endif
#include REF_SSW_C

#include <pthread.h>
#include <time.h>

/* flat result: score1, ref1, read1, score2, ref2 */
void refwrap_sw_byte(const int8_t* ref, int8_t ref_dir, int32_t refLen,
                     const int8_t* read, int32_t readLen, const int8_t* mat, int32_t n,
                     uint8_t gapO, uint8_t gapE, uint8_t terminate, uint8_t bias,
                     int32_t maskLen, int32_t* out5)
{
	__m128i* vP = qP_byte(read, mat, readLen, n, bias);
	alignment_end* b = sw_sse2_byte(ref, ref_dir, refLen, readLen, gapO, gapE, vP, terminate, bias, maskLen);
	out5[0] = b[0].score; out5[1] = b[0].ref; out5[2] = b[0].read;
	out5[3] = b[1].score; out5[4] = b[1].ref;
	free(b); free(vP);
}

void refwrap_sw_word(const int8_t* ref, int8_t ref_dir, int32_t refLen,
                     const int8_t* read, int32_t readLen, const int8_t* mat, int32_t n,
                     uint8_t gapO, uint8_t gapE, uint16_t terminate,
                     int32_t maskLen, int32_t* out5)
{
	__m128i* vP = qP_word(read, mat, readLen, n);
	alignment_end* b = sw_sse2_word(ref, ref_dir, refLen, readLen, gapO, gapE, vP, terminate, maskLen);
	out5[0] = b[0].score; out5[1] = b[0].ref; out5[2] = b[0].read;
	out5[3] = b[1].score; out5[4] = b[1].ref;
	free(b); free(vP);
}

/* returns cigar length, -1 when banded_sw returned NULL; cigar copied into out (cap words) */
int32_t refwrap_banded_sw(const int8_t* ref, const int8_t* read, int32_t refLen, int32_t readLen,
                          int32_t score, uint32_t gapO, uint32_t gapE, int32_t band_width,
                          const int8_t* mat, int32_t n, uint32_t* out, int32_t cap)
{
	cigar* c = banded_sw(ref, read, refLen, readLen, score, gapO, gapE, band_width, mat, n);
	if (!c) return -1;
	int32_t len = c->length;
	for (int32_t i = 0; i < len && i < cap; ++i) out[i] = c->seq[i];
	free(c->seq); free(c);
	return len;
}

/*
 * CPU baseline driver (bench.py "cpu_baseline", kind = "reference"): runs
 * ssw_init(...,2) + ssw_align + destroy for queries [0,nq) against ONE target
 * on `nthreads` host threads (the library is re-entrant), and
 * returns wall-clock seconds.  Results are written to res (10 int32 per query:
 * score1 score2 ref_begin1 ref_end1 read_begin1 read_end1 ref_end2 cigarLen flag isnull).
 */
typedef struct {
	const int8_t* qcodes; const int64_t* qoff; int32_t nq;
	const int8_t* ref; int32_t refLen; const int8_t* mat; int32_t n;
	uint8_t gapO, gapE, flag; uint16_t filters; int32_t filterd; int32_t maskLen;
	int32_t* res; int tid, nthreads;
	uint32_t* cig_hash;                                  /* optional: FNV-1a over the CIGAR words of every alignment */
	const int8_t* tcodes; const int64_t* toff; int32_t nt;   /* database mode: several targets per query */
} refwrap_job;

static uint32_t refwrap_fnv(const uint32_t* w, int32_t n)
{
	uint32_t h = 2166136261u;
	for (int32_t i = 0; i < n; ++i) for (int b = 0; b < 4; ++b) { h ^= (w[i] >> (8 * b)) & 0xffu; h *= 16777619u; }
	return h;
}

static void* refwrap_worker(void* arg)
{
	refwrap_job* j = (refwrap_job*)arg;
	for (int32_t q = j->tid; q < j->nq; q += j->nthreads) {
		const int8_t* rd = j->qcodes + j->qoff[q];
		int32_t len = (int32_t)(j->qoff[q + 1] - j->qoff[q]);
		int32_t maskLen = j->maskLen >= 0 ? j->maskLen : len / 2;
		s_profile* p = ssw_init(rd, len, j->mat, j->n, 2);
		s_align* a = ssw_align(p, j->ref, j->refLen, j->gapO, j->gapE, j->flag, j->filters, j->filterd, maskLen);
		int32_t* r = j->res + (int64_t)q * 10;
		if (a) {
			r[0] = a->score1; r[1] = a->score2; r[2] = a->ref_begin1; r[3] = a->ref_end1;
			r[4] = a->read_begin1; r[5] = a->read_end1; r[6] = a->ref_end2; r[7] = a->cigarLen;
			r[8] = a->flag; r[9] = 0;
			if (j->cig_hash) j->cig_hash[q] = a->cigarLen > 0 && a->cigar ? refwrap_fnv(a->cigar, a->cigarLen) : 0u;
			align_destroy(a);
		} else { for (int k = 0; k < 9; ++k) r[k] = 0; r[9] = 1; if (j->cig_hash) j->cig_hash[q] = 0u; }
		init_destroy(p);
	}
	return 0;
}

/* database mode (the loop of reference src/main.c:462-526): one profile per query, every target; score-only records of
   5 int32 (score1 score2 ref_end1 read_end1 ref_end2) at res[(q * nt + t) * 5] */
static void* refwrap_db_worker(void* arg)
{
	refwrap_job* j = (refwrap_job*)arg;
	for (int32_t q = j->tid; q < j->nq; q += j->nthreads) {
		const int8_t* rd = j->qcodes + j->qoff[q];
		int32_t len = (int32_t)(j->qoff[q + 1] - j->qoff[q]);
		int32_t maskLen = j->maskLen >= 0 ? j->maskLen : len / 2;
		s_profile* p = ssw_init(rd, len, j->mat, j->n, 2);
		for (int32_t t = 0; t < j->nt; ++t) {
			s_align* a = ssw_align(p, j->tcodes + j->toff[t], (int32_t)(j->toff[t + 1] - j->toff[t]), j->gapO, j->gapE, 0, 0, 0, maskLen);
			int32_t* r = j->res + ((int64_t)q * j->nt + t) * 5;
			if (a) { r[0] = a->score1; r[1] = a->score2; r[2] = a->ref_end1; r[3] = a->read_end1; r[4] = a->ref_end2; align_destroy(a); }
			else { r[0] = r[1] = r[2] = r[3] = r[4] = -9; }
		}
		init_destroy(p);
	}
	return 0;
}

/* database mode WITH the caller's flag / filters (the reference's loop calls ssw_align with flag 2 and a score filter for every
   (read, target) pair: src/main.c:493-506): full records of 10 int32 at res[(q * nt + t) * 10] as in refwrap_worker, and the
   FNV-1a of every CIGAR at cig_hash[q * nt + t] */
static void* refwrap_dbx_worker(void* arg)
{
	refwrap_job* j = (refwrap_job*)arg;
	for (int32_t q = j->tid; q < j->nq; q += j->nthreads) {
		const int8_t* rd = j->qcodes + j->qoff[q];
		int32_t len = (int32_t)(j->qoff[q + 1] - j->qoff[q]);
		int32_t maskLen = j->maskLen >= 0 ? j->maskLen : len / 2;
		s_profile* p = ssw_init(rd, len, j->mat, j->n, 2);
		for (int32_t t = 0; t < j->nt; ++t) {
			s_align* a = ssw_align(p, j->tcodes + j->toff[t], (int32_t)(j->toff[t + 1] - j->toff[t]), j->gapO, j->gapE, j->flag, j->filters, j->filterd, maskLen);
			int32_t* r = j->res + ((int64_t)q * j->nt + t) * 10;
			uint32_t h = 0u;
			if (a) {
				r[0] = a->score1; r[1] = a->score2; r[2] = a->ref_begin1; r[3] = a->ref_end1;
				r[4] = a->read_begin1; r[5] = a->read_end1; r[6] = a->ref_end2; r[7] = a->cigarLen;
				r[8] = a->flag; r[9] = 0;
				if (a->cigarLen > 0 && a->cigar) h = refwrap_fnv(a->cigar, a->cigarLen);
				align_destroy(a);
			} else { for (int k = 0; k < 9; ++k) r[k] = 0; r[9] = 1; }
			if (j->cig_hash) j->cig_hash[(int64_t)q * j->nt + t] = h;
		}
		init_destroy(p);
	}
	return 0;
}

static double refwrap_run(refwrap_job proto, int32_t nthreads, void* (*worker)(void*));

double refwrap_bench_dbx(const int8_t* qcodes, const int64_t* qoff, int32_t nq,
                         const int8_t* tcodes, const int64_t* toff, int32_t nt, const int8_t* mat, int32_t n,
                         uint8_t gapO, uint8_t gapE, uint8_t flag, uint16_t filters, int32_t filterd, int32_t maskLen, int32_t nthreads,
                         int32_t* res10, uint32_t* hash)
{
	refwrap_job j = { qcodes, qoff, nq, 0, 0, mat, n, gapO, gapE, flag, filters, filterd, maskLen, res10, 0, nthreads, hash, tcodes, toff, nt };
	return refwrap_run(j, nthreads, refwrap_dbx_worker);
}

double refwrap_bench(const int8_t* qcodes, const int64_t* qoff, int32_t nq,
                     const int8_t* ref, int32_t refLen, const int8_t* mat, int32_t n,
                     uint8_t gapO, uint8_t gapE, uint8_t flag, uint16_t filters, int32_t filterd,
                     int32_t maskLen, int32_t nthreads, int32_t* res)
{
	refwrap_job j = { qcodes, qoff, nq, ref, refLen, mat, n, gapO, gapE, flag, filters, filterd, maskLen, res, 0, nthreads, 0, 0, 0, 0 };
	return refwrap_run(j, nthreads, refwrap_worker);
}

/* same, additionally hash[q] = FNV-1a of the CIGAR words (0: no CIGAR) */
double refwrap_bench_hash(const int8_t* qcodes, const int64_t* qoff, int32_t nq,
                          const int8_t* ref, int32_t refLen, const int8_t* mat, int32_t n,
                          uint8_t gapO, uint8_t gapE, uint8_t flag, uint16_t filters, int32_t filterd,
                          int32_t maskLen, int32_t nthreads, int32_t* res, uint32_t* hash)
{
	refwrap_job j = { qcodes, qoff, nq, ref, refLen, mat, n, gapO, gapE, flag, filters, filterd, maskLen, res, 0, nthreads, hash, 0, 0, 0 };
	return refwrap_run(j, nthreads, refwrap_worker);
}

double refwrap_bench_db(const int8_t* qcodes, const int64_t* qoff, int32_t nq,
                        const int8_t* tcodes, const int64_t* toff, int32_t nt, const int8_t* mat, int32_t n,
                        uint8_t gapO, uint8_t gapE, int32_t maskLen, int32_t nthreads, int32_t* res5)
{
	refwrap_job j = { qcodes, qoff, nq, 0, 0, mat, n, gapO, gapE, 0, 0, 0, maskLen, res5, 0, nthreads, 0, tcodes, toff, nt };
	return refwrap_run(j, nthreads, refwrap_db_worker);
}

static double refwrap_run(refwrap_job proto, int32_t nthreads, void* (*worker)(void*))
{
	if (nthreads < 1) nthreads = 1;
	pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
	refwrap_job* jobs = (refwrap_job*)malloc(sizeof(refwrap_job) * nthreads);
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int t = 0; t < nthreads; ++t) {
		jobs[t] = proto; jobs[t].tid = t; jobs[t].nthreads = nthreads;
		pthread_create(&th[t], 0, worker, &jobs[t]);
	}
	for (int t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(th); free(jobs);
	return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
