/*
 * ssw_pool.c -- several devices, one batch: the per-GPU work queues of SURVEY 8e (north_star: "batches of independent
 * query x reference alignments shard trivially across the 8 GPUs of one node; no RCCL needed; per-GPU work queues only").
 *
 * What it replaces in the reference: the read loop of src/main.c:462-526 (one ssw_init + one ssw_align per target per read,
 * on one core).  Here a pool owns one worker per device -- a context (its own streams and workspaces, include/ssw_gpu.h)
 * plus one host thread for the duration of a call -- the target set is uploaded once to every device, and read-index
 * blocks are handed out through one atomic counter.  Records go to per-query slots, CIGAR pools are concatenated in block
 * order afterwards, so the output is identical to a single-device ssw_gpu_align_batch over the same reads.
 * Uses only the public batch ABI; host code, plain C.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ssw_gpu.h"

typedef struct {
	ssw_gpu_ctx* ctx;
	ssw_gpu_seqs* targets;
	int device;
	ssw_gpu_pool_stat st;
} pool_worker;

struct ssw_gpu_pool {
	int n;
	pool_worker* w;
	int32_t tcount;
	char err[512];
};

typedef struct { uint32_t* words; int64_t n; } block_cigars;

/* one ssw_gpu_pool_align call, shared by its worker threads */
typedef struct {
	ssw_gpu_pool* pool;
	const int8_t* qcodes; const int64_t* qoff; int32_t nq, block, nblocks;
	int32_t tfirst, tcount;
	const ssw_gpu_params* prm;
	ssw_gpu_result* results;
	block_cigars* cig;           /* per block */
	int want_cigars;
	int32_t next;                /* the queue: next block index (atomic) */
	int failed;                  /* first failure stops the hand-out (atomic) */
	pthread_mutex_t err_lock;
} pool_call;

typedef struct { pool_call* call; int widx; } pool_thread_arg;

static void* pool_thread(void* argp)
{
	pool_thread_arg* a = (pool_thread_arg*)argp;
	pool_call* k = a->call;
	pool_worker* w = &k->pool->w[a->widx];
	for (;;) {
		if (__atomic_load_n(&k->failed, __ATOMIC_ACQUIRE)) break;
		const int32_t b = __atomic_fetch_add(&k->next, 1, __ATOMIC_RELAXED);
		if (b >= k->nblocks) break;
		const int32_t q0 = b * k->block, cnt = k->nq - q0 < k->block ? k->nq - q0 : k->block;
		ssw_gpu_seqs* Q = ssw_gpu_seqs_upload(w->ctx, k->qcodes, k->qoff + q0, cnt);   /* offsets are taken relative to their first entry */
		int rc = Q ? 0 : -1;
		if (Q) {
			rc = ssw_gpu_align_batch(w->ctx, Q, w->targets, k->tfirst, k->tcount, k->prm, k->results + (int64_t)q0 * k->tcount,
			                         k->want_cigars ? &k->cig[b].words : 0, k->want_cigars ? &k->cig[b].n : 0);
			ssw_gpu_seqs_free(Q);
		}
		if (rc) {
			pthread_mutex_lock(&k->err_lock);
			if (!__atomic_exchange_n(&k->failed, 1, __ATOMIC_ACQ_REL))
				snprintf(k->pool->err, sizeof k->pool->err, "worker %d (device %d), reads %d..%d: %s", a->widx, w->device, q0, q0 + cnt - 1, ssw_gpu_last_error(w->ctx));
			pthread_mutex_unlock(&k->err_lock);
			break;
		}
		ssw_gpu_timing tm;
		if (ssw_gpu_last_timing(w->ctx, &tm) == 0) { w->st.cells += tm.cells; w->st.busy_ms += tm.total_ms; }
		w->st.blocks++; w->st.queries += cnt;
	}
	return 0;
}

ssw_gpu_pool* ssw_gpu_pool_open(const int* devices, int n)
{
	const int ndev = ssw_gpu_device_count();
	if (ndev < 1) { (void)ssw_gpu_open(0); return 0; }          /* leaves the "no device" message in ssw_gpu_last_error(NULL) */
	if (!devices) n = ndev;
	if (n < 1 || n > 1024) return 0;
	ssw_gpu_pool* p = (ssw_gpu_pool*)calloc(1, sizeof *p);
	if (!p) return 0;
	p->w = (pool_worker*)calloc((size_t)n, sizeof(pool_worker));
	if (!p->w) { free(p); return 0; }
	p->n = n;
	for (int i = 0; i < n; ++i) {
		p->w[i].device = devices ? devices[i] : i;
		p->w[i].st.device = p->w[i].device;
		p->w[i].ctx = ssw_gpu_open(p->w[i].device);
		if (!p->w[i].ctx) { ssw_gpu_pool_close(p); return 0; }   /* the reason stays in ssw_gpu_last_error(NULL) */
	}
	/* workers that share a device share its HBM: every context sized its scratch budget from what was free when IT was opened
	   (half of it) -- divide by the number of workers on that device (an explicit SSW_GPU_CM_BUDGET_MB is per context and kept) */
	if (!getenv("SSW_GPU_CM_BUDGET_MB"))
		for (int i = 0; i < n; ++i) {
			int same = 0;
			for (int j = 0; j < n; ++j) if (p->w[j].device == p->w[i].device) ++same;
			if (same > 1) ssw_gpu_set_budget(p->w[i].ctx, ssw_gpu_get_budget(p->w[i].ctx) / (size_t)same);
		}
	return p;
}

void ssw_gpu_pool_close(ssw_gpu_pool* p)
{
	if (!p) return;
	for (int i = 0; i < p->n; ++i) {
		if (p->w[i].targets) ssw_gpu_seqs_free(p->w[i].targets);
		if (p->w[i].ctx) ssw_gpu_close(p->w[i].ctx);
	}
	free(p->w); free(p);
}

int ssw_gpu_pool_size(const ssw_gpu_pool* p) { return p ? p->n : 0; }
size_t ssw_gpu_pool_budget(const ssw_gpu_pool* p, int worker) { return p && worker >= 0 && worker < p->n ? ssw_gpu_get_budget(p->w[worker].ctx) : 0; }
const char* ssw_gpu_pool_last_error(const ssw_gpu_pool* p) { return p ? p->err : ssw_gpu_last_error(0); }

int ssw_gpu_pool_set_targets(ssw_gpu_pool* p, const int8_t* codes, const int64_t* offsets, int32_t count)
{
	if (!p || !offsets || count < 0) return -1;
	for (int i = 0; i < p->n; ++i) {
		if (p->w[i].targets) { ssw_gpu_seqs_free(p->w[i].targets); p->w[i].targets = 0; }
		p->w[i].targets = ssw_gpu_seqs_upload(p->w[i].ctx, codes, offsets, count);
		if (!p->w[i].targets) {
			snprintf(p->err, sizeof p->err, "worker %d (device %d): %s", i, p->w[i].device, ssw_gpu_last_error(p->w[i].ctx));
			/* all or nothing: a pool with the new targets on some workers and the old count would answer with mixed data */
			for (int j = 0; j < p->n; ++j) if (p->w[j].targets) { ssw_gpu_seqs_free(p->w[j].targets); p->w[j].targets = 0; }
			p->tcount = 0;
			return -1;
		}
	}
	p->tcount = count;
	return 0;
}

int ssw_gpu_pool_stats(const ssw_gpu_pool* p, int worker, ssw_gpu_pool_stat* out)
{
	if (!p || !out || worker < 0 || worker >= p->n) return -1;
	*out = p->w[worker].st;
	return 0;
}

int ssw_gpu_pool_align(ssw_gpu_pool* p, const int8_t* qcodes, const int64_t* qoffsets, int32_t nq, int32_t block,
                       int32_t target_first, int32_t target_count, const ssw_gpu_params* prm,
                       ssw_gpu_result* results, uint32_t** cigar_pool, int64_t* cigar_words)
{
	if (!p) return -1;
	if (cigar_pool) *cigar_pool = 0;
	if (cigar_words) *cigar_words = 0;
	if (!qoffsets || nq < 0 || !prm || !results) { snprintf(p->err, sizeof p->err, "pool_align: bad arguments"); return -1; }
	if (!p->w[0].targets) { snprintf(p->err, sizeof p->err, "pool_align: no target set (ssw_gpu_pool_set_targets)"); return -1; }
	if (target_first < 0 || target_count < 0 || target_first + target_count > p->tcount) { snprintf(p->err, sizeof p->err, "pool_align: target range out of bounds"); return -1; }
	for (int i = 0; i < p->n; ++i) { const int dev = p->w[i].st.device; memset(&p->w[i].st, 0, sizeof p->w[i].st); p->w[i].st.device = dev; }
	if (nq == 0 || target_count == 0) return 0;
	if (block < 1) {      /* a few blocks per worker: the tail is one block, the upload of a block hides behind the others' compute */
		block = (nq + 4 * p->n - 1) / (4 * p->n);
		if (block < 256) block = 256;
	}
	if (block > nq) block = nq;
	pool_call k; memset(&k, 0, sizeof k);
	k.pool = p; k.qcodes = qcodes; k.qoff = qoffsets; k.nq = nq; k.block = block; k.nblocks = (nq + block - 1) / block;
	k.tfirst = target_first; k.tcount = target_count; k.prm = prm; k.results = results;
	k.want_cigars = cigar_pool != 0;
	k.cig = (block_cigars*)calloc((size_t)k.nblocks, sizeof(block_cigars));
	pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)p->n);
	pool_thread_arg* ta = (pool_thread_arg*)malloc(sizeof(pool_thread_arg) * (size_t)p->n);
	if (!k.cig || !th || !ta) { free(k.cig); free(th); free(ta); snprintf(p->err, sizeof p->err, "out of host memory"); return -1; }
	pthread_mutex_init(&k.err_lock, 0);
	int started = 0;
	for (int i = 0; i < p->n; ++i) {
		ta[i].call = &k; ta[i].widx = i;
		if (pthread_create(&th[i], 0, pool_thread, &ta[i])) break;
		++started;
	}
	if (started == 0) { __atomic_store_n(&k.failed, 1, __ATOMIC_RELEASE); snprintf(p->err, sizeof p->err, "pthread_create failed"); }
	for (int i = 0; i < started; ++i) pthread_join(th[i], 0);      /* (fewer threads than workers still drain the whole queue) */
	pthread_mutex_destroy(&k.err_lock);
	int rc = k.failed ? -1 : 0;
	if (rc == 0 && k.want_cigars) {   /* one pool in block order; a record's cigar_off was relative to its block's pool */
		int64_t total = 0;
		for (int32_t b = 0; b < k.nblocks; ++b) total += k.cig[b].n;
		uint32_t* all = total > 0 ? (uint32_t*)malloc(sizeof(uint32_t) * (size_t)total) : 0;
		if (total > 0 && !all) { snprintf(p->err, sizeof p->err, "out of host memory (CIGAR pool)"); rc = -1; }
		else {
			int64_t base = 0;
			for (int32_t b = 0; b < k.nblocks; ++b) {
				if (k.cig[b].n > 0) {
					memcpy(all + base, k.cig[b].words, sizeof(uint32_t) * (size_t)k.cig[b].n);
					const int32_t q0 = b * block, cnt = nq - q0 < block ? nq - q0 : block;
					ssw_gpu_result* r = results + (int64_t)q0 * target_count;
					if (base > 0) for (int64_t i = 0; i < (int64_t)cnt * target_count; ++i) if (r[i].cigar_off >= 0) r[i].cigar_off += base;
				}
				base += k.cig[b].n;
			}
			*cigar_pool = all;
			if (cigar_words) *cigar_words = total;
		}
	}
	for (int32_t b = 0; b < k.nblocks; ++b) free(k.cig[b].words);
	free(k.cig); free(th); free(ta);
	return rc;
}
