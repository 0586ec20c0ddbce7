/*
 * oracle/ssw_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, scalar CPU restatement of the reference's hot path
 * (mengyao/Complete-Striped-Smith-Waterman-Library, src/ssw.c v1.2.6).  It is the
 * checker that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * compare the HIP path against.  Nothing in the product library
 * (complete-striped-smith-waterman-library_amd/) may include, link or call
 * this file.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every function
 * here against the unmodified reference compiled into oracle/_ref/libssw_ref.so
 * (oracle/Makefile, oracle/ref_wrap.c) on the reference's demo fixtures and on
 * seeded random/adversarial batches, and tests/golden/ holds vectors generated
 * from that reference build (tests/golden/make_golden.py).
 *
 * Two models live here:
 *
 *  (1) "lane model"  -- orc_striped(): emulates the two SSE2 kernels lane by
 *      lane (16 unsigned 8-bit lanes / 8 signed 16-bit lanes), including the
 *      striped query layout, saturation, the lazy-F loop and its early exit.
 *      It is exact for every parameter combination, also gapO <= gapE where
 *      the result depends on the stripe layout.
 *      follows: sw_sse2_byte ssw.c:197-386, sw_sse2_word ssw.c:412-588,
 *               qP_byte ssw.c:163-188, qP_word ssw.c:388-410.
 *
 *  (2) "plain model" -- orc_plain_*(): the layout-independent affine-gap
 *      recurrence over the zero-padded query (P rows) that the HIP kernels
 *      implement; equal to (1) in every observable when gapO > gapE
 *      (tests/test_oracle_models.py).  It also carries the column-tiling with
 *      an exact halo that the GPU fill kernel uses.
 *
 * plus the scalar pieces of the driver:
 *      orc_banded()       <- banded_sw ssw.c:590-783
 *      orc_cigar_score()  <- cigar_alignment_score ssw.c:785-811
 *      orc_align()        <- ssw_init ssw.c:826-847 + ssw_align ssw.c:855-977
 *      orc_mark_mismatch()<- mark_mismatch ssw.c:1019-1074
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <limits.h>

typedef struct {
	int32_t score1, score2;
	int32_t ref_begin1, ref_end1, read_begin1, read_end1, ref_end2;
	int32_t cigarLen;
	int32_t flag;
	int32_t is_null;     /* 1 when the reference would have returned NULL */
	int32_t used_word;   /* 1 when the result came from the 16-bit kernel  */
} orc_result;

typedef struct { int32_t score, ref, read; } orc_end;

/* ------------------------------------------------------------------ */
/* (1) lane model                                                      */
/* ------------------------------------------------------------------ */

typedef struct {
	int lanes;        /* 16 or 8 */
	int is_byte;
	int bias;
} lane_cfg;

static inline int lm_add_score(const lane_cfg* c, int h, int s)
{
	if (c->is_byte) {             /* adds_epu8 with (s + bias), then subs_epu8 bias: ssw.c:275-276 */
		int v = h + s + c->bias; if (v > 255) v = 255;
		v -= c->bias; if (v < 0) v = 0;
		return v;
	} else {                      /* adds_epi16: ssw.c:483 */
		int v = h + s; if (v > 32767) v = 32767; if (v < -32768) v = -32768;
		return v;
	}
}
static inline int lm_sub(const lane_cfg* c, int a, int b)
{
	if (c->is_byte) { int v = a - b; return v < 0 ? 0 : v; }          /* subs_epu8 */
	else { unsigned ua = (uint16_t)a, ub = (uint16_t)b;                /* subs_epu16 */
	       unsigned r = ua > ub ? ua - ub : 0; return (int16_t)(uint16_t)r; }
}
static inline int lm_max(int a, int b) { return a > b ? a : b; }

/*
 * One call == one call of sw_sse2_byte / sw_sse2_word.
 * ends[0] = best (score, ref, read), ends[1] = second best (score, ref).
 * colmax (optional, refLen entries) receives the reference's maxColumn[].
 */
void orc_striped(int is_byte, const int8_t* ref, int ref_dir, int32_t refLen,
                 const int8_t* read, int32_t readLen, const int8_t* mat, int32_t n,
                 int gapO, int gapE, int terminate, int bias, int32_t maskLen,
                 orc_end ends[2], int32_t* colmax)
{
	lane_cfg cfg; cfg.is_byte = is_byte; cfg.lanes = is_byte ? 16 : 8; cfg.bias = is_byte ? bias : 0;
	const int L = cfg.lanes;
	const int32_t segLen = (readLen + L - 1) / L;
	const int32_t cells = segLen * L;
	int* Hst = (int*)calloc(cells, sizeof(int));   /* pvHStore: index j*L + k */
	int* Hld = (int*)calloc(cells, sizeof(int));   /* pvHLoad */
	int* E   = (int*)calloc(cells, sizeof(int));
	int* Hmx = (int*)calloc(cells, sizeof(int));
	int32_t* mc = (int32_t*)calloc(refLen > 0 ? refLen : 1, sizeof(int32_t));
	int* vF = (int*)malloc(L * sizeof(int));
	int* vH = (int*)malloc(L * sizeof(int));
	int* vMaxCol = (int*)malloc(L * sizeof(int));
	int32_t max = 0, end_read = readLen - 1, end_ref = is_byte ? -1 : 0;
	int32_t i, j, k, begin = 0, end = refLen, step = 1;
	if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; }

	for (i = begin; i != end; i += step) {
		/* vH = last segment of the previous column, shifted up one lane */
		for (k = L - 1; k > 0; --k) vH[k] = Hst[(segLen - 1) * L + k - 1];
		vH[0] = 0;
		for (k = 0; k < L; ++k) { vF[k] = 0; vMaxCol[k] = 0; }
		{ int* t = Hld; Hld = Hst; Hst = t; }
		const int8_t* mrow = mat + (int32_t)ref[i] * n;
		for (j = 0; j < segLen; ++j) {
			for (k = 0; k < L; ++k) {
				int32_t q = j + k * segLen;
				int s = q < readLen ? mrow[read[q]] : 0;
				int h = lm_add_score(&cfg, vH[k], s);
				int e = E[j * L + k];
				h = lm_max(h, e);
				h = lm_max(h, vF[k]);
				vMaxCol[k] = lm_max(vMaxCol[k], h);
				Hst[j * L + k] = h;
				h = lm_sub(&cfg, h, gapO);
				e = lm_sub(&cfg, e, gapE);
				e = lm_max(e, h);
				E[j * L + k] = e;
				vF[k] = lm_max(lm_sub(&cfg, vF[k], gapE), h);
				vH[k] = Hld[j * L + k];
			}
		}
		/* lazy-F loop: ssw.c:302-315 / 509-520 */
		int done = 0;
		for (k = 0; k < L && !done; ++k) {
			int kk;
			for (kk = L - 1; kk > 0; --kk) vF[kk] = vF[kk - 1];
			vF[0] = 0;
			for (j = 0; j < segLen; ++j) {
				int all_le = 1;
				for (kk = 0; kk < L; ++kk) {
					int h = lm_max(Hst[j * L + kk], vF[kk]);
					vMaxCol[kk] = lm_max(vMaxCol[kk], h);
					Hst[j * L + kk] = h;
					h = lm_sub(&cfg, h, gapO);
					vF[kk] = lm_sub(&cfg, vF[kk], gapE);
					if (vF[kk] > h) all_le = 0;
				}
				if (all_le) { done = 1; break; }
			}
		}
		/* column maximum; the reference's vMaxScore/vMaxMark bookkeeping
		   (ssw.c:318-335) is equivalent to "did the column max beat the running max" */
		int cm = 0;
		for (k = 0; k < L; ++k) cm = lm_max(cm, vMaxCol[k]);
		if (cm > max) {
			max = cm;
			if (is_byte && max + bias >= 255) break;   /* overflow: ssw.c:329 */
			end_ref = i;
			memcpy(Hmx, Hst, cells * sizeof(int));
		}
		mc[i] = cm;
		if (cm == terminate) break;
	}

	for (i = 0; i < cells; ++i) {
		if (Hmx[i] == max) {
			int32_t row = i / L + (i % L) * segLen;
			if (row < end_read) end_read = row;
		}
	}
	ends[0].score = (is_byte && max + bias >= 255) ? 255 : max;
	ends[0].ref = end_ref; ends[0].read = end_read;
	ends[1].score = 0; ends[1].ref = 0; ends[1].read = 0;
	int32_t edge = (end_ref - maskLen) > 0 ? (end_ref - maskLen) : 0;
	for (i = 0; i < edge; ++i)
		if (mc[i] > ends[1].score) { ends[1].score = mc[i]; ends[1].ref = i; }
	edge = (end_ref + maskLen) > refLen ? refLen : (end_ref + maskLen);
	for (i = is_byte ? edge + 1 : edge; i < refLen; ++i)      /* ssw.c:376 vs 578 */
		if (mc[i] > ends[1].score) { ends[1].score = mc[i]; ends[1].ref = i; }
	if (colmax) memcpy(colmax, mc, (size_t)refLen * sizeof(int32_t));
	free(Hst); free(Hld); free(E); free(Hmx); free(mc); free(vF); free(vH); free(vMaxCol);
}

/* ------------------------------------------------------------------ */
/* (2) plain model (what the HIP kernels compute)                       */
/* ------------------------------------------------------------------ */

static inline int pl_addsat(int a, int b) { int v = a + b; return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
static inline int pl_subsat(int a, int b) { int v = a - b; return v < 0 ? 0 : v; }

/*
 * Columns [c_begin, c_end) of the target, traversed forward (dir 0) or as
 * "column t of the traversal == ref[c_end-1-t]" (dir 1), starting from the
 * all-zero state at c_begin (dir 0) / c_end-1 (dir 1).
 * Rows 0..P-1: rows >= readLen score 0 against every residue.
 * colmax[t] for every traversed column t (0-based in traversal order), and the
 * best cell in traversal order: first column reaching the final maximum,
 * smallest row in it.  stop_at >= 0: stop after the first column whose
 * maximum equals stop_at (the reference's `terminate`).
 * Returns the number of columns traversed.
 */
int32_t orc_plain_fill(const int8_t* ref, int32_t c_begin, int32_t c_end, int dir,
                       const int8_t* read, int32_t readLen, int32_t P,
                       const int8_t* mat, int32_t n, int gapO, int gapE,
                       int stop_at, int32_t* colmax, orc_end* best)
{
	int* H = (int*)calloc(P + 1, sizeof(int));
	int* E = (int*)calloc(P + 1, sizeof(int));
	int32_t t, q, ncol = c_end - c_begin, done = 0;
	int32_t bscore = 0, bcol = -1, brow = -1;
	for (t = 0; t < ncol; ++t) {
		int32_t c = dir ? c_end - 1 - t : c_begin + t;
		const int8_t* mrow = mat + (int32_t)ref[c] * n;
		int diag = 0, f = 0, cm = 0, cmrow = -1;
		for (q = 0; q < P; ++q) {
			int s = q < readLen ? mrow[read[q]] : 0;
			int h0 = pl_addsat(diag, s);
			if (h0 < E[q]) h0 = E[q];          /* E >= 0 supplies the max(0, .) */
			int h = h0 > f ? h0 : f;
			int t0 = pl_subsat(h0, gapO);      /* E and F both extend from h0: see DESIGN.md */
			int e = pl_subsat(E[q], gapE); E[q] = e > t0 ? e : t0;
			int ff = pl_subsat(f, gapE);  f = ff > t0 ? ff : t0;
			diag = H[q]; H[q] = h;
			if (h > cm) { cm = h; cmrow = q; }
		}
		if (colmax) colmax[t] = cm;
		if (cm > bscore) { bscore = cm; bcol = t; brow = cmrow; }
		done = t + 1;
		if (stop_at >= 0 && cm == stop_at) break;
	}
	if (best) { best->score = bscore; best->ref = bcol; best->read = brow; }
	free(H); free(E);
	return done;
}

/* halo width after which the DP state is independent of earlier columns (DESIGN.md "exact halo") */
int32_t orc_plain_halo(int32_t P, const int8_t* mat, int32_t n, int gapE)
{
	int mx = 0;
	for (int32_t i = 0; i < n * n; ++i) if (mat[i] > mx) mx = mat[i];
	if (gapE <= 0) return INT32_MAX / 2;
	int64_t w = (int64_t)P + ((int64_t)P * mx + gapE - 1) / gapE + 1;
	return w > INT32_MAX / 2 ? INT32_MAX / 2 : (int32_t)w;
}

/* forward colmax over the whole target computed tile by tile, each tile restarted from zero `halo` columns early */
void orc_plain_colmax_tiled(const int8_t* ref, int32_t refLen, const int8_t* read, int32_t readLen, int32_t P,
                            const int8_t* mat, int32_t n, int gapO, int gapE, int32_t tile, int32_t halo,
                            int32_t* colmax)
{
	int32_t* tmp = (int32_t*)malloc(((size_t)tile + halo + 1) * sizeof(int32_t));
	for (int32_t c0 = 0; c0 < refLen; c0 += tile) {
		int32_t c1 = c0 + tile < refLen ? c0 + tile : refLen;
		int32_t h0 = c0 - halo > 0 ? c0 - halo : 0;
		orc_plain_fill(ref, h0, c1, 0, read, readLen, P, mat, n, gapO, gapE, -1, tmp, 0);
		memcpy(colmax + c0, tmp + (c0 - h0), (size_t)(c1 - c0) * sizeof(int32_t));
	}
	free(tmp);
}

/* second-best scan over a colmax array: ssw.c:368-381 (byte) / 570-583 (word) */
void orc_second_best(const int32_t* colmax, int32_t refLen, int32_t end_ref, int32_t maskLen, int is_byte,
                     int32_t* score2, int32_t* ref2)
{
	int32_t s = 0, r = 0, i;
	int32_t edge = (end_ref - maskLen) > 0 ? (end_ref - maskLen) : 0;
	for (i = 0; i < edge; ++i) if (colmax[i] > s) { s = colmax[i]; r = i; }
	edge = (end_ref + maskLen) > refLen ? refLen : (end_ref + maskLen);
	for (i = is_byte ? edge + 1 : edge; i < refLen; ++i) if (colmax[i] > s) { s = colmax[i]; r = i; }
	*score2 = s; *ref2 = r;
}

/* ------------------------------------------------------------------ */
/* banded traceback: banded_sw ssw.c:590-783                            */
/* ------------------------------------------------------------------ */

static inline uint32_t orc_pack(uint32_t len, int op) { return (len << 4) | (uint32_t)op; }  /* M=0 I=1 D=2 */

/* band-local column of matrix cell (i, j): set_u ssw.c:92 */
static inline int32_t band_u(int32_t w, int32_t i, int32_t j) { int32_t x = i - w; if (x < 0) x = 0; return j - x + 1; }
/* direction-line offset: set_d ssw.c:95 */
static inline int32_t band_d(int32_t w, int32_t i, int32_t j, int32_t p) { int32_t x = i - w; if (x < 0) x = 0; return (j - x) * 3 + p; }

/*
 * Returns the CIGAR length (>= 1) and writes BAM-packed ops to cig (cap words),
 * or -1 when the reference returns NULL ("Trace back error").
 */
int32_t orc_banded(const int8_t* ref, const int8_t* read, int32_t refLen, int32_t readLen, int32_t score,
                   uint32_t gapO, uint32_t gapE, int32_t band_width, const int8_t* mat, int32_t n,
                   uint32_t* cig, int32_t cap)
{
	const int32_t NEG = INT32_MIN / 2;
	int32_t len = refLen > readLen ? refLen : readLen;
	int32_t best = 0, best_i = 0, best_j = 0;
	int32_t *hb = 0, *eb = 0, *hc = 0; int8_t* dir = 0;
	int32_t width, width_d, i, j;

	do {
		width = band_width * 2 + 3; width_d = band_width * 2 + 1;
		hb = (int32_t*)realloc(hb, (size_t)(width + 1) * sizeof(int32_t));
		eb = (int32_t*)realloc(eb, (size_t)(width + 1) * sizeof(int32_t));
		hc = (int32_t*)realloc(hc, (size_t)(width + 1) * sizeof(int32_t));
		dir = (int8_t*)realloc(dir, (size_t)width_d * readLen * 3 + 16);
		/* NB: realloc keeps old contents like the reference; stale cells are never read (band grows). */
		for (j = 1; j < width - 1; ++j) hb[j] = 0;
		for (i = 0; i < readLen; ++i) {
			int32_t beg = i - band_width > 0 ? i - band_width : 0;
			int32_t end = i + band_width < refLen - 1 ? i + band_width : refLen - 1;
			int32_t edge = end + 1 < width - 1 ? end + 1 : width - 1;
			int32_t f = NEG, u = 0;
			int8_t* line = dir + (size_t)width_d * i * 3;
			hb[0] = hb[edge] = hc[0] = 0;
			eb[0] = eb[edge] = NEG;
			for (j = beg; j <= end; ++j) {
				u = band_u(band_width, i, j);
				int32_t up = band_u(band_width, i - 1, j);
				int32_t lf = band_u(band_width, i, j - 1);
				int32_t dg = band_u(band_width, i - 1, j - 1);
				int32_t open, ext, e1, f1, gap, dia;
				int8_t de, df;
				open = i == 0 ? -(int32_t)gapO : hb[up] - (int32_t)gapO;
				ext  = i == 0 ? NEG : eb[up] - (int32_t)gapE;
				eb[u] = open > ext ? open : ext;
				de = open > ext ? 3 : 2;
				line[band_d(band_width, i, j, 0)] = de;

				open = hc[lf] - (int32_t)gapO;
				ext  = f - (int32_t)gapE;
				f = open > ext ? open : ext;
				df = open > ext ? 5 : 4;
				line[band_d(band_width, i, j, 1)] = df;

				e1 = eb[u] > 0 ? eb[u] : 0;
				f1 = f > 0 ? f : 0;
				gap = e1 > f1 ? e1 : f1;
				dia = hb[dg] + mat[(int32_t)ref[j] * n + read[i]];
				hc[u] = gap > dia ? gap : dia;
				if (hc[u] > best) { best = hc[u]; best_i = i; best_j = j; }
				line[band_d(band_width, i, j, 2)] = gap <= dia ? 1 : (e1 > f1 ? de : df);
			}
			for (j = 1; j <= u; ++j) hb[j] = hc[j];
		}
		band_width *= 2;
	} while (best < score && band_width <= len);
	band_width /= 2;

	/* trace back from (best_i, best_j): ssw.c:682-762 */
	int32_t nops = 0, run = 0, state = 2, fail = 0;
	int cur = 0, prev = 0;  /* op codes: 0 M, 1 I, 2 D */
	uint32_t* rev = (uint32_t*)malloc(((size_t)refLen + readLen + 4) * sizeof(uint32_t));
	i = best_i; j = best_j;
	while (i >= 0 && j > 0) {
		int8_t d = dir[(size_t)width_d * i * 3 + band_d(band_width, i, j, state)];
		switch (d) {
			case 1: --i; --j; state = 2; cur = 0; break;
			case 2: --i;      state = 0; cur = 1; break;
			case 3: --i;      state = 2; cur = 1; break;
			case 4: --j;      state = 1; cur = 2; break;
			case 5: --j;      state = 2; cur = 2; break;
			default: fail = 1; break;
		}
		if (fail) break;
		if (cur == prev) ++run;
		else { rev[nops++] = orc_pack(run, prev); prev = cur; run = 1; }
	}
	int32_t out = -1;
	if (!fail) {
		if (cur == 0) rev[nops++] = orc_pack(run + 1, 0);
		else { rev[nops++] = orc_pack(run, cur); rev[nops++] = orc_pack(1, 0); }
		for (i = 0; i < nops && i < cap; ++i) cig[i] = rev[nops - 1 - i];
		out = nops;
	}
	free(rev); free(hb); free(eb); free(hc); free(dir);
	return out;
}

/* cigar_alignment_score ssw.c:785-811 */
int32_t orc_cigar_score(const uint32_t* cig, int32_t cigLen, const int8_t* ref, const int8_t* read,
                        const int8_t* mat, int32_t n, uint32_t gapO, uint32_t gapE)
{
	int32_t score = 0, rp = 0, qp = 0;
	for (int32_t i = 0; i < cigLen; ++i) {
		uint32_t len = cig[i] >> 4; uint32_t op = cig[i] & 0xf;
		if (op == 0 || op > 8) {   /* 'M' (codes > 8 print as 'M': ssw.h:180-182) */
			for (uint32_t k = 0; k < len; ++k) score += mat[(int32_t)ref[rp++] * n + read[qp++]];
		} else {
			score -= (int32_t)(gapO + (len > 1 ? (len - 1) * gapE : 0));
			if (op == 1) qp += len; else if (op == 2) rp += len;
		}
	}
	return score;
}

/* ------------------------------------------------------------------ */
/* driver: ssw_init + ssw_align (ssw.c:826-847, 855-977)                */
/* ------------------------------------------------------------------ */

int32_t orc_bias(const int8_t* mat, int32_t n)
{
	int32_t b = 0; for (int32_t i = 0; i < n * n; ++i) if (mat[i] < b) b = mat[i];
	return b < 0 ? -b : b;
}

/*
 * model: 0 = lane model (exact everywhere), 1 = plain model (requires gapO > gapE).
 * cig must hold at least readLen + refLen + 4 words when a CIGAR may be produced.
 */
void orc_align(int model, const int8_t* read, int32_t readLen, const int8_t* mat, int32_t n, int score_size,
               const int8_t* ref, int32_t refLen, int gapO, int gapE, int flag, int filters, int32_t filterd,
               int32_t maskLen, orc_result* r, uint32_t* cig, int32_t cigcap)
{
	orc_end e[2], er[2];
	int word = 0, have_byte = (score_size == 0 || score_size == 2), have_word = (score_size == 1 || score_size == 2);
	int32_t bias = have_byte ? orc_bias(mat, n) : 0;
	memset(r, 0, sizeof(*r));
	r->ref_begin1 = -1; r->read_begin1 = -1;

	if (model == 0) {
		if (have_byte) {
			orc_striped(1, ref, 0, refLen, read, readLen, mat, n, gapO, gapE, 255, bias, maskLen, e, 0);
			if (e[0].score == 255) {
				if (!have_word) { r->is_null = 1; return; }
				orc_striped(0, ref, 0, refLen, read, readLen, mat, n, gapO, gapE, 65535, 0, maskLen, e, 0);
				word = 1;
			}
		} else if (have_word) {
			orc_striped(0, ref, 0, refLen, read, readLen, mat, n, gapO, gapE, 65535, 0, maskLen, e, 0);
			word = 1;
		} else { r->is_null = 1; return; }
	} else {
		/* plain model: one 16-bit pass over the P16 rows decides everything */
		int32_t P16 = (readLen + 15) / 16 * 16, P8 = (readLen + 7) / 8 * 8;
		int32_t* cm = (int32_t*)malloc((size_t)(refLen > 0 ? refLen : 1) * sizeof(int32_t));
		orc_end b;
		if (!have_byte && !have_word) { free(cm); r->is_null = 1; return; }
		orc_plain_fill(ref, 0, refLen, 0, read, readLen, P16, mat, n, gapO, gapE, -1, cm, &b);
		if (have_byte && b.score < 255 - bias) word = 0;
		else if (have_word) word = 1;
		else { free(cm); r->is_null = 1; return; }
		if (word && P8 != P16) orc_plain_fill(ref, 0, refLen, 0, read, readLen, P8, mat, n, gapO, gapE, -1, cm, &b);
		e[0].score = b.score; e[0].ref = b.score > 0 ? b.ref : (word ? 0 : -1);
		e[0].read = b.score > 0 ? (b.read < readLen - 1 ? b.read : readLen - 1) : 0;
		orc_second_best(cm, refLen, e[0].ref, maskLen, !word, &e[1].score, &e[1].ref);
		free(cm);
	}
	r->used_word = word;
	if (e[0].score <= 0) return;
	r->score1 = e[0].score; r->ref_end1 = e[0].ref; r->read_end1 = e[0].read;
	if (maskLen >= 15) { r->score2 = e[1].score; r->ref_end2 = e[1].ref; }
	else { r->score2 = 0; r->ref_end2 = -1; }
	if (flag == 0 || (flag == 2 && r->score1 < filters)) return;

	/* begin position: reverse pass on the reversed read prefix, ssw.c:919-930 */
	int32_t plen = r->read_end1 + 1;
	int8_t* rr = (int8_t*)malloc(plen);
	for (int32_t k = 0; k < plen; ++k) rr[k] = read[r->read_end1 - k];
	if (model == 0) {
		if (!word) orc_striped(1, ref, 1, r->ref_end1 + 1, rr, plen, mat, n, gapO, gapE, (uint8_t)r->score1, bias, maskLen, er, 0);
		else       orc_striped(0, ref, 1, r->ref_end1 + 1, rr, plen, mat, n, gapO, gapE, (uint16_t)r->score1, 0, maskLen, er, 0);
	} else {
		int32_t P = word ? (plen + 7) / 8 * 8 : (plen + 15) / 16 * 16;
		orc_end b;
		orc_plain_fill(ref, 0, r->ref_end1 + 1, 1, rr, plen, P, mat, n, gapO, gapE, r->score1, 0, &b);
		er[0].score = b.score;
		er[0].ref = b.score > 0 ? r->ref_end1 - b.ref : (word ? 0 : -1);
		er[0].read = b.score > 0 ? (b.read < plen - 1 ? b.read : plen - 1) : plen - 1;
	}
	free(rr);
	r->ref_begin1 = er[0].ref;
	r->read_begin1 = r->read_end1 - er[0].read;
	if (r->score1 > er[0].score) r->flag = 2;

	if ((7 & flag) == 0 || ((2 & flag) != 0 && r->score1 < filters) ||
	    ((4 & flag) != 0 && (r->ref_end1 - r->ref_begin1 > filterd || r->read_end1 - r->read_begin1 > filterd))) return;

	int32_t sub_ref = r->ref_end1 - r->ref_begin1 + 1, sub_read = r->read_end1 - r->read_begin1 + 1;
	int32_t band = abs(sub_ref - sub_read) + 1, full = sub_ref > sub_read ? sub_ref : sub_read, clen;
	for (;;) {
		clen = orc_banded(ref + r->ref_begin1, read + r->read_begin1, sub_ref, sub_read, r->score1,
		                  (uint32_t)gapO, (uint32_t)gapE, band, mat, n, cig, cigcap);
		if (clen < 0) break;
		if (orc_cigar_score(cig, clen, ref + r->ref_begin1, read + r->read_begin1, mat, n, gapO, gapE) == r->score1) break;
		if (band >= full) { clen = -1; break; }
		band = full;
	}
	if (clen < 0) r->flag = 1; else r->cigarLen = clen;
}

/* mark_mismatch ssw.c:1019-1074: returns NM; out receives the rewritten CIGAR (cap >= cigLen + 2*readLen + 2) */
int32_t orc_mark_mismatch(int32_t ref_begin1, int32_t read_begin1, int32_t read_end1, const int8_t* ref,
                          const int8_t* read, int32_t readLen, const uint32_t* cig, int32_t cigLen,
                          uint32_t* out, int32_t* outLen)
{
	int32_t nm = 0, p = 0; uint32_t eq = 0, ne = 0;
	const int8_t* rp = ref + ref_begin1; const int8_t* qp = read + read_begin1;
	if (read_begin1 > 0) out[p++] = orc_pack(read_begin1, 4);
	for (int32_t i = 0; i < cigLen; ++i) {
		uint32_t len = cig[i] >> 4, op = cig[i] & 0xf;
		if (op == 0 || op > 8) {
			for (uint32_t k = 0; k < len; ++k, ++rp, ++qp) {
				if (*rp != *qp) { ++nm; if (eq) { out[p++] = orc_pack(eq, 7); eq = 0; } ++ne; }
				else { if (ne) { out[p++] = orc_pack(ne, 8); ne = 0; } ++eq; }
			}
		} else if (op == 1 || op == 2) {
			if (op == 1) qp += len; else rp += len;
			nm += len;
			if (eq) { out[p++] = orc_pack(eq, 7); eq = 0; } else if (ne) { out[p++] = orc_pack(ne, 8); ne = 0; }
			out[p++] = orc_pack(len, op);
		}
	}
	if (eq) out[p++] = orc_pack(eq, 7); else if (ne) out[p++] = orc_pack(ne, 8);
	if (readLen - read_end1 - 1 > 0) out[p++] = orc_pack(readLen - read_end1 - 1, 4);
	*outLen = p;
	return nm;
}
