/*
 * ssw_cli.c -- "ssw_test_gpu": batched command-line front end with the options and the byte-identical stdout of the
 * reference's ssw_test (reference src/main.c:395-547; output formats of ssw_write, src/main.c:118-245), SURVEY 8f-1.
 *
 * What differs from the reference CLI is only the loop structure: the target file is parsed ONCE and stays resident
 * in HBM, reads are taken in batches, and each batch is ONE ssw_gpu_align_batch() call (with -r: both orientations as one
 * batch of 2 N queries) instead of reads x targets synchronous ssw_align() calls.  Three stages run on three threads with two
 * batches in flight between them: parse | upload + translate on the device + align | format + write, so the host work hides
 * behind the device.  Output order is the reference's: reads outer, targets inner.
 *
 *   ssw_test_gpu [-m N] [-x N] [-o N] [-e N] [-p] [-a FILE] [-c] [-f N] [-r] [-s] [-h] <target.fa> <query.fa|fq>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <zlib.h>
#include "ssw.h"
#include "ssw_gpu.h"

/* BLOSUM50, 24 letters ARNDCQEGHILKMFPSTWYVBZX* (the reference's default protein matrix, src/main.c:43-69) */
static const int8_t blosum50[24 * 24] = {
	 5,-2,-1,-2,-1,-1,-1, 0,-2,-1,-2,-1,-1,-3,-1, 1, 0,-3,-2, 0,-2,-1,-1,-5,
	-2, 7,-1,-2,-4, 1, 0,-3, 0,-4,-3, 3,-2,-3,-3,-1,-1,-3,-1,-3,-1, 0,-1,-5,
	-1,-1, 7, 2,-2, 0, 0, 0, 1,-3,-4, 0,-2,-4,-2, 1, 0,-4,-2,-3, 5, 0,-1,-5,
	-2,-2, 2, 8,-4, 0, 2,-1,-1,-4,-4,-1,-4,-5,-1, 0,-1,-5,-3,-4, 6, 1,-1,-5,
	-1,-4,-2,-4,13,-3,-3,-3,-3,-2,-2,-3,-2,-2,-4,-1,-1,-5,-3,-1,-3,-3,-1,-5,
	-1, 1, 0, 0,-3, 7, 2,-2, 1,-3,-2, 2, 0,-4,-1, 0,-1,-1,-1,-3, 0, 4,-1,-5,
	-1, 0, 0, 2,-3, 2, 6,-3, 0,-4,-3, 1,-2,-3,-1,-1,-1,-3,-2,-3, 1, 5,-1,-5,
	 0,-3, 0,-1,-3,-2,-3, 8,-2,-4,-4,-2,-3,-4,-2, 0,-2,-3,-3,-4,-1,-2,-1,-5,
	-2, 0, 1,-1,-3, 1, 0,-2,10,-4,-3, 0,-1,-1,-2,-1,-2,-3, 2,-4, 0, 0,-1,-5,
	-1,-4,-3,-4,-2,-3,-4,-4,-4, 5, 2,-3, 2, 0,-3,-3,-1,-3,-1, 4,-4,-3,-1,-5,
	-2,-3,-4,-4,-2,-2,-3,-4,-3, 2, 5,-3, 3, 1,-4,-3,-1,-2,-1, 1,-4,-3,-1,-5,
	-1, 3, 0,-1,-3, 2, 1,-2, 0,-3,-3, 6,-2,-4,-1, 0,-1,-3,-2,-3, 0, 1,-1,-5,
	-1,-2,-2,-4,-2, 0,-2,-3,-1, 2, 3,-2, 7, 0,-3,-2,-1,-1, 0, 1,-3,-1,-1,-5,
	-3,-3,-4,-5,-2,-4,-3,-4,-1, 0, 1,-4, 0, 8,-4,-3,-2, 1, 4,-1,-4,-4,-1,-5,
	-1,-3,-2,-1,-4,-1,-1,-2,-2,-3,-4,-1,-3,-4,10,-1,-1,-4,-3,-3,-2,-1,-1,-5,
	 1,-1, 1, 0,-1, 0,-1, 0,-1,-3,-3, 0,-2,-3,-1, 5, 2,-4,-2,-2, 0, 0,-1,-5,
	 0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 2, 5,-3,-2, 0, 0,-1,-1,-5,
	-3,-3,-4,-5,-5,-1,-3,-3,-3,-3,-2,-3,-1, 1,-4,-4,-3,15, 2,-3,-5,-2,-1,-5,
	-2,-1,-2,-3,-3,-1,-2,-3, 2,-1,-1,-2, 0, 4,-3,-2,-2, 2, 8,-1,-3,-2,-1,-5,
	 0,-3,-3,-4,-1,-3,-3,-4,-4, 4, 1,-3, 1,-1,-3,-2, 0,-3,-1, 5,-3,-3,-1,-5,
	-2,-1, 5, 6,-3, 0, 1,-1, 0,-4,-4, 0,-3,-4,-2, 0, 0,-5,-3,-3, 6, 1,-1,-5,
	-1, 0, 0, 1,-3, 4, 5,-2, 0,-3,-3, 1,-1,-4,-1, 0,-1,-2,-2,-3, 1, 5,-1,-5,
	-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-5,
	-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5,-5, 1
};

static int8_t aa_code[128], nt_code[128];

/* SSW_CLI_TRACE=1: wall-clock of the run's milestones on stderr (where an end-to-end run spends its time beside the device) */
static int cli_trace_on;
static double cli_t0;
static double cli_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
#define CLI_TRACE(...) do { if (cli_trace_on) { fprintf(stderr, "[ssw_test_gpu %8.3f s] ", cli_now() - cli_t0); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)

static void init_tables(void)
{
	static const char* aa = "ARNDCQEGHILKMFPSTWYVBZX";
	memset(aa_code, 23, sizeof aa_code);       /* everything unknown is '*' (reference src/main.c:72-81) */
	for (int i = 0; aa[i]; ++i) { aa_code[(int)aa[i]] = (int8_t)i; aa_code[(int)aa[i] + 32] = (int8_t)i; }
	memset(nt_code, 4, sizeof nt_code);        /* reference src/main.c:84-93 */
	nt_code['A'] = nt_code['a'] = 0; nt_code['C'] = nt_code['c'] = 1; nt_code['G'] = nt_code['g'] = 2;
	nt_code['T'] = nt_code['t'] = 3; nt_code['U'] = nt_code['u'] = 3;
}

/* ---------------------------------------------------------------- FASTA / FASTQ reader (plain or gz) */
/* a command-line tool: running out of memory ends the run with a message */
static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "ssw_test_gpu: out of memory\n"); exit(EXIT_FAILURE); } return p; }
static void* xrealloc(void* q, size_t n) { void* p = realloc(q, n ? n : 1); if (!p) { fprintf(stderr, "ssw_test_gpu: out of memory\n"); exit(EXIT_FAILURE); } return p; }

typedef struct { char* name; char* seq; char* qual; int32_t len; } record;
typedef struct { gzFile f; unsigned char buf[1 << 16]; int n, pos, eof, last_header; } reader;

static int rd_getc(reader* r)
{
	if (r->pos >= r->n) {
		if (r->eof) return -1;
		r->n = gzread(r->f, r->buf, sizeof r->buf);
		r->pos = 0;
		if (r->n <= 0) { r->eof = 1; r->n = 0; return -1; }
	}
	return r->buf[r->pos++];
}

typedef struct { char* s; size_t l, cap; } str;
static void str_push(str* s, int ch)
{
	if (s->l + 2 > s->cap) { s->cap = s->cap ? s->cap * 2 : 256; s->s = (char*)xrealloc(s->s, s->cap); }
	s->s[s->l++] = (char)ch; s->s[s->l] = 0;
}

/* returns 1 and fills rec, 0 at end of input; same record model as the reference's kseq.h (header token up to the
   first blank, sequence lines concatenated until a line starts with '>', '@' or '+', FASTQ quality as long as the sequence) */
static int read_record(reader* r, record* rec)
{
	int c;
	if (!r->last_header) {
		while ((c = rd_getc(r)) != -1 && c != '>' && c != '@') {}
		if (c == -1) return 0;
		r->last_header = c;
	}
	str name = { 0, 0, 0 }, seq = { 0, 0, 0 }, qual = { 0, 0, 0 };
	str_push(&name, 0); name.l = 0;
	while ((c = rd_getc(r)) != -1 && c != ' ' && c != '\t' && c != '\n' && c != '\r') str_push(&name, c);
	while (c != -1 && c != '\n') c = rd_getc(r);
	str_push(&seq, 0); seq.l = 0;
	while ((c = rd_getc(r)) != -1 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue;
		do { if (c != '\r') str_push(&seq, c); c = rd_getc(r); } while (c != -1 && c != '\n');
	}
	r->last_header = (c == '>' || c == '@') ? c : 0;
	if (c == '+') {
		while ((c = rd_getc(r)) != -1 && c != '\n') {}
		str_push(&qual, 0); qual.l = 0;
		while (qual.l < seq.l && (c = rd_getc(r)) != -1) if (c != '\n' && c != '\r') str_push(&qual, c);
		r->last_header = 0;
	}
	rec->name = name.s; rec->seq = seq.s; rec->len = (int32_t)seq.l;
	rec->qual = qual.s;
	return 1;
}

static void free_record(record* r) { free(r->name); free(r->seq); free(r->qual); }

/* ---------------------------------------------------------------- output (reference src/main.c:118-245) */
static void write_alignment(const s_align* a0, const record* ref, const record* read, const char* read_seq, const int8_t* ref_num,
                            const int8_t* read_num, const int8_t* table, int strand, int sam, int32_t nm_dev)
{
	s_align* a = (s_align*)a0;
	if (!sam) {
		fprintf(stdout, "target_name: %s\nquery_name: %s\noptimal_alignment_score: %d\t", ref->name, read->name, a->score1);
		if (a->score2 > 0) fprintf(stdout, "suboptimal_alignment_score: %d\t", a->score2);
		fprintf(stdout, strand == 0 ? "strand: +\t" : "strand: -\t");
		if (a->ref_begin1 + 1) fprintf(stdout, "target_begin: %d\t", a->ref_begin1 + 1);
		fprintf(stdout, "target_end: %d\t", a->ref_end1 + 1);
		if (a->read_begin1 + 1) fprintf(stdout, "query_begin: %d\t", a->read_begin1 + 1);
		fprintf(stdout, "query_end: %d\n\n", a->read_end1 + 1);
		if (!a->cigar) return;
		/* 60-column blocks: target line, match line, query line; an op may be split across blocks */
		int32_t op_i = 0, left = 0, tpos = a->ref_begin1, qpos = a->read_begin1;
		while (op_i < a->cigarLen || left > 0) {
			int32_t count, c, t = tpos, q = qpos, next_op = op_i, next_left = 0;
			uint32_t i;
			fprintf(stdout, "Target: %8d    ", t + 1);
			for (count = 0, c = op_i; c < a->cigarLen && count < 60; ++c) {
				const char letter = cigar_int_to_op(a->cigar[c]);
				const uint32_t l = (c == op_i && left > 0) ? (uint32_t)left : cigar_int_to_len(a->cigar[c]);
				for (i = 0; i < l && count < 60; ++i, ++count) {
					if (letter == 'I') fputc('-', stdout);
					else fputc(ref->seq[t++], stdout);
				}
			}
			fprintf(stdout, "    %d\n                    ", t);
			t = tpos;
			for (count = 0, c = op_i; c < a->cigarLen && count < 60; ++c) {
				const char letter = cigar_int_to_op(a->cigar[c]);
				const uint32_t l = (c == op_i && left > 0) ? (uint32_t)left : cigar_int_to_len(a->cigar[c]);
				for (i = 0; i < l && count < 60; ++i, ++count) {
					if (letter == 'M') {
						fputc(table[(int)ref->seq[t]] == table[(int)read_seq[q]] ? '|' : '*', stdout);
						++t; ++q;
					} else { fputc(' ', stdout); if (letter == 'I') ++q; else ++t; }
				}
			}
			const int32_t t_after = t;
			q = qpos;
			fprintf(stdout, "\nQuery:  %8d    ", q + 1);
			for (count = 0, c = op_i; c < a->cigarLen && count < 60; ++c) {
				const char letter = cigar_int_to_op(a->cigar[c]);
				const uint32_t l = (c == op_i && left > 0) ? (uint32_t)left : cigar_int_to_len(a->cigar[c]);
				for (i = 0; i < l && count < 60; ++i, ++count) {
					if (letter == 'D') fputc('-', stdout);
					else fputc(read_seq[q++], stdout);
				}
				if (count == 60 && i < l) { next_op = c; next_left = (int32_t)(l - i); }
				else { next_op = c + 1; next_left = 0; }
			}
			fprintf(stdout, "    %d\n\n", q);
			tpos = t_after; qpos = q; op_i = next_op; left = next_left;
		}
		return;
	}
	fprintf(stdout, "%s\t", read->name);
	if (a->score1 == 0) { fprintf(stdout, "4\t*\t0\t255\t*\t*\t0\t0\t*\t*\n"); return; }
	int32_t c, p;
	uint32_t mapq = -4.343 * log(1 - (double)abs(a->score1 - a->score2) / (double)a->score1);
	mapq = (uint32_t)(mapq + 4.99);
	mapq = mapq < 254 ? mapq : 254;
	fprintf(stdout, strand ? "16\t" : "0\t");
	fprintf(stdout, "%s\t%d\t%d\t", ref->name, a->ref_begin1 + 1, mapq);
	/* nm_dev >= 0: the CIGAR was already rewritten ('=', 'X', soft clips) and counted on the device (ssw_gpu_params.mark_mismatch) */
	const int32_t nm = nm_dev >= 0 ? nm_dev : mark_mismatch(a->ref_begin1, a->read_begin1, a->read_end1, ref_num, read_num, read->len, &a->cigar, &a->cigarLen);
	for (c = 0; c < a->cigarLen; ++c) fprintf(stdout, "%lu%c", (unsigned long)cigar_int_to_len(a->cigar[c]), cigar_int_to_op(a->cigar[c]));
	fprintf(stdout, "\t*\t0\t0\t%s\t", read_seq);
	if (read->qual && strand) { for (p = read->len - 1; p >= 0; --p) fputc(read->qual[p], stdout); }
	else if (read->qual) fprintf(stdout, "%s", read->qual);
	else fputc('*', stdout);
	fprintf(stdout, "\tAS:i:%d\tNM:i:%d\t", a->score1, nm);
	if (a->score2 > 0) fprintf(stdout, "ZS:i:%d\n", a->score2);
	else fputc('\n', stdout);
}

static void reverse_complement(const char* s, int32_t len, char* out)
{
	for (int32_t i = 0; i < len; ++i) {
		const char ch = s[len - 1 - i];
		char o = 4;   /* the reference maps every non-ACGT character to byte value 4 (src/main.c:97-106) */
		if (ch == 'A' || ch == 'a') o = 'T'; else if (ch == 'C' || ch == 'c') o = 'G';
		else if (ch == 'G' || ch == 'g') o = 'C'; else if (ch == 'T' || ch == 't' || ch == 'U' || ch == 'u') o = 'A';
		else if (ch == 'N' || ch == 'n') o = 'N';
		out[i] = o;
	}
	out[len] = 0;
}

static int load_matrix(const char* path, int8_t** mat, int32_t* n)
{
	FILE* f = fopen(path, "r");
	if (!f) { fprintf(stderr, "Failed to open the weight matrix file.\n"); return 1; }
	int8_t* m = (int8_t*)xmalloc(1024);
	char line[128];
	int32_t k = 0, rows = 0;
	while (fgets(line, sizeof line, f)) {
		if (!(line[0] == '*' || (line[0] >= 'A' && line[0] <= 'Z'))) continue;
		if (line[0] >= 'A' && line[0] <= 'Z') aa_code[(int)line[0]] = aa_code[(int)line[0] + 32] = (int8_t)rows;
		char tok[8]; int tl = 0;
		for (int l = 1; ; ++l) {
			const char ch = line[l];
			if ((ch >= '0' && ch <= '9') || ch == '-') { if (tl < 7) tok[tl++] = ch; }
			else { if (tl > 0 && k < 1024) { tok[tl] = 0; m[k++] = (int8_t)atoi(tok); tl = 0; } }
			if (!ch) break;
		}
		++rows;
	}
	fclose(f);
	if (k == 0) { fprintf(stderr, "Problem of reading the weight matrix file.\n"); free(m); return 1; }
	*mat = m; *n = rows;
	return 0;
}

/* ---------------------------------------------------------------- the three-stage pipeline */
#include <pthread.h>
/* one batch of reads on its way through the stages */
typedef struct {
	int32_t nr; int64_t total;
	record* reads;               /* parsed records (names, ASCII sequences, qualities) */
	int64_t* qoff;               /* nr + 1 offsets into qtext */
	char* qtext;                 /* the reads' residues back to back: what is uploaded; translation to codes happens on the device */
	ssw_gpu_result* res;         /* nr x nt records ('+' strand), with -r followed by nr x nt of the reverse complements */
	uint32_t* pool; int64_t words;
	int fatal;                   /* non-zero: the run ends with this exit code (message already printed) */
} work;

#define QUEUE_DEPTH 2            /* batches in flight between two stages */
typedef struct { void* item[QUEUE_DEPTH]; int head, count, closed; pthread_mutex_t mu; pthread_cond_t cv; } queue;
static void queue_init(queue* q) { memset(q, 0, sizeof *q); pthread_mutex_init(&q->mu, 0); pthread_cond_init(&q->cv, 0); }
static void queue_push(queue* q, void* it)      /* it == NULL: no more items */
{
	pthread_mutex_lock(&q->mu);
	if (!it) q->closed = 1;
	else {
		while (q->count == QUEUE_DEPTH) pthread_cond_wait(&q->cv, &q->mu);
		q->item[(q->head + q->count++) % QUEUE_DEPTH] = it;
	}
	pthread_cond_broadcast(&q->cv);
	pthread_mutex_unlock(&q->mu);
}
static void* queue_pop(queue* q)                /* NULL: closed and drained */
{
	pthread_mutex_lock(&q->mu);
	while (q->count == 0 && !q->closed) pthread_cond_wait(&q->cv, &q->mu);
	void* it = 0;
	if (q->count > 0) { it = q->item[q->head]; q->head = (q->head + 1) % QUEUE_DEPTH; --q->count; }
	pthread_cond_broadcast(&q->cv);
	pthread_mutex_unlock(&q->mu);
	return it;
}

typedef struct {
	gzFile qf; int32_t batch; int reverse, reverse_fatal; const int8_t* table; int nt_table;
	ssw_gpu_ctx* g; ssw_gpu_pool* gp; ssw_gpu_seqs* T; int32_t nt;
	ssw_gpu_params prm;
	queue parsed, aligned;
} pipeline;

static void free_work(work* w, int reverse)
{
	(void)reverse;
	for (int32_t q = 0; q < w->nr; ++q) free_record(&w->reads[q]);
	free(w->reads); free(w->qoff); free(w->qtext); free(w->res); free(w->pool); free(w);
}

/* stage 1: the query file -> batches of records + one ASCII block per batch.  The first batches are small so that the device starts
   early (the parse of a full batch would otherwise stand in front of the whole run), then `batch` reads each. */
static void* stage_parse(void* arg)
{
	pipeline* pl = (pipeline*)arg;
	reader* qr = (reader*)xmalloc(sizeof(reader)); memset(qr, 0, sizeof *qr);
	qr->f = pl->qf;      /* opened by main() before anything else, like the reference (src/main.c:434-439) */
	gzbuffer(qr->f, 1 << 20);
	int32_t ramp = pl->batch < 4096 ? pl->batch : 4096;
	for (int first = 1; ; first = 0) {
		work* w = (work*)xmalloc(sizeof(work)); memset(w, 0, sizeof *w);
		int32_t cap = ramp;
		w->reads = (record*)xmalloc(sizeof(record) * (size_t)cap);
		while (w->nr < cap && read_record(qr, &w->reads[w->nr])) { w->total += w->reads[w->nr].len; ++w->nr; }
		if (w->nr == 0) { free(w->reads); free(w); break; }
		if (first && pl->reverse_fatal) {      /* reference src/main.c:483-486: the run stops at the first read */
			fprintf(stderr, "Reverse complement alignment is not available for protein sequences. \n");
			w->fatal = 1; queue_push(&pl->parsed, w); break;
		}
		w->qoff = (int64_t*)xmalloc(sizeof(int64_t) * ((size_t)w->nr + 1));
		w->qtext = (char*)xmalloc((size_t)w->total + 1);
		w->qoff[0] = 0;
		for (int32_t q = 0; q < w->nr; ++q) {
			memcpy(w->qtext + w->qoff[q], w->reads[q].seq, (size_t)w->reads[q].len);
			w->qoff[q + 1] = w->qoff[q] + w->reads[q].len;
		}
		CLI_TRACE("parsed a batch of %d reads", w->nr);
		const int short_batch = w->nr < cap;      /* (w belongs to the next stage once it is pushed) */
		queue_push(&pl->parsed, w);
		if (short_batch) break;
		if (ramp < pl->batch) ramp = (int64_t)ramp * 4 < pl->batch ? ramp * 4 : pl->batch;
	}
	queue_push(&pl->parsed, 0);
	gzclose(qr->f); free(qr);
	return 0;
}

/* stage 2: upload (ASCII: the translation to residue codes runs on the device), with -r the reverse complements on the device too and
   both orientations as ONE batch of 2 N queries, then the batch call.  One thread is enough here: an upload is a fraction of a
   millisecond per megabyte against seconds of alignment -- what has to overlap with the device are the two HOST stages. */
static void* stage_device(void* arg)
{
	pipeline* pl = (pipeline*)arg;
	const int32_t nt = pl->nt;
	for (;;) {
		work* w = (work*)queue_pop(&pl->parsed);
		if (!w) break;
		if (w->fatal) { queue_push(&pl->aligned, w); break; }
		const int32_t nr = w->nr, sets = pl->reverse ? 2 : 1;
		w->res = (ssw_gpu_result*)xmalloc(sizeof(ssw_gpu_result) * (size_t)nr * (size_t)(nt ? nt : 1) * (size_t)sets);
		if (pl->gp) {      /* -g N: reads stay on the host (codes: the workers upload the blocks they take); both orientations as one batch of 2 N */
			int8_t* codes = (int8_t*)xmalloc((size_t)w->total * (size_t)sets + 1);
			int64_t* off2 = (int64_t*)xmalloc(sizeof(int64_t) * ((size_t)nr * (size_t)sets + 1));
			for (int64_t i = 0; i < w->total; ++i) codes[i] = pl->table[(int)w->qtext[i] & 127];
			for (int32_t q = 0; q <= nr; ++q) off2[q] = w->qoff[q];
			if (pl->reverse) {
				char* tmp = 0; size_t cap = 0;
				for (int32_t q = 0; q < nr; ++q) {
					const int32_t len = w->reads[q].len;
					if (cap < (size_t)len + 1) { cap = (size_t)len * 2 + 64; tmp = (char*)xrealloc(tmp, cap); }
					reverse_complement(w->reads[q].seq, len, tmp);
					for (int32_t i = 0; i < len; ++i) codes[w->total + w->qoff[q] + i] = pl->table[(int)tmp[i] & 127];
					off2[nr + q + 1] = w->total + w->qoff[q + 1];
				}
				free(tmp);
			}
			if (ssw_gpu_pool_align(pl->gp, codes, off2, nr * sets, 0, 0, nt, &pl->prm, w->res, &w->pool, &w->words)) {
				fprintf(stderr, "ssw_test_gpu: %s\n", ssw_gpu_pool_last_error(pl->gp));
				free(codes); free(off2);
				w->fatal = EXIT_FAILURE; queue_push(&pl->aligned, w); return 0;
			}
			free(codes); free(off2);
		} else {
			ssw_gpu_seqs* Qs = ssw_gpu_seqs_upload_ascii(pl->g, w->qtext, w->qoff, nr, pl->table);
			ssw_gpu_seqs* Qa = Qs;
			if (Qs && pl->reverse) {
				if (pl->nt_table) Qa = ssw_gpu_seqs_with_revcomp(pl->g, Qs);
				else {      /* a matrix file's own letter table: the complement is taken on the text (reference src/main.c:95-116), both orientations uploaded as text */
					char* text2 = (char*)xmalloc((size_t)w->total * 2 + 1);
					int64_t* off2 = (int64_t*)xmalloc(sizeof(int64_t) * ((size_t)nr * 2 + 1));
					memcpy(text2, w->qtext, (size_t)w->total);
					for (int32_t q = 0; q <= nr; ++q) off2[q] = w->qoff[q];
					char* tmp = 0; size_t cap = 0;
					for (int32_t q = 0; q < nr; ++q) {
						const int32_t len = w->reads[q].len;
						if (cap < (size_t)len + 1) { cap = (size_t)len * 2 + 64; tmp = (char*)xrealloc(tmp, cap); }
						reverse_complement(w->reads[q].seq, len, tmp);
						memcpy(text2 + w->total + w->qoff[q], tmp, (size_t)len);
						off2[nr + q + 1] = w->total + w->qoff[q + 1];
					}
					free(tmp);
					Qa = ssw_gpu_seqs_upload_ascii(pl->g, text2, off2, 2 * nr, pl->table);
					free(text2); free(off2);
				}
			}
			if (!Qs || !Qa || ssw_gpu_align_batch(pl->g, Qa, pl->T, 0, nt, &pl->prm, w->res, &w->pool, &w->words)) {
				fprintf(stderr, "ssw_test_gpu: %s\n", ssw_gpu_last_error(pl->g));
				if (Qa && Qa != Qs) ssw_gpu_seqs_free(Qa);
				if (Qs) ssw_gpu_seqs_free(Qs);
				w->fatal = EXIT_FAILURE; queue_push(&pl->aligned, w); return 0;
			}
			if (Qa != Qs) ssw_gpu_seqs_free(Qa);
			ssw_gpu_seqs_free(Qs);
		}
		CLI_TRACE("aligned a batch of %d reads", nr);
		queue_push(&pl->aligned, w);
	}
	queue_push(&pl->aligned, 0);
	return 0;
}

static void usage(void)
{
	fprintf(stderr, "\nUsage: ssw_test_gpu [options] ... <target.fasta> <query.fasta>(or <query.fastq>)\n"
	                "Options:\n"
	                "\t-m N\tN is a positive integer for weight match in genome sequence alignment. [default: 2]\n"
	                "\t-x N\tN is a positive integer. -N will be used as weight mismatch in genome sequence alignment. [default: 2]\n"
	                "\t-o N\tN is a positive integer. -N will be used as the weight for the gap opening. [default: 3]\n"
	                "\t-e N\tN is a positive integer. -N will be used as the weight for the gap extension. [default: 1]\n"
	                "\t-p\tDo protein sequence alignment. Without this option, the ssw_test will do genome sequence alignment.\n"
	                "\t-a FILE\tFILE is either the Blosum or Pam weight matrix. [default: Blosum50]\n"
	                "\t-c\tReturn the alignment path.\n"
	                "\t-f N\tN is a positive integer. Only output the alignments with the Smith-Waterman score >= N.\n"
	                "\t-r\tThe best alignment will be picked between the original read alignment and the reverse complement read alignment.\n"
	                "\t-s\tOutput in SAM format. [default: no header]\n"
	                "\t-h\tIf -s is used, include header in SAM output.\n"
	                "\t-b N\tReads per GPU batch. [default: 65536]\n"
	                "\t-g N\tSpread every batch over N workers (one per device, round-robin over the visible devices). [default: one device]\n\n");
}


/* ---- the command line ------------------------------------------------------------------------------------------------------
 * Default: the reference's own scanner, reproduced literally (src/main.c:247-330), because "the same command line gives the same
 * bytes" is what a drop-in means -- and the reference's scanner is NOT getopt:
 *   - every character of an argument that starts with '-' is an option letter ("-cs" = -c -s; unknown letters are ignored);
 *   - a value option (m x o e a f) takes the NEXT argument unless that starts with '-' ("-m5" and "-x -3" give no value);
 *   - after taking a value it goes on scanning at the SAME character index in the value's string.  A value shorter than that
 *     index ("-e 1": index 2 of "1") puts the scan behind the string's end -- undefined in C, but on Linux the argument
 *     strings lie back to back (then the environment's), so the reference binary reads the following argument as option letters:
 *     "ssw_test -o 5 -e 2 ref.fa reads.fq" takes the 'r' of "ref.fa" as -r and its 'e' as "-e ref.fa" (gap extension 0).
 *     Reproduced here on the same memory, bounds-checked: only strings that are verified to follow each other are read, and a
 *     note on stderr says what the overrun changed.  (Value options first and a flag such as -c last, or two-character values
 *     like "-e 02", never trigger it.)
 *   - the two files are the first argument that is not an option (or the value of one) and the one after it.
 * -b / -g (this program's own options) are honoured only inside a genuine option argument: the reference ignores those letters.
 * SSW_CLI_ARGS=getopt selects a conventional parser instead (options anywhere, "-m5", negative values). */
typedef struct {
	int32_t *match, *mismatch, *gap_open, *gap_ext, *filter, *batch, *gpus, *protein, *path, *reverse, *sam, *header;
	const char** mat_name;
} cli_opts;

static int scan_args_getopt(int argc, char* const argv[], cli_opts* o, const char* files[2])
{
	int nfiles = 0;
	for (int i = 1; i < argc; ++i) {
		if (argv[i][0] != '-' || !argv[i][1]) { if (nfiles < 2) files[nfiles++] = argv[i]; continue; }
		for (int j = 1; argv[i][j]; ++j) {
			const char c = argv[i][j];
			if (c == 'p') *o->protein = 1; else if (c == 'c') *o->path = 1; else if (c == 'r') *o->reverse = 1;
			else if (c == 's') *o->sam = 1; else if (c == 'h') *o->header = 1;
			else if (strchr("mxoefabg", c)) {
				const char* val = argv[i][j + 1] ? &argv[i][j + 1] : (i + 1 < argc ? argv[++i] : 0);
				if (!val) return 1;
				if (c == 'm') *o->match = atoi(val); else if (c == 'x') *o->mismatch = atoi(val); else if (c == 'o') *o->gap_open = atoi(val);
				else if (c == 'e') *o->gap_ext = atoi(val); else if (c == 'f') *o->filter = atoi(val); else if (c == 'b') *o->batch = atoi(val);
				else if (c == 'g') *o->gpus = atoi(val); else *o->mat_name = val;
				break;
			}
		}
	}
	return nfiles < 2;
}

extern char** environ;

static int scan_args_reference(int argc, char* const argv[], cli_opts* o, const char* files[2])
{
	/* the bytes the reference's scan can reach: argv[1..] and, behind them, the environment -- as far as they really lie back to back */
	const char* lim = argc > 1 ? argv[1] + strlen(argv[1]) + 1 : 0;
	int contiguous = argc > 1;
	for (int k = 2; k < argc && contiguous; ++k) { if (argv[k] == lim) lim += strlen(argv[k]) + 1; else contiguous = 0; }
	for (char** e = environ; contiguous && e && *e; ++e) { if (*e == lim) lim += strlen(*e) + 1; else break; }
	char note[512]; size_t nn = 0; note[0] = 0;
	for (int i = 1; i < argc; i++) {
		if (argv[i][0] != '-') continue;
		const char* own = argv[i];
		const char* own_end = own + strlen(own);               /* behind this: not this option argument any more */
		const char* cur_end = own_end;                         /* end of the string the index is applied to (it changes when a value is taken) */
		for (int j = 1; ; j++) {
			const char* p = argv[i] + j;
			const int over = argv[i] != own || p > own_end;
			if (p > cur_end && !(contiguous && p >= argv[1] && p < lim)) break;      /* nothing verified to read there: stop like at a terminator */
			const char c = *p;
			if (!c) break;
			int32_t* iv = c == 'm' ? o->match : c == 'x' ? o->mismatch : c == 'o' ? o->gap_open : c == 'e' ? o->gap_ext : c == 'f' ? o->filter :
			              (!over && c == 'b') ? o->batch : (!over && c == 'g') ? o->gpus : 0;
			int hit = 0;
			if (iv || c == 'a') {
				if (i + 1 < argc && argv[i + 1][0] != '-') {
					if (iv) *iv = atoi(argv[i + 1]); else *o->mat_name = argv[i + 1];
					i++; hit = 1; cur_end = argv[i] + strlen(argv[i]);
				}
			} else if (c == 'p') { *o->protein = 1; hit = 1; } else if (c == 'c') { *o->path = 1; hit = 1; } else if (c == 'r') { *o->reverse = 1; hit = 1; }
			else if (c == 's') { *o->sam = 1; hit = 1; } else if (c == 'h') { *o->header = 1; hit = 1; }
			if (hit && over && nn + 48 < sizeof note) nn += (size_t)snprintf(note + nn, sizeof note - nn, hit && (iv || c == 'a') ? " -%c %.24s" : " -%c", c, argv[i]);
		}
		if (nn) {
			fprintf(stderr, "ssw_test_gpu: note: the option scan behind \"%s\" ran on into the following arguments, as the reference's ssw_test does "
			                "(src/main.c:253-300), and took from them:%s.  Give value options first and a flag (e.g. -c) last, or SSW_CLI_ARGS=getopt.\n", own, note);
			nn = 0; note[0] = 0;
		}
	}
	int first = 1;
	while (first < argc && argv[first][0] == '-') first += (argv[first][1] && strchr("mxoeafbg", argv[first][1])) ? 2 : 1;
	if (first + 2 > argc) return 1;
	files[0] = argv[first]; files[1] = argv[first + 1];
	return 0;
}

int main(int argc, char* const argv[])
{
	/* stdout buffer first, before ANY output (the SAM header below): setvbuf after I/O on the stream is undefined (round-5 advisor) */
	static char obuf[1 << 22];
	setvbuf(stdout, obuf, _IOFBF, sizeof obuf);
	int32_t match = 2, mismatch = 2, gap_open = 3, gap_ext = 1, path = 0, reverse = 0, n = 5, sam = 0, protein = 0, header = 0, filter = 0;
	int32_t batch = 65536, gpus = 0;
	const char* mat_name = 0;
	const char* files[2] = {0, 0};
	cli_opts o = {&match, &mismatch, &gap_open, &gap_ext, &filter, &batch, &gpus, &protein, &path, &reverse, &sam, &header, &mat_name};
	const char* mode = getenv("SSW_CLI_ARGS");
	if (mode && !strcmp(mode, "getopt")) { if (scan_args_getopt(argc, argv, &o, files)) { usage(); return 1; } }
	else if (scan_args_reference(argc, argv, &o, files)) { usage(); return 1; }
	if (batch < 1) batch = 1;
	cli_trace_on = getenv("SSW_CLI_TRACE") != 0; cli_t0 = cli_now();
	init_tables();

	int8_t dna[25]; int32_t k = 0;
	for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) dna[k++] = i == j ? match : -mismatch; dna[k++] = 0; }
	for (int j = 0; j < 5; ++j) dna[k++] = 0;
	const int8_t* mat = dna; int8_t* mat_file = 0; const int8_t* table = nt_code;
	if (protein && !mat_name) { n = 24; table = aa_code; mat = blosum50; }
	else if (mat_name) { if (load_matrix(mat_name, &mat_file, &n)) return 1; mat = mat_file; table = aa_code; }
	/* -r follows the reference's conditions (src/main.c:457-490): reverse-complement profiles exist only for a 5-letter matrix
	   and are only aligned without -p; with the 24-letter matrices the run stops at the first read; other sizes ignore -r */
	const int reverse_fatal = reverse && n == 24;
	if (reverse && (n != 5 || protein)) reverse = 0;

	/* the reference's order (src/main.c:434-449): the query file is opened first -- a failure ends the run before a byte is written -- */
	gzFile qf = gzopen(files[1], "r");
	if (!qf) { fprintf(stderr, "gzopen of '%s' failed.\n", files[1]); return EXIT_FAILURE; }
	/* -- and the target file is read once.  One that cannot be opened is a target set without sequences there (kseq reads nothing from a
	   NULL handle, src/main.c:493-495): no alignment lines, exit code 0; said on stderr here */
	reader tr; memset(&tr, 0, sizeof tr);
	tr.f = gzopen(files[0], "r");
	record* targets = 0; int32_t nt = 0, capt = 0;
	if (!tr.f) fprintf(stderr, "ssw_test_gpu: gzopen of the target file '%s' failed: no target sequences.\n", files[0]);
	else {
		for (record rec; read_record(&tr, &rec); ) {
			if (nt == capt) { capt = capt ? capt * 2 : 16; targets = (record*)xrealloc(targets, sizeof(record) * capt); }
			targets[nt++] = rec;
		}
		gzclose(tr.f);
	}
	CLI_TRACE("target file parsed: %d sequences", nt);
	if (sam && header && path) {
		fprintf(stdout, "@HD\tVN:1.4\tSO:queryname\n");
		for (int32_t t = 0; t < nt; ++t) fprintf(stdout, "@SQ\tSN:%s\tLN:%d\n", targets[t].name, targets[t].len);
	} else if (sam && !path) {
		fprintf(stderr, "SAM format output is only available together with option -c.\n");
		sam = 0;
	}
	int64_t* toff = (int64_t*)xmalloc(sizeof(int64_t) * ((size_t)nt + 1));
	toff[0] = 0;
	for (int32_t t = 0; t < nt; ++t) toff[t + 1] = toff[t] + targets[t].len;
	int8_t* tcodes = (int8_t*)xmalloc((size_t)toff[nt] + 1);
	for (int32_t t = 0; t < nt; ++t) for (int32_t i = 0; i < targets[t].len; ++i) tcodes[toff[t] + i] = table[(int)targets[t].seq[i] & 127];

	ssw_gpu_ctx* g = ssw_gpu_open(getenv("SSW_GPU_DEVICE") ? atoi(getenv("SSW_GPU_DEVICE")) : 0);
	if (!g) { fprintf(stderr, "ssw_test_gpu: %s\n", ssw_gpu_last_error(0)); return EXIT_FAILURE; }
	CLI_TRACE("device context open");
	/* Scratch budget of a command-line run: 16 GiB unless SSW_GPU_CM_BUDGET_MB says otherwise.  A run pays for its allocations itself: the
	   whole-HBM budget of ssw_gpu_set_budget_exclusive buys 1.5 % in steady state over the library's default 64 GiB and 4.5 % over 16 GiB
	   (10 397 / 10 236 / 9 947 GCUPS on config 2), but its first allocation costs 3.7 s, and right after another process released tens of
	   gigabytes even 2 x 30 GB can take seconds (the driver hands out scrubbed memory): profiles/round5_budget_sweep_config2.txt,
	   round5_cli_end_to_end.json. */
	/* ... and never more than the library's own default for this device (half of the free HBM: a shared or partitioned device, round-5 advisor) */
	if (!getenv("SSW_GPU_CM_BUDGET_MB") && ssw_gpu_get_budget(g) > ((size_t)16 << 30)) ssw_gpu_set_budget(g, (size_t)16 << 30);
	ssw_gpu_seqs* T = ssw_gpu_seqs_upload(g, tcodes, toff, nt);
	if (!T) { fprintf(stderr, "ssw_test_gpu: %s\n", ssw_gpu_last_error(g)); return EXIT_FAILURE; }
	CLI_TRACE("targets resident");
	/* -g N: the per-GPU work queues of the library (ssw_gpu_pool): the target set is replicated, read blocks are pulled by N workers */
	ssw_gpu_pool* gp = 0;
	if (gpus > 0) {
		const int ndev = ssw_gpu_device_count();
		int* devs = (int*)xmalloc(sizeof(int) * (size_t)gpus);
		for (int i = 0; i < gpus; ++i) devs[i] = i % (ndev > 0 ? ndev : 1);
		gp = ssw_gpu_pool_open(devs, gpus);
		free(devs);
		if (!gp) { fprintf(stderr, "ssw_test_gpu: %s\n", ssw_gpu_last_error(0)); return EXIT_FAILURE; }
		if (ssw_gpu_pool_set_targets(gp, tcodes, toff, nt)) { fprintf(stderr, "ssw_test_gpu: %s\n", ssw_gpu_pool_last_error(gp)); return EXIT_FAILURE; }
	}

	/* ---- three stages on three threads, two batches in flight between each pair of them (reference loop: src/main.c:462-526):
	       parse (file -> records + one ASCII block)  ->  device (upload + translation + alignment)  ->  format + write (this thread) */
	pipeline pl; memset(&pl, 0, sizeof pl);
	pl.qf = qf; pl.batch = batch; pl.reverse = reverse; pl.reverse_fatal = reverse_fatal; pl.table = table; pl.nt_table = table == nt_code;
	pl.g = g; pl.gp = gp; pl.T = T; pl.nt = nt;
	pl.prm.mat = mat; pl.prm.n = n; pl.prm.gapO = (uint8_t)gap_open; pl.prm.gapE = (uint8_t)gap_ext; pl.prm.flag = path ? 2 : 0; pl.prm.filters = (uint16_t)filter;
	pl.prm.filterd = 0; pl.prm.maskLen = -1; pl.prm.score_size = 2; pl.prm.mark_mismatch = sam ? 1 : 0;
	queue_init(&pl.parsed); queue_init(&pl.aligned);
	const clock_t t_start = clock();
	pthread_t th_parse, th_dev;
	if (pthread_create(&th_parse, 0, stage_parse, &pl) || pthread_create(&th_dev, 0, stage_device, &pl)) { fprintf(stderr, "ssw_test_gpu: cannot start the pipeline threads\n"); return EXIT_FAILURE; }
	int rc_main = 0;
	for (;;) {
		work* w = (work*)queue_pop(&pl.aligned);
		if (!w) break;                                  /* end of input */
		if (w->fatal) { rc_main = w->fatal; free_work(w, reverse); break; }      /* (message already on stderr) */
		const int32_t nr = w->nr;
		const ssw_gpu_result* res = w->res; const ssw_gpu_result* res_rc = reverse ? w->res + (int64_t)nr * nt : 0;
		char* rcseq = 0; size_t rccap = 0;                /* reverse-complement text / codes of the read being printed: only reads whose '-' alignment wins need them */
		int8_t* codes = 0; size_t ccap = 0;
		for (int32_t q = 0; q < nr; ++q) {
			const record* rd = &w->reads[q];
			int have_rc = 0;
			for (int32_t t = 0; t < nt; ++t) {
				const ssw_gpu_result* r = &res[(int64_t)q * nt + t];
				const ssw_gpu_result* rr = reverse ? &res_rc[(int64_t)q * nt + t] : 0;
				if (r->status != 0) {
					fprintf(stderr, "Warning: Alignment between the following sequences is failed.\nref_name: %s\nread_name: %s\n\n", targets[t].name, rd->name);
					continue;
				}
				const int minus = rr && rr->status == 0 && rr->score1 > r->score1 && rr->score1 >= filter;
				const ssw_gpu_result* win = minus ? rr : r;
				if (!minus && !(r->score1 > 0 && r->score1 >= filter)) {
					if (r->score1 <= 0) fprintf(stderr, "There is no identical residue between the following reference and read seqeunces.\nref_name: %s\nread_name: %s\n\n", targets[t].name, rd->name);
					continue;
				}
				const char* seq = rd->seq;
				if (minus) {
					if (!have_rc) {
						if (rccap < (size_t)rd->len + 1) { rccap = (size_t)rd->len * 2 + 64; rcseq = (char*)xrealloc(rcseq, rccap); }
						reverse_complement(rd->seq, rd->len, rcseq); have_rc = 1;
					}
					seq = rcseq;
				}
				s_align* a = ssw_gpu_result_to_align(win, w->pool);
				if (a->flag == 2) fprintf(stderr, minus ? "Warning: The reverse compliment alignment of the following sequences may miss a small part.\nref_seq: %s\nread_seq: %s\n\n"
				                                        : "Warning: The alignment of the following sequences may miss a small part.\nref_seq: %s\nread_seq: %s\n\n", targets[t].name, rd->name);
				const int32_t nm_dev = sam && win->cigarLen > 0 ? win->edit_distance : -1;
				const int8_t* read_num = 0;
				if (sam && nm_dev < 0) {      /* (a traceback the reference gives up on: its mark_mismatch runs on the empty CIGAR and wants the codes) */
					if (ccap < (size_t)rd->len + 1) { ccap = (size_t)rd->len * 2 + 64; codes = (int8_t*)xrealloc(codes, ccap); }
					for (int32_t i = 0; i < rd->len; ++i) codes[i] = table[(int)seq[i] & 127];
					read_num = codes;
				}
				write_alignment(a, &targets[t], rd, seq, tcodes + toff[t], read_num, table, minus, sam, nm_dev);
				align_destroy(a);
			}
		}
		free(rcseq); free(codes);
		CLI_TRACE("wrote a batch of %d reads", nr);
		free_work(w, reverse);
	}
	fflush(stdout);
	CLI_TRACE("stdout flushed");
	if (rc_main) exit(rc_main);      /* (the other stages may be blocked on a full queue: the process ends here, as the reference's does) */
	pthread_join(th_parse, 0); pthread_join(th_dev, 0);
	fprintf(stderr, "CPU time: %f seconds\n", ((float)(clock() - t_start)) / CLOCKS_PER_SEC);
	ssw_gpu_seqs_free(T);
	ssw_gpu_pool_close(gp);
	ssw_gpu_close(g);
	for (int32_t t = 0; t < nt; ++t) free_record(&targets[t]);
	free(targets); free(toff); free(tcodes); free(mat_file);
	return 0;
}
