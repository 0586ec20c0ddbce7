/*
 * ssw_cli.c -- "ssw_test_gpu": batched command-line front end with the options and the byte-identical stdout of the
 * reference's ssw_test (reference src/main.c:395-547; output formats of ssw_write, src/main.c:118-245), SURVEY 8f-1.
 *
 * What differs from the reference CLI is only the loop structure: the target file is parsed ONCE and stays resident
 * in HBM, reads are taken in batches, and each batch is one ssw_gpu_align_batch() call (two with -r) instead of
 * reads x targets synchronous ssw_align() calls.  Output order is the reference's: reads outer, targets inner.
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

int main(int argc, char* const argv[])
{
	int32_t match = 2, mismatch = 2, gap_open = 3, gap_ext = 1, path = 0, reverse = 0, n = 5, sam = 0, protein = 0, header = 0, filter = 0;
	int32_t batch = 65536, gpus = 0;
	const char* mat_name = 0;
	const char* files[2]; int nfiles = 0;
	for (int i = 1; i < argc; ++i) {
		if (argv[i][0] != '-' || !argv[i][1]) { if (nfiles < 2) files[nfiles++] = argv[i]; continue; }
		for (int j = 1; argv[i][j]; ++j) {
			const char o = argv[i][j];
			if (o == 'p') protein = 1; else if (o == 'c') path = 1; else if (o == 'r') reverse = 1;
			else if (o == 's') sam = 1; else if (o == 'h') header = 1;
			else if (o == 'm' || o == 'x' || o == 'o' || o == 'e' || o == 'f' || o == 'a' || o == 'b' || o == 'g') {
				const char* val = argv[i][j + 1] ? &argv[i][j + 1] : (i + 1 < argc ? argv[++i] : 0);
				if (!val) { usage(); return 1; }
				if (o == 'm') match = atoi(val); else if (o == 'x') mismatch = atoi(val); else if (o == 'o') gap_open = atoi(val);
				else if (o == 'e') gap_ext = atoi(val); else if (o == 'f') filter = atoi(val); else if (o == 'b') batch = atoi(val); else if (o == 'g') gpus = atoi(val);
				else mat_name = val;
				break;
			}
		}
	}
	if (nfiles < 2) { usage(); return 1; }
	if (batch < 1) batch = 1;
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

	/* the target file is read once */
	reader tr; memset(&tr, 0, sizeof tr);
	tr.f = gzopen(files[0], "r");
	if (!tr.f) { fprintf(stderr, "gzopen of '%s' failed.\n", files[0]); return EXIT_FAILURE; }
	record* targets = 0; int32_t nt = 0, capt = 0;
	for (record rec; read_record(&tr, &rec); ) {
		if (nt == capt) { capt = capt ? capt * 2 : 16; targets = (record*)xrealloc(targets, sizeof(record) * capt); }
		targets[nt++] = rec;
	}
	gzclose(tr.f);
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
	ssw_gpu_seqs* T = ssw_gpu_seqs_upload(g, tcodes, toff, nt);
	if (!T) { fprintf(stderr, "ssw_test_gpu: %s\n", ssw_gpu_last_error(g)); return EXIT_FAILURE; }
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

	reader qr; memset(&qr, 0, sizeof qr);
	qr.f = gzopen(files[1], "r");
	if (!qr.f) { fprintf(stderr, "gzopen of '%s' failed.\n", files[1]); exit(EXIT_FAILURE); }
	const clock_t t_start = clock();
	record* reads = (record*)xmalloc(sizeof(record) * (size_t)batch);
	for (;;) {
		int32_t nr = 0; int64_t total = 0;
		while (nr < batch && read_record(&qr, &reads[nr])) { total += reads[nr].len; ++nr; }
		if (nr == 0) break;
		if (reverse_fatal) { fprintf(stderr, "Reverse complement alignment is not available for protein sequences. \n"); return 1; }
		int64_t* qoff = (int64_t*)xmalloc(sizeof(int64_t) * ((size_t)nr + 1));
		int8_t* qcodes = (int8_t*)xmalloc((size_t)total + 1);
		int8_t* rcodes = reverse ? (int8_t*)xmalloc((size_t)total + 1) : 0;
		char** rcseq = reverse ? (char**)xmalloc(sizeof(char*) * (size_t)nr) : 0;
		qoff[0] = 0;
		for (int32_t q = 0; q < nr; ++q) {
			qoff[q + 1] = qoff[q] + reads[q].len;
			for (int32_t i = 0; i < reads[q].len; ++i) qcodes[qoff[q] + i] = table[(int)reads[q].seq[i] & 127];
			if (reverse) {
				rcseq[q] = (char*)xmalloc((size_t)reads[q].len + 1);
				reverse_complement(reads[q].seq, reads[q].len, rcseq[q]);
				for (int32_t i = 0; i < reads[q].len; ++i) rcodes[qoff[q] + i] = table[(int)rcseq[q][i] & 127];
			}
		}
		ssw_gpu_params p;
		p.mat = mat; p.n = n; p.gapO = (uint8_t)gap_open; p.gapE = (uint8_t)gap_ext; p.flag = path ? 2 : 0; p.filters = (uint16_t)filter;
		p.filterd = 0; p.maskLen = -1; p.score_size = 2; p.mark_mismatch = sam ? 1 : 0;
		ssw_gpu_result* res = (ssw_gpu_result*)xmalloc(sizeof(ssw_gpu_result) * (size_t)nr * (size_t)(nt ? nt : 1));
		ssw_gpu_result* res_rc = reverse ? (ssw_gpu_result*)xmalloc(sizeof(ssw_gpu_result) * (size_t)nr * (size_t)(nt ? nt : 1)) : 0;
		uint32_t *pool = 0, *pool_rc = 0; int64_t words = 0;
		if (gp) {   /* reads stay on the host; every worker uploads the blocks it takes */
			if (ssw_gpu_pool_align(gp, qcodes, qoff, nr, 0, 0, nt, &p, res, &pool, &words) ||
			    (reverse && ssw_gpu_pool_align(gp, rcodes, qoff, nr, 0, 0, nt, &p, res_rc, &pool_rc, &words))) {
				fprintf(stderr, "ssw_test_gpu: %s\n", ssw_gpu_pool_last_error(gp)); return EXIT_FAILURE;
			}
		} else {
			/* residue translation and (with -r) the reverse complement run on the device; the host copies made
			   above are only used for printing (SAM needs the codes for mark_mismatch) */
			char* qtext = (char*)xmalloc((size_t)total + 1);
			for (int32_t q = 0; q < nr; ++q) memcpy(qtext + qoff[q], reads[q].seq, (size_t)reads[q].len);
			ssw_gpu_seqs* Qs = ssw_gpu_seqs_upload_ascii(g, qtext, qoff, nr, table);
			free(qtext);
			if (!Qs || ssw_gpu_align_batch(g, Qs, T, 0, nt, &p, res, &pool, &words)) { fprintf(stderr, "ssw_test_gpu: %s\n", ssw_gpu_last_error(g)); return EXIT_FAILURE; }
			if (reverse) {
				/* (the device reverse complement works on nucleotide codes; with a matrix file the table is the file's own) */
				ssw_gpu_seqs* Qr = table == nt_code ? ssw_gpu_seqs_revcomp(g, Qs) : ssw_gpu_seqs_upload(g, rcodes, qoff, nr);
				if (!Qr || ssw_gpu_align_batch(g, Qr, T, 0, nt, &p, res_rc, &pool_rc, &words)) { fprintf(stderr, "ssw_test_gpu: %s\n", ssw_gpu_last_error(g)); return EXIT_FAILURE; }
				ssw_gpu_seqs_free(Qr);
			}
			ssw_gpu_seqs_free(Qs);
		}
		for (int32_t q = 0; q < nr; ++q)
			for (int32_t t = 0; t < nt; ++t) {
				const ssw_gpu_result* r = &res[(int64_t)q * nt + t];
				const ssw_gpu_result* rr = reverse ? &res_rc[(int64_t)q * nt + t] : 0;
				if (r->status != 0) {
					fprintf(stderr, "Warning: Alignment between the following sequences is failed.\nref_name: %s\nread_name: %s\n\n", targets[t].name, reads[q].name);
					continue;
				}
				if (rr && rr->status == 0 && rr->score1 > r->score1 && rr->score1 >= filter) {
					s_align* a = ssw_gpu_result_to_align(rr, pool_rc);
					if (a->flag == 2) fprintf(stderr, "Warning: The reverse compliment alignment of the following sequences may miss a small part.\nref_seq: %s\nread_seq: %s\n\n", targets[t].name, reads[q].name);
					write_alignment(a, &targets[t], &reads[q], rcseq[q], tcodes + toff[t], rcodes + qoff[q], table, 1, sam, sam && rr->cigarLen > 0 ? rr->edit_distance : -1);
					align_destroy(a);
				} else if (r->score1 > 0 && r->score1 >= filter) {
					s_align* a = ssw_gpu_result_to_align(r, pool);
					if (a->flag == 2) fprintf(stderr, "Warning: The alignment of the following sequences may miss a small part.\nref_seq: %s\nread_seq: %s\n\n", targets[t].name, reads[q].name);
					write_alignment(a, &targets[t], &reads[q], reads[q].seq, tcodes + toff[t], qcodes + qoff[q], table, 0, sam, sam && r->cigarLen > 0 ? r->edit_distance : -1);
					align_destroy(a);
				} else if (r->score1 <= 0) {
					fprintf(stderr, "There is no identical residue between the following reference and read seqeunces.\nref_name: %s\nread_name: %s\n\n", targets[t].name, reads[q].name);
				}
			}
		free(pool); free(pool_rc); free(res); free(res_rc); free(qoff); free(qcodes); free(rcodes);
		for (int32_t q = 0; q < nr; ++q) { if (reverse) free(rcseq[q]); free_record(&reads[q]); }
		free(rcseq);
	}
	fprintf(stderr, "CPU time: %f seconds\n", ((float)(clock() - t_start)) / CLOCKS_PER_SEC);
	gzclose(qr.f);
	free(reads);
	ssw_gpu_seqs_free(T);
	ssw_gpu_pool_close(gp);
	ssw_gpu_close(g);
	for (int32_t t = 0; t < nt; ++t) free_record(&targets[t]);
	free(targets); free(toff); free(tcodes); free(mat_file);
	return 0;
}
