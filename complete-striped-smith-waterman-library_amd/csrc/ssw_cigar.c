/*
 * ssw_cigar.c -- host-side CIGAR utilities of the ssw.h ABI (plain C, no device work).
 *
 * Replaces: encoded_ops (reference src/ssw.c:127-160), mark_mismatch (src/ssw.c:1019-1074) and the two
 * helpers the reference exports by accident and that therefore belong to its dynamic symbol table,
 * add_cigar (src/ssw.c:984-992) and store_previous_m (src/ssw.c:994-1009).
 */
#include <stdlib.h>
#include "ssw.h"

/* ASCII -> op code; everything that is not one of "MIDNSHP=X" maps to 0 ('M') */
const uint8_t encoded_ops[128] = {
	['M'] = 0, ['I'] = 1, ['D'] = 2, ['N'] = 3, ['S'] = 4, ['H'] = 5, ['P'] = 6, ['='] = 7, ['X'] = 8
};

static uint32_t next_pow2(uint32_t x)
{
	--x; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16;
	return x + 1;
}

/* append one op, growing the buffer to the next power of two when full */
uint32_t* add_cigar(uint32_t* new_cigar, int32_t* p, int32_t* s, uint32_t length, char op)
{
	if (*p >= *s) {
		*s = (int32_t)next_pow2((uint32_t)*s + 1);
		new_cigar = (uint32_t*)realloc(new_cigar, (size_t)*s * sizeof(uint32_t));
	}
	new_cigar[(*p)++] = to_cigar_int(length, (unsigned char)op);
	return new_cigar;
}

/* flush a pending '=' run (when a mismatch or a gap follows) or a pending 'X' run (match or gap follows) */
uint32_t* store_previous_m(int8_t choice, uint32_t* length_m, uint32_t* length_x, int32_t* p, int32_t* s,
                           uint32_t* new_cigar)
{
	if (*length_m && choice != 1) {
		new_cigar = add_cigar(new_cigar, p, s, *length_m, '=');
		*length_m = 0;
	} else if (*length_x && choice != 2) {
		new_cigar = add_cigar(new_cigar, p, s, *length_x, 'X');
		*length_x = 0;
	}
	return new_cigar;
}

int32_t mark_mismatch(int32_t ref_begin1, int32_t read_begin1, int32_t read_end1, const int8_t* ref,
                      const int8_t* read, int32_t readLen, uint32_t** cigar, int32_t* cigarLen)
{
	int32_t edits = 0, used = 0, cap = *cigarLen + 2;
	uint32_t* out = (uint32_t*)malloc((size_t)cap * sizeof(uint32_t));
	uint32_t run_eq = 0, run_ne = 0;
	const int8_t* t = ref + ref_begin1;
	const int8_t* q = read + read_begin1;

	if (read_begin1 > 0) out[used++] = to_cigar_int((uint32_t)read_begin1, 'S');
	for (int32_t i = 0; i < *cigarLen; ++i) {
		const char op = cigar_int_to_op((*cigar)[i]);
		const int32_t len = (int32_t)cigar_int_to_len((*cigar)[i]);
		if (op == 'M') {
			for (int32_t k = 0; k < len; ++k, ++t, ++q) {
				if (*t != *q) {
					++edits;
					out = store_previous_m(2, &run_eq, &run_ne, &used, &cap, out);
					++run_ne;
				} else {
					out = store_previous_m(1, &run_eq, &run_ne, &used, &cap, out);
					++run_eq;
				}
			}
		} else if (op == 'I' || op == 'D') {
			if (op == 'I') q += len; else t += len;
			edits += len;
			out = store_previous_m(0, &run_eq, &run_ne, &used, &cap, out);
			out = add_cigar(out, &used, &cap, (uint32_t)len, op);
		}
	}
	out = store_previous_m(0, &run_eq, &run_ne, &used, &cap, out);
	if (readLen - read_end1 - 1 > 0) out = add_cigar(out, &used, &cap, (uint32_t)(readLen - read_end1 - 1), 'S');
	*cigarLen = used;
	free(*cigar);
	*cigar = out;
	return edits;
}
