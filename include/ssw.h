/*
 * ssw.h -- C-ABI of the MI355X-native Smith-Waterman library.
 *
 * Drop-in for the header of mengyao/Complete-Striped-Smith-Waterman-Library
 * (reference src/ssw.h): same symbols, same argument meaning, same result
 * struct layout, so the reference's own callers (src/main.c "ssw_test",
 * src/example.c, src/ssw_cpp.cpp, src/ssw_lib.py via ctypes, src/sswjni.c)
 * build and run unchanged against libssw.so from this repository.  The work
 * behind ssw_align() runs as HIP kernels on gfx950; see DESIGN.md.
 *
 * Each declaration cites the reference interface it replaces.
 */
#ifndef SSW_H
#define SSW_H

#include <stdio.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* BAM-style CIGAR packing (reference src/ssw.h:29-34): op letter table,
   4-bit op code in the low bits, run length above it. */
#define MAPSTR "MIDNSHP=X"
#ifndef BAM_CIGAR_SHIFT
#define BAM_CIGAR_SHIFT 4u
#endif

/* ASCII -> CIGAR op code (reference src/ssw.c:127-160). */
extern const uint8_t encoded_ops[];

/* Opaque query handle (reference src/ssw.h:37-38).  It BORROWS the read and
   matrix pointers given to ssw_init(): both must outlive it (reference
   src/ssw.c:842-843). */
struct _profile;
typedef struct _profile s_profile;

/*
 * Alignment result (reference src/ssw.h:55-66; layout pinned by ctypes users,
 * reference src/ssw_lib.py:61-69).  All coordinates are 0-based and inclusive.
 *   score1       best local alignment score
 *   score2       heuristic second-best score (0 when maskLen < 15)
 *   ref_begin1   -1 when begin positions were not requested / computed
 *   ref_end1
 *   read_begin1  -1 when not computed
 *   read_end1
 *   ref_end2     end of the second-best alignment on the target (-1 when maskLen < 15)
 *   cigar        malloc()ed BAM-packed ops (M=0, I=1, D=2), NULL when no path is returned
 *   cigarLen
 *   flag         0 ok; 1 traceback failed (cigar == NULL); 2 path may miss a part
 */
typedef struct {
	uint16_t score1;
	uint16_t score2;
	int32_t ref_begin1;
	int32_t ref_end1;
	int32_t read_begin1;
	int32_t read_end1;
	int32_t ref_end2;
	uint32_t* cigar;
	int32_t cigarLen;
	uint16_t flag;
} s_align;

/*
 * Register a query (replaces reference ssw_init, src/ssw.h:86, src/ssw.c:826-847).
 *   read       residue codes in [0, n)
 *   mat        n*n substitution scores, indexed mat[target_code * n + read_code]
 *   score_size 0: scores are known to stay below 255 (8-bit semantics only);
 *              1: 16-bit semantics only; 2: decide per alignment (8-bit rules
 *              unless the score reaches 255 - bias, exactly like the reference's
 *              byte -> word fallback).
 * The returned object is freed with init_destroy().  Like the reference (src/ssw.c:842-843) it BORROWS `read` and
 * `mat`: both must stay valid until init_destroy().  readLen == 0 is legal (every alignment of it is the empty
 * record); a negative length or missing arrays return NULL with a message.
 */
s_profile* ssw_init(const int8_t* read, const int32_t readLen, const int8_t* mat, const int32_t n,
                    const int8_t score_size);

/* Replaces reference init_destroy (src/ssw.h:91, src/ssw.c:849-853). */
void init_destroy(s_profile* p);

/*
 * Align the registered query to one target (replaces reference ssw_align,
 * src/ssw.h:126-134, src/ssw.c:855-977).
 *   weight_gapO / weight_gapE  penalty of the first / every further gap base (absolute values)
 *   flag     bit 0x08: report begin positions; 0x04: CIGAR only if both spans <= filterd;
 *            0x02: CIGAR only if score1 >= filters; 0x01: always report the CIGAR.
 *            0: scores and end positions only.  (Exact gating: reference src/ssw.c:916, 938.)
 *   maskLen  second-best hits closer than this to ref_end1 are ignored; < 15 disables score2.
 * Returns a calloc()ed result (free with align_destroy) or NULL with a message on stderr when
 * the reference would (8-bit overflow without 16-bit semantics enabled; unusable profile) or when
 * the GPU path cannot run (no device, unsupported parameters) -- there is no CPU fallback.
 * Re-entrant like the reference: every calling thread has its own implicit device context (devices are handed
 * out round-robin), so concurrent calls -- also on one shared profile -- are legal.  A call is a handful of small kernel
 * launches (milliseconds): the same target handed in again is not re-uploaded, but throughput needs the batch ABI of
 * ssw_gpu.h, which aligns whole read sets per call.
 */
s_align* ssw_align(const s_profile* prof, const int8_t* ref, int32_t refLen, const uint8_t weight_gapO,
                   const uint8_t weight_gapE, const uint8_t flag, const uint16_t filters, const int32_t filterd,
                   const int32_t maskLen);

/* Replaces reference align_destroy (src/ssw.h:139, src/ssw.c:979-982). */
void align_destroy(s_align* a);

/*
 * Rewrite a CIGAR in place: M -> runs of '=' / 'X', soft clips added at both ends; returns the
 * edit distance (mismatches + inserted + deleted bases).  The old buffer is freed and replaced.
 * Replaces reference mark_mismatch (src/ssw.h:157-164, src/ssw.c:1019-1074).
 */
int32_t mark_mismatch(int32_t ref_begin1, int32_t read_begin1, int32_t read_end1, const int8_t* ref,
                      const int8_t* read, int32_t readLen, uint32_t** cigar, int32_t* cigarLen);

/* CIGAR word helpers (reference src/ssw.h:171-190). */
static inline uint32_t to_cigar_int(uint32_t length, unsigned char op_letter)
{
	return (length << BAM_CIGAR_SHIFT) | (encoded_ops[op_letter]);
}
static inline char cigar_int_to_op(uint32_t cigar_int)
{
	uint32_t code = cigar_int & 0xfU;
	return code > 8 ? 'M' : MAPSTR[code];
}
static inline uint32_t cigar_int_to_len(uint32_t cigar_int)
{
	return cigar_int >> BAM_CIGAR_SHIFT;
}

#ifdef __cplusplus
}
#endif
#endif /* SSW_H */
