/*
 * ssw_kernels.hip -- gfx950 kernels of the Smith-Waterman hot path + the thin C shim the
 * C host driver (ssw_host.c) calls.
 *
 * What is computed (reference file:line is mengyao/Complete-Striped-Smith-Waterman-Library src/ssw.c):
 *   k_fill        the DP fill of sw_sse2_byte / sw_sse2_word (197-386 / 412-588) for queries up to 384 residues: per target
 *                 column the maximum H over the (zero-padded) query, for both padding rules at once, plus the maximum of every
 *                 group of 16 columns
 *   k_chainq      the same fill for longer queries, cut into row strips of 64 x R rows that a persistent launch draws from a
 *                 work queue (k_chainx: the 16-lane form, strips in sequence); its window mode does the "where" passes
 *   k_filldb      database search: fill + best cell + reduction of one query pair against 16 short targets in one launch
 *   k_reduce_seg  the bookkeeping around the fill: best score / first best column (317-340, 523-542) and the masked
 *   / k_reduce    second-best scan (368-381 / 570-583), plus ssw_align's choice between 8-bit and 16-bit rules (881-899) and its
 *                 early exits (900-916) -- over the group maxima / over the strip kernel's columns
 *   k_capture     the "where" passes of short queries: read_end1 (342-351 / 544-553) and the reverse pass that finds the
 *                 begin position (ssw_align 919-935, sw_sse2_* with ref_dir = 1 and `terminate`)
 *   k_literal     both SSE2 kernels re-enacted lane for lane, for gapO <= gapE (the reference's answer depends on its layout there)
 *   k_trace /     banded_sw (590-783) + cigar_alignment_score (785-811) + ssw_align's band retry (941-973): one thread, or a
 *   k_trace_wave  team of 1 / 4 / 16 wavefronts per alignment
 *   k_mark, k_prep, k_gather   mark_mismatch (1019-1074) on the device, residue translation / reverse complement (main.c:84-116),
 *                 CIGAR pool compaction
 *
 * How (MI355X-first, not the SSE2 layout): the reference keeps one query in 16/8 SIMD lanes with a
 * striped layout and repairs the vertical (F) dependency with a lazy loop.  Here a DPP row of 16
 * lanes is a systolic chain: lane l owns R consecutive query rows, processes target column (step - l)
 * at each step and hands H / F / running column max to lane l+1 with row_shr DPP moves, so the F
 * dependency is resolved exactly with no lazy loop.  Every VGPR carries TWO queries (packed 2 x int16,
 * saturating VOP3P arithmetic).  A wavefront runs 4 chains, a 256-thread workgroup 16 chains that
 * share one query-pair score profile in LDS and take 16 different tiles of the target.  Tiles start
 * `halo` columns early from the all-zero state, which reproduces the untiled state exactly
 * (DESIGN.md "exact halo").  See DESIGN.md for the proof sketches and the roofline.
 */
#include "lanes.h"
#include "ssw_dev.h"

#define DEAD2 0x80008000u   /* packed (-32768, -32768): a score that pins H to max(E, F) */

/* virtual ids (ssw_vmap, ssw_dev.h): the query whose residues job id `v` uses, and the start of its target */
SSW_DEV int vm_query(const ssw_vmap& m, int v) { return m.vq ? m.vq[v] : v; }
SSW_DEV const int8_t* vm_target(const ssw_vmap& m, const int8_t* tgt, int v) { return m.vq ? m.tcodes + m.toff[m.vt[v]] : tgt; }

/* ------------------------------------------------------------------------------------------------
 * LDS map of one chain-group kernel:
 *   [0, prof_bytes)                      score profile(s): word ((b*C + c)*16 + l)*4 + k holds the packed
 *                                        scores of residue b against rows l*R + 4c + k of the two queries
 *   per chain: 160 B   target ring       80 x u16 profile byte-offsets of target columns (64 + 16 mirrored)
 *              256 B   out16 ring        64 x u32 finished column maxima (all rows)
 *              256 B   out8 ring         64 x u32 finished column maxima (first 16R-8 rows)
 * ------------------------------------------------------------------------------------------------ */
#define RING_BYTES 160
#define CHAIN_BYTES (RING_BYTES + 256 + 256)

template <int R> struct ChainGeom {
	static constexpr int C = (R + 3) / 4;          /* 16-byte profile chunks per lane and residue */
	static constexpr int PSTRIDE = C * 256;        /* profile bytes per residue */
	static constexpr int A8 = 16 * R - 8;          /* rows that exist under 16-bit rules when the read is padded */
	static constexpr int TAP = A8 / R;             /* lane whose running maximum covers exactly rows < A8 ... */
	static constexpr int K8 = A8 % R;              /* ... after K8 of its own rows */
};

/* build one packed profile: rows of query A in the low halves, query B in the high halves */
/* PM: 0 packed int16, 2 column frame (score + gapE per live row, FR_DEAD for dead rows: lanes.h) */
template <int R, int PM = 0>
SSW_DEV void build_profile(unsigned char* lds, u32 base, int first, int nthreads,
                           const int8_t* mat, int n,
                           const int8_t* qa, int lena, int reva,
                           const int8_t* qb, int lenb, int p16a = 0x7fffffff, int p16b = 0x7fffffff, int gapE = 0)
{
	constexpr int C = ChainGeom<R>::C;
	const int total = (n + 1) * C * 64;
	for (int w = first; w < total; w += nthreads) {
		const int b = w / (C * 64), rem = w - b * (C * 64);
		const int c = rem >> 6, l = (rem & 63) >> 2, k = rem & 3;
		const int r = c * 4 + k, row = l * R + r;
		u32 v;
		if (b == n) v = PM == 2 ? fr_pack(FR_DEAD, FR_DEAD) : DEAD2;
		else if (r >= R) v = 0;
		else {
			int lo = row < p16a ? 0 : -32768, hi = row < p16b ? 0 : -32768;   /* rows below a padded query are dead */
			if (row < lena) lo = mat[b * n + (reva ? qa[lena - 1 - row] : qa[row])];
			if (qb && row < lenb) hi = mat[b * n + qb[row]];
			if (PM == 2) v = fr_pack(lo == -32768 ? FR_DEAD : lo + gapE, hi == -32768 ? FR_DEAD : hi + gapE);
			else v = pk_make(lo, hi);
		}
		lds_st32(lds, base + (u32)w * 4u, v);
	}
}

/* one DP step of a chain lane: R rows of one target column for two packed queries */
/* column-frame form of rows [R0, R1) (lanes.h): 3 plain 32-bit adds + 3.5 packed maxima per row.  c1 = gapO - gapE (packed),
   fl = phi(column + 1): the floor that keeps E at "0" or above.
     h = max3(d + s', E, F)        t = h - c1        E' = max3(E, t, fl)        F' = max(F, t) - gapE        cm = max3(cm, h_r, h_r+1) */
/* OPEN: the last row leaves f BEFORE its "- gapE": the lane below subtracts while it takes the value over (xl_row_shr1_sub_keep) */
#ifndef FILL_GROUP_ADDS
#define FILL_GROUP_ADDS 0
#endif
template <int R, int R0, int R1, bool OPEN = false>
SSW_DEV void chain_rows_fr(const u32x4* sc, u32 (&H)[R], u32 (&E)[R], u32& d, u32& f, u32& cm, u32 c1, u32 gapE2, u32 fl)
{
#if FILL_GROUP_ADDS
	/* EXPERIMENT (round-5 verdict, item 5): the R independent `diag + score` adds of a step issued as ONE group ahead of the dependent chain
	   (they need only the previous column's H and this column's scores): profiles/round3_mix_issue_probe.txt prices grouped 32-bit adds at
	   2.3-3.0 cycles against 4.0 between maxima */
	if (R1 > R0) {
		u32 x[R];
#pragma unroll
		for (int r = R0; r < R1; ++r) x[r] = (r == R0 ? d : H[r - 1 >= 0 ? r - 1 : 0]) + sc[r >> 2][r & 3];
		d = H[R1 - 1];
		sched_fence();
#pragma unroll
		for (int r = R0; r < R1; ++r) {
			const u32 h = pk_max3_fr(x[r], E[r], f);
			const u32 t = h - c1;
			E[r] = pk_max3_fr(E[r], t, fl);
			f = pk_max(f, t);
			if (!(OPEN && r == R1 - 1)) f -= gapE2;
			if (((r - R0) & 1) == 1) cm = pk_max3_fr(cm, H[r - 1 >= 0 ? r - 1 : 0], h);
			else if (r == R1 - 1) cm = pk_max(cm, h);
			H[r] = h;
		}
	}
	return;
#endif
#pragma unroll
	for (int r = R0; r < R1; ++r) {
		const u32 hold = H[r];
		const u32 h = pk_max3_fr(d + sc[r >> 2][r & 3], E[r], f);
		const u32 t = h - c1;
		E[r] = pk_max3_fr(E[r], t, fl);
		f = pk_max(f, t);
		if (!(OPEN && r == R1 - 1)) f -= gapE2;
		if (((r - R0) & 1) == 1) cm = pk_max3_fr(cm, H[r - 1 >= 0 ? r - 1 : 0], h);      /* H[r-1] was just written: this pair's first row */
		else if (r == R1 - 1) cm = pk_max(cm, h);                                         /* odd row left over */
		H[r] = h;
		d = hold;
	}
}

/* FORM 0: plain int16 with the reference's saturation, 9 instructions per row (buckets whose scores may pass the frame form's range);
   3: column frame (scores + frame offsets < 31744), 6.5 of which 3 are 32-bit adds (gapO2 then carries gapO - gapE); f comes back OPEN (chain_rows_fr) */
template <int R, bool TRACK8, int FORM = 0>
SSW_DEV void chain_rows(const u32x4* sc, u32 (&H)[R], u32 (&E)[R], u32 d, u32& f, u32& cm, u32& ck,
                        u32 gapO2, u32 gapE2, u32 fl = 0)
{
	constexpr int K8 = ChainGeom<R>::K8;
	if (FORM == 3) {
		if (TRACK8) {
			chain_rows_fr<R, 0, K8>(sc, H, E, d, f, cm, gapO2, gapE2, fl);
			ck = cm;
			chain_rows_fr<R, K8, R, true>(sc, H, E, d, f, cm, gapO2, gapE2, fl);
		} else chain_rows_fr<R, 0, R, true>(sc, H, E, d, f, cm, gapO2, gapE2, fl);
		return;
	}
#pragma unroll
	for (int r = 0; r < R; ++r) {
		if (TRACK8 && r == K8) ck = cm;
		const u32 hold = H[r];
		const u32 s = sc[r >> 2][r & 3];
		const u32 h0 = pk_max(pk_adds(d, s), E[r]);   /* E >= 0 supplies the max(0, .) of local alignment */
		const u32 h = pk_max(h0, f);
		const u32 t0 = pk_subu(h0, gapO2);            /* gap opened from the F-free value */
		E[r] = pk_max(pk_subu(E[r], gapE2), t0);
		f = pk_max(pk_subu(f, gapE2), t0);
		cm = pk_max(cm, h);
		H[r] = h;
		d = hold;
	}
}

/* the 16 lanes of a chain write the finished maxima of traversal columns [base, base + 16) (both padding rules) and the
   maximum over the group: a 4-step row_ror butterfly on the packed values (all 16 lanes end up with it, lane 0 stores it) */
template <int R, int FORM>
SSW_DEV void fill_flush16(unsigned char* lds, u32 out16, u32 out8, int base, int l16, int store_from, int ncols,
                          uint32_t* o16, uint32_t* o8, uint32_t* g16, uint32_t* g8, int fr_base = 0, int fr_kmask = 0, int gapE = 0)
{
	typedef ChainGeom<R> G;
	const int tc = base + l16;
	u32 i16 = 0, i8 = 0;
	if (tc >= store_from && tc < ncols) {
		const u32 v16 = lds_ld32(lds, out16 + 4u * ((tc + 15) & 63)), v8 = lds_ld32(lds, out8 + 4u * ((tc + G::TAP) & 63));
		if (FORM == 3) {   /* parked in the frame of the step that finished them: column tc at step tc + 15 (lane 15) / tc + TAP (lane TAP) */
			i16 = v16 - pk_dup(fr_phi(tc + 15, 15, 16, fr_base, fr_kmask, gapE));
			i8 = v8 - pk_dup(fr_phi(tc + G::TAP, G::TAP, 16, fr_base, fr_kmask, gapE));
		} else { i16 = v16; i8 = v8; }
		o16[tc] = i16;
		o8[tc] = i8;
	}
	if (g16) {
		u32 m16 = i16, m8 = i8;
		m16 = pk_max(m16, xl_row_ror<1>(m16)); m8 = pk_max(m8, xl_row_ror<1>(m8));
		m16 = pk_max(m16, xl_row_ror<2>(m16)); m8 = pk_max(m8, xl_row_ror<2>(m8));
		m16 = pk_max(m16, xl_row_ror<4>(m16)); m8 = pk_max(m8, xl_row_ror<4>(m8));
		m16 = pk_max(m16, xl_row_ror<8>(m16)); m8 = pk_max(m8, xl_row_ror<8>(m8));
		if (l16 == 0 && base >= store_from && base < ncols) { g16[base >> 4] = m16; g8[base >> 4] = m8; }
	}
}

/* ================================================================================================
 * k_fill: forward fill, column maxima only.  grid = npairs * bpp workgroups of 256 threads.
 * ================================================================================================ */
/* FORM 0: plain int16, 9 instructions per row; 3: column frame, 6.5 (chain_rows) */
/* (up to 10 rows per lane the kernel is held to 72 registers -- seven wavefronts per SIMD; hipcc otherwise takes 74, i.e. 80, and
   the one spill this costs is a pointer reloaded once per 16 steps) */
template <int R, int FORM>
SSW_DEV void fill_body(const ssw_fill_args& a, const int bid, unsigned char* lds)
{
	constexpr bool FR = FORM == 3;
	typedef ChainGeom<R> G;
	constexpr int C = G::C;
	const int tid = (int)threadIdx.x, l16 = tid & 15, grp = tid >> 4;
	const int pair = bid / a.bpp, tchunk = bid - pair * a.bpp;
	const int gapEi = (int)(a.gapE2 & 0xffffu);
	const u32 prof_bytes = (u32)(a.n + 1) * G::PSTRIDE;
	const u32 ring = prof_bytes + (u32)grp * CHAIN_BYTES, out16 = ring + RING_BYTES, out8 = out16 + 256;
	const u32 nulloff = (u32)a.n * G::PSTRIDE;

	{   /* score profile of this pair, shared by the 16 chains of the workgroup */
		const ssw_pair pr = a.pairs[pair];
		const int8_t* qa = a.qcodes + a.qoff[pr.qa];
		const int lena = (int)(a.qoff[pr.qa + 1] - a.qoff[pr.qa]);
		const int8_t* qb = pr.qb >= 0 ? a.qcodes + a.qoff[pr.qb] : (const int8_t*)0;
		const int lenb = pr.qb >= 0 ? (int)(a.qoff[pr.qb + 1] - a.qoff[pr.qb]) : 0;
		build_profile<R, FR ? 2 : 0>(lds, 0, tid, 256, a.mat, a.n, qa, lena, 0, qb, lenb, 0x7fffffff, 0x7fffffff, gapEi);
	}

	/* this chain's tile */
	const int t = tchunk * 16 + grp;
	const bool active = t < a.ntiles;
	const int tile_lo = active ? t * a.tile : 0;
	const int tile_hi = active ? (tile_lo + a.tile < a.refLen ? tile_lo + a.tile : a.refLen) : 0;
	const int c_first = tile_lo - a.halo > 0 ? tile_lo - a.halo : 0;
	const int ncols = tile_hi - c_first;
	int maxcols = a.tile + a.halo; if (maxcols > a.refLen) maxcols = a.refLen;
	const int nsteps = (maxcols + 16 + 15) & ~15;
	const int8_t* tg = a.tgt + c_first;
	uint32_t* o16 = a.cm16 + (int64_t)pair * a.cm_stride + c_first;
	uint32_t* o8 = a.cm8 + (int64_t)pair * a.cm_stride + c_first;
	const int store_from = tile_lo - c_first;   /* first traversal column whose maximum is kept */
	/* maxima of every aligned group of 16 columns (tiles and halos are multiples of 16 columns, so a group belongs to one tile):
	   k_reduce_seg scans these instead of the columns themselves */
	uint32_t* g16 = a.sg16 ? a.sg16 + (int64_t)pair * a.seg_stride + (c_first >> 4) : (uint32_t*)0;
	uint32_t* g8 = a.sg16 ? a.sg8 + (int64_t)pair * a.seg_stride + (c_first >> 4) : (uint32_t*)0;

	/* target ring: columns -16..-1 are "null" columns, 0..15 loaded now, 16..31 in flight */
	lds_st16(lds, ring + 2u * (48 + l16), nulloff);
	{
		int code = l16 < ncols ? tg[l16] : a.n;
		if (code < 0 || code > a.n) code = a.n;
		const u32 off = (u32)code * G::PSTRIDE;
		lds_st16(lds, ring + 2u * l16, off);
		lds_st16(lds, ring + 2u * (64 + l16), off);
	}
	u32 nxt;
	{
		const int tc = 16 + l16;
		int code = tc < ncols ? tg[tc] : a.n;
		if (code < 0 || code > a.n) code = a.n;
		nxt = (u32)code * G::PSTRIDE;
	}
	__syncthreads();

	/* frame form: the all-zero state of the column before the lane's first one is phi of that column; `fl` runs one column ahead */
	const u32 zero0 = FR ? pk_dup(fr_phi(0, l16, 16, a.fr_base, a.fr_kmask, gapEi) - gapEi) : 0u;
	u32 fl = zero0 + a.gapE2;
	u32 H[R], E[R];
#pragma unroll
	for (int r = 0; r < R; ++r) { H[r] = zero0; E[r] = FR ? fl : 0u; }
	u32 Hlast = zero0, Fout = FR ? zero0 + a.gapE2 : zero0, cmout = zero0, ck = 0, hsave = zero0;      /* (frame form: Fout is OPEN, one gapE above what the next lane takes) */
	const u32 lane_prof = (u32)l16 * 16u;
	const u32 gO = FR ? a.gapO2 - a.gapE2 : a.gapO2;      /* frame form: gapO - gapE */
	const u32 gE = a.gapE2;
	const u32 gEv = opaque(gE);                             /* in a vector register: the second operand of a DPP instruction cannot be scalar */
	u32 fin = 0;                                            /* frame form: the F a lane takes over; lane 0 keeps this zero (below every frame value) */

	for (int s0 = 0; s0 < nsteps; s0 += 16) {
		if (FR && s0 > 0 && (s0 & a.fr_kmask) == 0) {   /* renormalisation: every frame value drops by K x gapE */
			const u32 k = (u32)(a.fr_kmask + 1) * a.gapE2;
#pragma unroll
			for (int r = 0; r < R; ++r) { H[r] -= k; E[r] -= k; }
			Hlast -= k; Fout -= k; cmout -= k; hsave -= k; fl -= k;
		}
		{   /* stage target columns [s0+16, s0+32), prefetch [s0+32, s0+48) */
			const int p = (s0 + 16 + l16) & 63;
			lds_st16(lds, ring + 2u * p, nxt);
			if (p < 16) lds_st16(lds, ring + 2u * (64 + p), nxt);
			const int tc = s0 + 32 + l16;
#ifdef FILL_STAGE_FREE      /* MEASUREMENT ONLY (wrong results): no target load, no range check -- the upper bound of what any cheaper target staging
                               (2-bit / 4-bit packed targets) could gain; see DESIGN.md "packed targets: closed by measurement" */
			nxt = (u32)((tc >> 2) & 3) * G::PSTRIDE;
#else
			int code = tc < ncols ? tg[tc] : a.n;
			if (code < 0 || code > a.n) code = a.n;
			nxt = (u32)code * G::PSTRIDE;
#endif
		}
		wave_lds_fence();   /* lane 0's ring writes of the previous 16 steps are visible to the chain */
		if (s0 >= 32) fill_flush16<R, FORM>(lds, out16, out8, s0 - 32, l16, store_from, ncols, o16, o8, g16, g8, a.fr_base, a.fr_kmask, gapEi);   /* columns [s0-32, s0-16) are complete in the out rings */
		wave_lds_fence();
		const u32 rp = ring + 2u * (u32)((s0 - l16) & 63);
		/* the lanes that FINISH a maximum park it themselves, in the slot of the step: lane 15 the column's (all rows; its column is
		   s - 15), lane TAP the one of the rows < A8 (its column is s - TAP) -- nothing travels back to lane 0, whose column starts
		   from the zero the row_shr move fills in */
		const u32 ob16 = out16 + 4u * (u32)(s0 & 63), ob8 = out8 + 4u * (u32)(s0 & 63);
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			const u32 paddr = lds_ld16(lds, rp + 2u * j) + lane_prof;
			u32x4 sc[C];
#pragma unroll
			for (int c = 0; c < C; ++c) sc[c] = lds_ld128(lds, paddr + 256u * c);
			u32 hin, f;
			if (FR) {   /* two hand-offs fused with the arithmetic behind them (lanes.h) */
				hin = xl_row_shr1_umax(Hlast, fl); fl += gE;      /* lane 0: the zero that row_shr fills in becomes phi(column) -- a plain u32 max is exact here, both halves of every H are >= phi */
				xl_row_shr1_sub_keep(fin, Fout, gEv);              /* the last row's "- gapE" happens here (Fout is OPEN); lane 0 keeps its zero */
				f = fin;
			} else { hin = xl_row_shr1_zero(Hlast); f = xl_row_shr1_zero(Fout); }
			u32 cm = xl_row_shr1_zero(cmout);      /* this column's maximum of the rows above */
			chain_rows<R, true, FORM>(sc, H, E, hsave, f, cm, ck, gO, gE, fl);
			hsave = hin; Hlast = H[R - 1]; Fout = f; cmout = cm;
			if (G::TAP == 15) {
				if (l16 == 15) { lds_st32(lds, ob16 + 4u * j, cm); lds_st32(lds, ob8 + 4u * j, ck); }
			} else {
				if (l16 == 15) lds_st32(lds, ob16 + 4u * j, cm);
				if (l16 == G::TAP) lds_st32(lds, ob8 + 4u * j, ck);
			}
		}
	}
	wave_lds_fence();
	for (int base = nsteps - 32; base < nsteps; base += 16)
		if (base >= 0) fill_flush16<R, FORM>(lds, out16, out8, base, l16, store_from, ncols, o16, o8, g16, g8, a.fr_base, a.fr_kmask, gapEi);
}

template <int R, int FORM>
__global__ void __launch_bounds__(256) SSW_WAVES_PER_EU(R <= 10 ? 7 : 1, 8) k_fill(ssw_fill_args a)
{
	SSW_DYN_LDS(lds);
	fill_body<R, FORM>(a, (int)blockIdx.x, lds);
}

/* k_fillm: SEVERAL geometry buckets in one launch (round 4).  A batch of mixed read lengths -- the reference's own benchmark: 1000 reads of
   25-540 bp -- is ~25 buckets with a few pairs each; as 25 launches they queue up behind each other on the four hardware queues and the
   device ends with a long thin tail of small kernels (profiles/round4_config6_timeline.txt).  Here the workgroups of all buckets of one
   register class (rows per lane 1-10 / 11-16 / 17-24: 72 / 96 / 128 registers) form one grid -- the sub-launch of a workgroup is found from
   a table, the body is the same fill_body<R> -- longest chains first, and the hardware dispatches them as slots free up. */
template <int CLS> struct FillClass { static constexpr int LO = CLS == 0 ? 1 : CLS == 1 ? 11 : 17, HI = CLS == 0 ? 10 : CLS == 1 ? 16 : 24; };
template <int R, int CLS, int FORM> SSW_DEV void fillm_case(const ssw_fill_args& a, int bid, unsigned char* lds)
{
	if (R >= FillClass<CLS>::LO && R <= FillClass<CLS>::HI) fill_body<(R >= FillClass<CLS>::LO && R <= FillClass<CLS>::HI) ? R : FillClass<CLS>::LO, FORM>(a, bid, lds);
}
template <int CLS, int FORM>
__global__ void __launch_bounds__(256) SSW_WAVES_PER_EU(CLS == 0 ? 7 : 1, 8) k_fillm(ssw_fillm_args m)
{
	SSW_DYN_LDS(lds);
	int s = 0;
	while (s + 1 < m.nsub && (int)blockIdx.x >= m.first_wg[s + 1]) ++s;      /* (uniform: scalar loads) */
	const ssw_fill_args a = m.sub[s];
	const int bid = (int)blockIdx.x - m.first_wg[s];
	switch (m.subR[s]) {
#define X(r) case r: fillm_case<r, CLS, FORM>(a, bid, lds); break;
		X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
#undef X
		default: break;
	}
}

/* ================================================================================================
 * k_filldb: database search.  Same chains as k_fill, but the 16 chains of a workgroup take 16 different (short,
 * untiled) targets, each lane also remembers where the running column maximum last grew, and the chain reduces its
 * own column maxima afterwards: score1 / ref_end1 / read_end1 / score2 / ref_end2 come out of ONE launch.
 * grid = npairs * ceil(ntl / NCH) workgroups of NCH chains (16 * NCH threads: the profile of a pair -- 25 residues x R rows x
 * 64 bytes for proteins -- is what limits the wavefronts per CU, and more chains per workgroup share it).
 *
 * a.form != 0: the column-frame form of the recurrence (chain_rows_fr), else plain int16 with the two-row column maximum.
 * ================================================================================================ */
#define DB_OUT_RING 32                                   /* finished column maxima parked per chain (k_fill keeps 64) */
#define DB_RING_BYTES 192                                /* target ring: 64 entries + 32 mirrored (entries are read two steps ahead) */
#define DB_CHAIN_BYTES (DB_RING_BYTES + 8 * DB_OUT_RING)

/* the packed scores of N <= 4 consecutive rows of a lane (one 16-byte profile chunk; only the rows that exist are loaded) */
template <int N> SSW_DEV u32x4 lds_ld_rows(const unsigned char* lds, u32 off)
{
	if (N >= 3) return lds_ld128(lds, off);
	u32x4 r = { 0u, 0u, 0u, 0u };
	if (N == 2) { const u32x2 t = lds_ld64(lds, off); r[0] = t[0]; r[1] = t[1]; }
	else r[0] = lds_ld32(lds, off);
	return r;
}

/* k_filldb: the R rows of one step (same arithmetic and pairing as chain_rows with TRACK8), with the score chunks of the NEXT step
   requested as soon as the rows of a chunk are done: one set of score registers instead of two, and every profile read still has
   a whole step to land.  db_done(r) = the last row that is finished after iteration r of the unrolled loop. */
template <int R> constexpr int db_done(int r) { return r < 0 ? -1 : r; }

/* FR: column-frame form (chain_rows_fr; gO then carries gapO - gapE and fl = phi one column ahead), else plain int16 with the two-row maximum */
template <int R, bool FR>
SSW_DEV void db_rows(const unsigned char* lds, u32 pa_next, u32x4 (&sc)[(R + 3) / 4], u32 (&H)[R], u32 (&E)[R], u32 d, u32& f, u32& cm, u32& ck,
                     u32 gO, u32 gE, u32 fl)
{
	constexpr int C = (R + 3) / 4, K8 = ChainGeom<R>::K8;
	u32 pa = pa_next;
#pragma unroll
	for (int r = 0; r < R; ++r) {
		const int seg0 = r < K8 ? 0 : K8, seg1 = r < K8 ? K8 : R;
		const bool second = ((r - seg0) & 1) == 1;      /* second row of a pair */
		const bool pair = !second && r + 1 < seg1;      /* first row of a pair (otherwise a single row left over) */
		if (r == K8) ck = cm;                           /* rows < A8 end here in lane TAP: the 16-bit-rule maximum */
		{
			const u32 hold = H[r];
			u32 h;
			if (FR) {
				h = pk_max3_fr(d + sc[r >> 2][r & 3], E[r], f);
				const u32 t = h - gO;
				E[r] = pk_max3_fr(E[r], t, fl);
				f = pk_max(f, t);
				if (r != R - 1) f -= gE;      /* the last row leaves f OPEN: the lane below subtracts while it takes it over (k_fill) */
			} else {
				const u32 h0 = pk_max(pk_adds(d, sc[r >> 2][r & 3]), E[r]);
				h = pk_max(h0, f);
				const u32 t0 = pk_subu(h0, gO);
				E[r] = pk_max(pk_subu(E[r], gE), t0);
				f = pk_max(pk_subu(f, gE), t0);
			}
			if (second) cm = pk_max3_nonneg(cm, H[r - 1 >= 0 ? r - 1 : 0], h);   /* H[r-1] was just written: this pair's first row */
			else if (!pair) cm = pk_max(cm, h);
			H[r] = h;
			d = hold;
		}
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const int last = 4 * c + 3 < R - 1 ? 4 * c + 3 : R - 1;      /* the chunk's last row */
			if (last <= db_done<R>(r) && last > db_done<R>(r - 1)) {
				pa = after(pa, H[last]);      /* not before the chunk's rows have used the old scores.  (CHAINED through one variable: `after`
				                                 on the untouched pa_next cost a register copy per chunk -- the original stayed live -- i.e. four
				                                 vector instructions per step of a 20-row lane) */
				if (c + 1 < C) sc[c] = lds_ld128(lds, pa + 256u * c);
				else sc[c] = lds_ld_rows<R - 4 * (C - 1)>(lds, pa + 256u * c);
			}
		}
	}
}

/* Best-cell tracking of k_filldb.  `now` = the lane's running record after a step, `pre` = before it (with the column's rows above
   folded in): the lane whose OWN rows raised a half has a new candidate for that query's best cell and keeps the column and that
   half of its H column.  On unrelated proteins some lane of the wavefront (4 chains x 2 queries) sets one in 60 % of the steps (every
   new maximum of the rows above and the columns before counts, and ties with the chain's best do), so the record itself is
   branch-free: a per-half byte mask of the halves that rose, one v_bfi per row for both queries at once -- no compare / select
   chains, no changes of the execution mask (R / 2 byte permutes per half behind two more branches cost 23 % of the kernel, a
   look-up of the row at once or stores of the column to scratch more).  The value and the row are read off the kept column
   after the last step. */
template <int R>
SSW_DEV void db_record(u32 now, u32 pre, int tc, const u32 (&H)[R], u32 (&snap)[R], u32& btc2)
{
	const u32 m = pk_gt_mask(now, pre);      /* 0xffff in the halves that rose */
	btc2 = bfi32(m, (u32)tc * 0x00010001u, btc2);
#pragma unroll
	for (int r = 0; r < R; ++r) snap[r] = bfi32(m, H[r], snap[r]);
}

template <int R, int NCH, bool FR, int UNROLL>
SSW_DEV void filldb_pass(const ssw_filldb_args& a, unsigned char* lds)
{
	typedef ChainGeom<R> G;
	constexpr int C = G::C;
	constexpr u32 OM = DB_OUT_RING - 1;
	const int tid = (int)threadIdx.x, l16 = tid & 15, grp = tid >> 4;
	const int tchunks = (a.ntl + NCH - 1) / NCH;
	const int pair = (int)blockIdx.x / tchunks, tchunk = (int)blockIdx.x - pair * tchunks;
	const u32 prof_bytes = (u32)(a.n + 1) * G::PSTRIDE;
	const u32 ring = prof_bytes + (u32)grp * DB_CHAIN_BYTES, out16 = ring + DB_RING_BYTES, out8 = out16 + 4u * DB_OUT_RING;
	const u32 nulloff = (u32)a.n * G::PSTRIDE;
	const ssw_pair pr = a.pairs[pair];
	const int lena = (int)(a.qoff[pr.qa + 1] - a.qoff[pr.qa]);
	const int lenb = pr.qb >= 0 ? (int)(a.qoff[pr.qb + 1] - a.qoff[pr.qb]) : 0;
	const int gapEi = (int)(a.gapE2 & 0xffffu);
#ifndef DB_MEASURE_NO_PROLOGUE      /* MEASUREMENT ONLY (wrong results): what the per-workgroup profile build costs (scripts/build_variants.sh) */
	build_profile<R, FR ? 2 : 0>(lds, 0, tid, 16 * NCH, a.mat, a.n, a.qcodes + a.qoff[pr.qa], lena, 0,
	                             pr.qb >= 0 ? a.qcodes + a.qoff[pr.qb] : (const int8_t*)0, lenb, 0x7fffffff, 0x7fffffff, gapEi);
#endif

	const int slot = tchunk * NCH + grp;
	const bool active = slot < a.ntl;
	const int t = active ? a.tlist[slot] : 0;
	const int8_t* tg = a.tcodes + a.toff[t];
	const int ncols = active ? (int)(a.toff[t + 1] - a.toff[t]) : 0;
	uint32_t* o16 = a.cm16 + ((int64_t)pair * a.ntl + (active ? slot : 0)) * a.cm_stride;
	uint32_t* o8 = a.cm8 + ((int64_t)pair * a.ntl + (active ? slot : 0)) * a.cm_stride;
	/* uniform step count of the workgroup: the longest of its NCH targets (lists are sorted by length) */
	int maxcols = 0;
	{
		const int last = tchunk * NCH + NCH - 1 < a.ntl ? tchunk * NCH + NCH - 1 : a.ntl - 1;
		for (int k = tchunk * NCH; k <= last; ++k) { const int tt = a.tlist[k]; const int L = (int)(a.toff[tt + 1] - a.toff[tt]); maxcols = L > maxcols ? L : maxcols; }
	}
	const int nsteps = (maxcols + 16 + 15) & ~15;
	const int last_step = maxcols + 15;      /* steps [0, last_step) finish every column of the longest target in every lane */

	lds_st16(lds, ring + 2u * (48 + l16), nulloff);
	{
		int code = l16 < ncols ? tg[l16] : a.n;
		if (code < 0 || code > a.n) code = a.n;
		const u32 off = (u32)code * G::PSTRIDE;
		lds_st16(lds, ring + 2u * l16, off);
		lds_st16(lds, ring + 2u * (64 + l16), off);
	}
	u32 nxt;
	{
		const int tc = 16 + l16;
		int code = tc < ncols ? tg[tc] : a.n;
		if (code < 0 || code > a.n) code = a.n;
		nxt = (u32)code * G::PSTRIDE;
	}
	__syncthreads();

	/* frame form: the all-zero state of the column before the lane's first one is phi of that column; `fl` runs one column ahead.  H, the
	   column maxima and `best` carry the phi of the lane's current column (`best` is moved along with it); the column maxima and the
	   best cell become true values where they leave the chain */
	const u32 zero0 = FR ? pk_dup(fr_phi(0, l16, 16, a.fr_base, a.fr_kmask, gapEi) - gapEi) : 0u;
	u32 fl = zero0 + a.gapE2;
	u32 H[R], E[R], snap[R];                        /* snap: per query half, the lane's H column at its last record (db_record) */
#pragma unroll
	for (int r = 0; r < R; ++r) { H[r] = zero0; E[r] = FR ? fl : 0u; snap[r] = 0; }
	u32 Hlast = zero0, Fout = FR ? zero0 + a.gapE2 : zero0, cmout = zero0, ck = 0, hsave = zero0;      /* (frame form: Fout is OPEN) */
	u32 best = zero0;                               /* packed: highest running column maximum this lane has seen */
	u32 btc2 = 0xffffffffu;                         /* packed: column of the last record per half (the host keeps targets below 65000 residues here) */
	const u32 lane_prof = (u32)l16 * 16u;
	const u32 gO = FR ? a.gapO2 - a.gapE2 : a.gapO2;
	const u32 gE = a.gapE2;
	const u32 gEv = opaque(gE);
	u32 fin = 0;                                    /* frame form: the F a lane takes over; lane 0 keeps this zero */

	/* software pipeline of the LDS reads (three or four wavefronts per SIMD do not hide a ring entry -> address -> scores round
	   trip per step): the ring entry of step s+2 is requested before the rows of step s run, the score chunks of step s+1 while
	   they run (db_rows).  Ring entries are staged a chunk ahead and the mirror is 32 entries deep, so reading 17 entries past
	   a chunk's first one is safe. */
	u32x4 sc[C];
	u32 pa_n;
	{
		const u32 r0 = ring + 2u * (u32)((0 - l16) & 63);
		const u32 pa0 = lds_ld16(lds, r0) + lane_prof;
#pragma unroll
		for (int c = 0; c + 1 < C; ++c) sc[c] = lds_ld128(lds, pa0 + 256u * c);
		sc[C - 1] = lds_ld_rows<R - 4 * (C - 1)>(lds, pa0 + 256u * (C - 1));
		pa_n = lds_ld16(lds, r0 + 2u) + lane_prof;
	}

	for (int s0 = 0; s0 < nsteps; s0 += 16) {
		if (FR && s0 > 0 && (s0 & a.fr_kmask) == 0) {   /* renormalisation: every frame value drops by K x gapE (snap keeps the frame of its record) */
			const u32 k = (u32)(a.fr_kmask + 1) * a.gapE2;
#pragma unroll
			for (int r = 0; r < R; ++r) { H[r] -= k; E[r] -= k; }
			Hlast -= k; Fout -= k; cmout -= k; hsave -= k; fl -= k; best -= k;
		}
		{
			const int p = (s0 + 16 + l16) & 63;
			lds_st16(lds, ring + 2u * p, nxt);
			if (p < 32) lds_st16(lds, ring + 2u * (64 + p), nxt);
			const int tc = s0 + 32 + l16;
			int code = tc < ncols ? tg[tc] : a.n;
			if (code < 0 || code > a.n) code = a.n;
			nxt = (u32)code * G::PSTRIDE;
		}
		wave_lds_fence();
		if (s0 >= 32) {
			const int tc = s0 - 32 + l16;
			if (tc < ncols) {
				const u32 v16 = lds_ld32(lds, out16 + 4u * ((u32)(tc + 15) & OM)), v8 = lds_ld32(lds, out8 + 4u * ((u32)(tc + G::TAP) & OM));
				o16[tc] = FR ? v16 - pk_dup(fr_phi(tc + 15, 15, 16, a.fr_base, a.fr_kmask, gapEi)) : v16;
				o8[tc] = FR ? v8 - pk_dup(fr_phi(tc + G::TAP, G::TAP, 16, a.fr_base, a.fr_kmask, gapEi)) : v8;
			}
		}
		wave_lds_fence();
		{   /* once per 16 steps the lanes of a chain learn the chain's best so far: a lane's own creeping maximum below it is no
		       candidate for the best cell, so it need not be recorded (>= the chain's best still is: the first column wins ties,
		       and lanes higher up are ahead in columns) -- records per target drop from ~100 to the handful of true improvements */
			if (a.chain_best) {
				const u32 ph = FR ? fl - a.gapE2 : 0u;      /* the lanes stand in different columns: compare true values (best >= phi of the lane's column) */
				u32 g = best - ph;
				g = pk_max(g, xl_row_ror<1>(g)); g = pk_max(g, xl_row_ror<2>(g)); g = pk_max(g, xl_row_ror<4>(g)); g = pk_max(g, xl_row_ror<8>(g));
				best = pk_max(best, pk_subu(g, 0x00010001u) + ph);
			}
		}
		const u32 rp = ring + 2u * (u32)((s0 - l16) & 63);
		const u32 ob16 = out16 + 4u * ((u32)s0 & OM), ob8 = out8 + 4u * ((u32)s0 & OM);      /* lanes 15 / TAP park what they finish (k_fill) */
		/* the last column of the workgroup's longest target leaves lane 15 at step maxcols + 14: the steps up to the next multiple of 16
		   (8 of ~330 on config 5's proteins) would only push dead columns through the chains */
		const int jend = last_step - s0 < 16 ? (last_step - s0 + UNROLL - 1) / UNROLL * UNROLL : 16;      /* (whole groups of UNROLL steps: the unrolled body has no exit in its middle) */
		for (int j0 = 0; j0 < jend; j0 += UNROLL)
#pragma unroll
		for (int ju = 0; ju < UNROLL; ++ju) {
			const int j = j0 + ju;
			const int tc = s0 + j - l16;
			const u32 pa_next = lds_ld16(lds, rp + 2u * (j + 2));      /* the ring entry of step s + 2 */
			u32 hin, f;
			if (FR) {   /* two hand-offs fused with the arithmetic behind them (k_fill) */
				hin = xl_row_shr1_umax(Hlast, fl); fl += gE; best += gE;      /* lane 0: the zero row_shr fills in becomes phi(column); the record follows the lane's frame */
				xl_row_shr1_sub_keep(fin, Fout, gEv);
				f = fin;
			} else { hin = xl_row_shr1_zero(Hlast); f = xl_row_shr1_zero(Fout); }
			u32 cm = xl_row_shr1_zero(cmout);      /* this column's maximum of the rows above (lane 0 starts a new column with 0) */
			/* best cell: `pre` = the lane's running record and the rows above in this column; only the lane whose OWN rows
			   beat it -- the lane holding the new record cell, not every lane below -- takes the branch further down */
			const u32 pre = pk_max(best, cm);
			db_rows<R, FR>(lds, pa_n, sc, H, E, hsave, f, cm, ck, gO, gE, fl);   /* int16 form: the host keeps max(mat) x 640 below 31744 on this path */
			pa_n = pa_next + lane_prof;
			hsave = hin; Hlast = H[R - 1]; Fout = f; cmout = cm;
			best = pk_max(best, cm);     /* = max(pre, own rows) */
			if (G::TAP == 15) {
				if (l16 == 15) { lds_st32(lds, ob16 + 4u * j, cm); lds_st32(lds, ob8 + 4u * j, ck); }
			} else {
				if (l16 == 15) lds_st32(lds, ob16 + 4u * j, cm);
				if (l16 == G::TAP) lds_st32(lds, ob8 + 4u * j, ck);
			}
			/* (columns outside the target score "dead": H = max(E, F) there, which decays and never sets a record) */
			if (wave_any(best != pre)) db_record<R>(best, pre, tc, H, snap, btc2);   /* a scalar branch: hipcc otherwise if-converts half of the row search into every step */
		}
	}
	/* the lane's last record per half: the value is the largest of the kept column (its own rows had just raised the maximum) */
	int bval[2], btc[2], brow[2];
	{
		u32 mx = 0;
#pragma unroll
		for (int r = 0; r < R; ++r) mx = pk_max(mx, snap[r]);
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			bval[h] = (int)((mx >> (16 * h)) & 0xffffu);
			btc[h] = bval[h] > 0 ? (int)((btc2 >> (16 * h)) & 0xffffu) : 0x7fffffff;
			brow[h] = 0x7fffffff;
#pragma unroll
			for (int k = R - 1; k >= 0; --k) if (bval[h] > 0 && (int)((snap[k] >> (16 * h)) & 0xffffu) == bval[h]) brow[h] = l16 * R + k;
			if (FR && bval[h] > 0) bval[h] -= fr_phi(btc[h] + l16, l16, 16, a.fr_base, a.fr_kmask, gapEi);      /* the kept column is in the frame of its step */
		}
	}
	wave_lds_fence();
	for (int base = nsteps - 32; base < nsteps; base += 16) {
		const int tc = base + l16;
		if (tc >= 0 && tc < ncols) {
			const u32 v16 = lds_ld32(lds, out16 + 4u * ((u32)(tc + 15) & OM)), v8 = lds_ld32(lds, out8 + 4u * ((u32)(tc + G::TAP) & OM));
			o16[tc] = FR ? v16 - pk_dup(fr_phi(tc + 15, 15, 16, a.fr_base, a.fr_kmask, gapEi)) : v16;
			o8[tc] = FR ? v8 - pk_dup(fr_phi(tc + G::TAP, G::TAP, 16, a.fr_base, a.fr_kmask, gapEi)) : v8;
		}
	}
	dev_fence();   /* the chain re-reads its own column maxima below */

	/* ---- per chain, per query: reduce (the role of k_reduce + the locate pass) ---- */
	const u32 red = ring;   /* the rings are free now: 16 lanes x 16 bytes */
	for (int h = 0; h < 2; ++h) {
		const int q = h ? pr.qb : pr.qa;
		if (q < 0) continue;                         /* uniform in the workgroup */
		const int len = h ? lenb : lena;
		lds_st32(lds, red + 16u * l16, (u32)bval[h]);
		lds_st32(lds, red + 16u * l16 + 4, (u32)btc[h]);
		lds_st32(lds, red + 16u * l16 + 8, (u32)brow[h]);
		wave_lds_fence();
		int bv = 0, bc = 0x7fffffff, br = 0x7fffffff;
		for (int k = 0; k < 16; ++k) {               /* every lane computes the same winner: value, then column, then lane order */
			const int v = (int)lds_ld32(lds, red + 16u * k), cc = (int)lds_ld32(lds, red + 16u * k + 4), w = (int)lds_ld32(lds, red + 16u * k + 8);
			if (v > bv || (v == bv && v > 0 && cc < bc)) { bv = v; bc = cc; br = w; }
		}
		wave_lds_fence();
		const bool padded = (len & 15) >= 1 && (len & 15) <= 8;
		const int maskLen = a.maskLen >= 0 ? a.maskLen : len / 2;
		const bool have_byte = a.score_size == 0 || a.score_size == 2, have_word = a.score_size == 1 || a.score_size == 2;
		int word = 0, status = 0;
		if (have_byte && bv < 255 - a.bias) word = 0;
		else if (have_word) word = 1;
		else status = 1;
		int s2 = 0, i2 = 0x7fffffff;
#ifndef DB_MEASURE_NO_SCAN          /* MEASUREMENT ONLY (wrong score2): what the fused second-best scan over the chain's own column maxima costs */
		if (status == 0 && bv > 0) {
			const uint32_t* arr = (word && padded) ? o8 : o16;
			const int lo_edge = bc - maskLen > 0 ? bc - maskLen : 0;
			const int hi_edge = bc + maskLen > ncols ? ncols : bc + maskLen;
			const int up_from = word ? hi_edge : hi_edge + 1;
			for (int c = l16; c < ncols; c += 16) {
				if (c < lo_edge || c >= up_from) {
					const int v = (int)((arr[c] >> (16 * h)) & 0xffffu);
					if (v > s2) { s2 = v; i2 = c; }
				}
			}
		}
#endif
		lds_st32(lds, red + 16u * l16, (u32)s2);
		lds_st32(lds, red + 16u * l16 + 4, (u32)i2);
		wave_lds_fence();
		if (l16 == 0 && active) {
			for (int k = 1; k < 16; ++k) {
				const int v = (int)lds_ld32(lds, red + 16u * k), cc = (int)lds_ld32(lds, red + 16u * k + 4);
				if (v > s2 || (v == s2 && cc < i2)) { s2 = v; i2 = cc; }
			}
			ssw_dres r;
			r.score1 = 0; r.score2 = 0; r.ref_begin1 = -1; r.ref_end1 = 0; r.read_begin1 = -1; r.read_end1 = 0;
			r.ref_end2 = 0; r.cigarLen = 0; r.flag = 0; r.status = status; r.word = word; r.want_begin = 0; r.want_cigar = 0;
			r.rev_score = 0; r.loc_done = 0; r.nm = 0; r.cigar_off = 0;
			if (status == 0 && bv > 0) {
				r.score1 = bv; r.ref_end1 = bc; r.read_end1 = br < len - 1 ? br : len - 1;
				if (maskLen >= 15) { r.score2 = s2; r.ref_end2 = s2 > 0 ? i2 : 0; }
				else { r.score2 = 0; r.ref_end2 = -1; }
			}
			if (a.hits) {      /* compact score-only record of the streaming database search (include/ssw_gpu.h ssw_gpu_hit) */
				ssw_hit_rec o;
				o.score1 = (uint16_t)r.score1; o.score2 = (uint16_t)r.score2; o.ref_end1 = r.ref_end1; o.read_end1 = r.read_end1;
				o.ref_end2 = r.status == 1 ? -2 : r.ref_end2;
				a.hits[(int64_t)q * a.res_nt + (t - a.tfirst)] = o;
				if (a.counters && r.status == 0 && r.score1 > 0) atomicAdd(a.counters + (r.word ? 0 : 1), 1);
			} else if (a.out) {
				ssw_out_rec o;
				o.score1 = (uint16_t)r.score1; o.score2 = (uint16_t)r.score2; o.ref_begin1 = -1; o.ref_end1 = r.ref_end1; o.read_begin1 = -1;
				o.read_end1 = r.read_end1; o.ref_end2 = r.ref_end2; o.cigarLen = 0; o.edit_distance = 0; o.cigar_off = -1; o.flag = 0;
				o.status = (uint16_t)(r.status | (a.mark_word && r.word ? SSW_OUT_WORD : 0));
				a.out[(int64_t)q * a.res_nt + (t - a.tfirst)] = o;
				if (a.counters && r.status == 0 && r.score1 > 0) atomicAdd(a.counters + (r.word ? 0 : 1), 1);
			} else a.res[(int64_t)q * a.res_nt + (t - a.tfirst)] = r;
		}
		wave_lds_fence();
	}
}

/* FR: the column-frame form (whenever the bucket's scores and the frame offsets stay below 31744: the host decides), else plain int16 */
#ifndef DB_UNROLL
#define DB_UNROLL 2
#endif
#ifndef DB_WAVES_PER_EU      /* register budget of k_filldb as wavefronts per SIMD (experiments: scripts/build_variants.sh) */
#define DB_WAVES_PER_EU 1
#endif
template <int R, int NCH, bool FR>
__global__ void __launch_bounds__(16 * NCH) SSW_WAVES_PER_EU(DB_WAVES_PER_EU, 8) k_filldb(ssw_filldb_args a)
{
	SSW_DYN_LDS(lds);
	if (FR) filldb_pass<R, NCH, true, DB_UNROLL>(a, lds);
	else filldb_pass<R, NCH, false, 4>(a, lds);
}

/* ================================================================================================
 * k_reduce: one workgroup per pair; both queries of the pair.
 * ================================================================================================ */
SSW_DEV int half16(u32 w, int hi) { return (int)((hi ? (w >> 16) : w) & 0xffffu); }

/* block-wide (max value, then min index) reduction; result valid in every thread */
SSW_DEV void block_argmax(unsigned char* lds, int tid, int& val, int& idx, int nthreads = 256)
{
	lds_st32(lds, 8u * tid, (u32)val);
	lds_st32(lds, 8u * tid + 4, (u32)idx);
	__syncthreads();
	for (int st = nthreads >> 1; st > 0; st >>= 1) {
		if (tid < st) {
			int v0 = (int)lds_ld32(lds, 8u * tid), i0 = (int)lds_ld32(lds, 8u * tid + 4);
			int v1 = (int)lds_ld32(lds, 8u * (tid + st)), i1 = (int)lds_ld32(lds, 8u * (tid + st) + 4);
			if (v1 > v0 || (v1 == v0 && i1 < i0)) { lds_st32(lds, 8u * tid, (u32)v1); lds_st32(lds, 8u * tid + 4, (u32)i1); }
		}
		__syncthreads();
	}
	val = (int)lds_ld32(lds, 0); idx = (int)lds_ld32(lds, 4);
	__syncthreads();
}

__global__ void __launch_bounds__(256) k_reduce(ssw_reduce_args a)
{
	SSW_DYN_LDS(lds);
	const int tid = (int)threadIdx.x, pair = (int)blockIdx.x;
	const ssw_pair pr = a.pairs[pair];
	const u32x4* w16 = (const u32x4*)(a.cm16 + (int64_t)pair * a.cm_stride);   /* cm_stride is a multiple of 16 words */
	const u32x4* w8 = (const u32x4*)(a.cm8 + (int64_t)pair * a.cm_stride);
	const int nvec = (a.refLen + 3) >> 2;

	/* pass 1 (one sweep for both queries): best score and its first column */
	int best[2] = { 0, 0 }, bidx[2] = { 0x7fffffff, 0x7fffffff };
	for (int v = tid; v < nvec; v += 256) {
		const u32x4 w = w16[v];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int c = v * 4 + k;
			if (c < a.refLen) {
				const int lo = (int)(w[k] & 0xffffu), hi = (int)(w[k] >> 16);
				if (lo > best[0]) { best[0] = lo; bidx[0] = c; }
				if (hi > best[1]) { best[1] = hi; bidx[1] = c; }
			}
		}
	}
	block_argmax(lds, tid, best[0], bidx[0]);
	block_argmax(lds, tid, best[1], bidx[1]);

	ssw_dres r[2];
	int use8[2] = { 0, 0 }, lo_edge[2] = { 0, 0 }, up_from[2] = { 0, 0 }, live[2] = { 0, 0 }, mlen[2] = { 0, 0 };
	for (int h = 0; h < 2; ++h) {
		const int q = h ? pr.qb : pr.qa;
		ssw_dres& x = r[h];
		x.score1 = 0; x.score2 = 0; x.ref_begin1 = -1; x.ref_end1 = 0; x.read_begin1 = -1; x.read_end1 = 0;
		x.ref_end2 = 0; x.cigarLen = 0; x.flag = 0; x.status = 0; x.word = 0; x.want_begin = 0; x.want_cigar = 0;
		x.rev_score = 0; x.loc_done = 0; x.nm = 0; x.cigar_off = 0;
		if (q < 0) continue;
		const int len = (int)(a.qoff[q + 1] - a.qoff[q]);
		const bool padded = (len & 15) >= 1 && (len & 15) <= 8;      /* 16-bit rules see 8 rows fewer */
		mlen[h] = a.maskLen >= 0 ? a.maskLen : len / 2;
		const bool have_byte = a.score_size == 0 || a.score_size == 2, have_word = a.score_size == 1 || a.score_size == 2;
		int word = 0;
		if (have_byte && best[h] < 255 - a.bias) word = 0;             /* ssw.c:881-899 */
		else if (have_word) word = 1;
		else x.status = 1;
		x.word = word;
		if (x.status == 0 && best[h] > 0) {
			live[h] = 1;
			use8[h] = word && padded;
			lo_edge[h] = bidx[h] - mlen[h] > 0 ? bidx[h] - mlen[h] : 0;
			const int hi_edge = bidx[h] + mlen[h] > a.refLen ? a.refLen : bidx[h] + mlen[h];
			up_from[h] = word ? hi_edge : hi_edge + 1;                 /* ssw.c:376 vs 578 */
		}
	}

	/* pass 2: masked second best; each column-maximum stream is swept at most once */
	int s2[2] = { 0, 0 }, i2[2] = { 0x7fffffff, 0x7fffffff };
	for (int arr = 0; arr < 2; ++arr) {
		const bool want0 = live[0] && use8[0] == arr, want1 = live[1] && use8[1] == arr;
		if (!want0 && !want1) continue;
		const u32x4* src = arr ? w8 : w16;
		for (int v = tid; v < nvec; v += 256) {
			const u32x4 w = src[v];
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int c = v * 4 + k;
				if (c < a.refLen) {
					const int lo = (int)(w[k] & 0xffffu), hi = (int)(w[k] >> 16);
					if (want0 && (c < lo_edge[0] || c >= up_from[0]) && lo > s2[0]) { s2[0] = lo; i2[0] = c; }
					if (want1 && (c < lo_edge[1] || c >= up_from[1]) && hi > s2[1]) { s2[1] = hi; i2[1] = c; }
				}
			}
		}
	}
	block_argmax(lds, tid, s2[0], i2[0]);
	block_argmax(lds, tid, s2[1], i2[1]);
	if (tid == 0) {
		for (int h = 0; h < 2; ++h) {
			const int q = h ? pr.qb : pr.qa;
			if (q < 0) continue;
			if (live[h]) {
				r[h].score1 = best[h]; r[h].ref_end1 = bidx[h];
				if (mlen[h] >= 15) { r[h].score2 = s2[h]; r[h].ref_end2 = s2[h] > 0 ? i2[h] : 0; }
				else { r[h].score2 = 0; r[h].ref_end2 = -1; }
				r[h].want_begin = !(a.flag == 0 || (a.flag == 2 && best[h] < a.filters));   /* ssw.c:916 */
				if (a.cand) {   /* the fill tracked its best cell: the tile that owns ref_end1 knows the row */
					const int32_t* cd = a.cand + (((int64_t)pair * a.ntiles + bidx[h] / a.tile) * 2 + h) * 4;
					if (cd[0] == best[h] && cd[1] == bidx[h]) {
						const int len = (int)(a.qoff[q + 1] - a.qoff[q]);
						r[h].read_end1 = cd[2] < len - 1 ? cd[2] : len - 1;
						r[h].loc_done = 1;
					}
				}
			}
			a.res[q] = r[h];
		}
	}
}

/* k_reduce_seg: the same reduction over the maxima of 16-column groups that k_fill leaves behind (1/16 of the words); the
   columns themselves are only read in the group that holds the best column and in the (at most two per query) groups that the
   edges of the mask window cut.  Results are identical to k_reduce's: "first column holding the maximum" = first group holding
   it, then first column inside. */
SSW_DEV void seg_first_column(const uint32_t* cols, int seg, int refLen, int hi, int value, int lo_edge, int up_from, bool masked, int& col)
{
	col = 0x7fffffff;
	for (int k = 0; k < 16; ++k) {
		const int c = seg * 16 + k;
		if (c >= refLen) break;
		if (masked && !(c < lo_edge || c >= up_from)) continue;
		if (half16(cols[c], hi) == value) { col = c; break; }
	}
}

SSW_DEV void reduce_seg_body(const ssw_reduce_args& a, const int pair, unsigned char* lds)
{
	const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;      /* 256, or 1024 when a handful of pairs meets a long target (a single ssw_align call: the scan is a chain of memory latencies) */
	const ssw_pair pr = a.pairs[pair];
	const uint32_t* c16 = a.cm16 + (int64_t)pair * a.cm_stride;
	const uint32_t* c8 = a.cm8 + (int64_t)pair * a.cm_stride;
	const uint32_t* g16 = a.sg16 + (int64_t)pair * a.seg_stride;
	const uint32_t* g8 = a.sg8 + (int64_t)pair * a.seg_stride;
	const int nseg = (a.refLen + 15) >> 4;

	/* pass 1: best score and the first group that holds it, both queries in one sweep */
	/* (four groups per 16-byte load, the next load in flight while these are compared: with ONE pair in the launch -- a single ssw_align call
	   against a 1 Mb target is 62 500 groups -- the scan is a chain of memory latencies, 0.3 ms with one word per iteration; the rows of the
	   group arrays are 16-byte aligned and padded: seg_stride is a multiple of 4 and >= nseg + 4) */
	int best[2] = { 0, 0 }, bseg[2] = { 0x7fffffff, 0x7fffffff };
	for (int g0 = tid * 4; g0 < nseg; g0 += 4 * nthr) {
		const u32x4 w4 = *(const u32x4*)(g16 + g0);
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int g = g0 + k;
			if (g < nseg) {
				const int lo = (int)(w4[k] & 0xffffu), hi = (int)(w4[k] >> 16);
				if (lo > best[0]) { best[0] = lo; bseg[0] = g; }
				if (hi > best[1]) { best[1] = hi; bseg[1] = g; }
			}
		}
	}
	block_argmax(lds, tid, best[0], bseg[0], nthr);
	block_argmax(lds, tid, best[1], bseg[1], nthr);
	int bidx[2] = { 0x7fffffff, 0x7fffffff };
	for (int h = 0; h < 2; ++h) if (best[h] > 0) seg_first_column(c16, bseg[h], a.refLen, h, best[h], 0, 0, false, bidx[h]);

	ssw_dres r[2];
	int use8[2] = { 0, 0 }, lo_edge[2] = { 0, 0 }, up_from[2] = { 0, 0 }, live[2] = { 0, 0 }, mlen[2] = { 0, 0 };
	for (int h = 0; h < 2; ++h) {
		const int q = h ? pr.qb : pr.qa;
		ssw_dres& x = r[h];
		x.score1 = 0; x.score2 = 0; x.ref_begin1 = -1; x.ref_end1 = 0; x.read_begin1 = -1; x.read_end1 = 0;
		x.ref_end2 = 0; x.cigarLen = 0; x.flag = 0; x.status = 0; x.word = 0; x.want_begin = 0; x.want_cigar = 0;
		x.rev_score = 0; x.loc_done = 0; x.nm = 0; x.cigar_off = 0;
		if (q < 0) continue;
		const int len = (int)(a.qoff[q + 1] - a.qoff[q]);
		const bool padded = (len & 15) >= 1 && (len & 15) <= 8;      /* 16-bit rules see 8 rows fewer */
		mlen[h] = a.maskLen >= 0 ? a.maskLen : len / 2;
		const bool have_byte = a.score_size == 0 || a.score_size == 2, have_word = a.score_size == 1 || a.score_size == 2;
		int word = 0;
		if (have_byte && best[h] < 255 - a.bias) word = 0;             /* ssw.c:881-899 */
		else if (have_word) word = 1;
		else x.status = 1;
		x.word = word;
		if (x.status == 0 && best[h] > 0) {
			live[h] = 1;
			use8[h] = word && padded;
			lo_edge[h] = bidx[h] - mlen[h] > 0 ? bidx[h] - mlen[h] : 0;
			const int hi_edge = bidx[h] + mlen[h] > a.refLen ? a.refLen : bidx[h] + mlen[h];
			up_from[h] = word ? hi_edge : hi_edge + 1;                 /* ssw.c:376 vs 578 */
		}
	}

	/* pass 2: masked second best.  A group entirely outside the mask window counts with its maximum; a group the window cuts
	   is looked at column by column; ties go to the lowest group, then the lowest column (the reference scans upwards, strict >) */
	int s2[2] = { 0, 0 }, g2[2] = { 0x7fffffff, 0x7fffffff };
	for (int h = 0; h < 2; ++h) {
		if (!live[h]) continue;
		const uint32_t* gs = use8[h] ? g8 : g16;
		const uint32_t* cs = use8[h] ? c8 : c16;
		for (int g0 = tid * 4; g0 < nseg; g0 += 4 * nthr) {
			const u32x4 w4 = *(const u32x4*)(gs + g0);
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int g = g0 + k;
				if (g >= nseg) continue;
				const int c0 = g * 16, c1 = c0 + 16 < a.refLen ? c0 + 16 : a.refLen;      /* columns [c0, c1) */
				int v = 0;
				if (c1 <= lo_edge[h] || c0 >= up_from[h]) v = half16(w4[k], h);              /* wholly allowed */
				else if (c0 >= lo_edge[h] && c1 <= up_from[h]) continue;                      /* wholly masked */
				else for (int c = c0; c < c1; ++c) if (c < lo_edge[h] || c >= up_from[h]) { const int w = half16(cs[c], h); v = w > v ? w : v; }
				if (v > s2[h]) { s2[h] = v; g2[h] = g; }
			}
		}
	}
	block_argmax(lds, tid, s2[0], g2[0], nthr);
	block_argmax(lds, tid, s2[1], g2[1], nthr);
	if (tid == 0) {
		for (int h = 0; h < 2; ++h) {
			const int q = h ? pr.qb : pr.qa;
			if (q < 0) continue;
			if (live[h]) {
				r[h].score1 = best[h]; r[h].ref_end1 = bidx[h];
				if (mlen[h] >= 15) {
					int i2 = 0;
					if (s2[h] > 0) seg_first_column(use8[h] ? c8 : c16, g2[h], a.refLen, h, s2[h], lo_edge[h], up_from[h], true, i2);
					r[h].score2 = s2[h]; r[h].ref_end2 = s2[h] > 0 ? i2 : 0;
				} else { r[h].score2 = 0; r[h].ref_end2 = -1; }
				r[h].want_begin = !(a.flag == 0 || (a.flag == 2 && best[h] < a.filters));   /* ssw.c:916 */
				if (a.cand) {   /* the strip kernel tracked its best cell: the tile that owns ref_end1 knows the row (as k_reduce) */
					const int32_t* cd = a.cand + (((int64_t)pair * a.ntiles + bidx[h] / a.tile) * 2 + h) * 4;
					if (cd[0] == best[h] && cd[1] == bidx[h]) {
						const int len = (int)(a.qoff[q + 1] - a.qoff[q]);
						r[h].read_end1 = cd[2] < len - 1 ? cd[2] : len - 1;
						r[h].loc_done = 1;
					}
				}
			}
			a.res[q] = r[h];
		}
	}
}

__global__ void __launch_bounds__(1024) k_reduce_seg(ssw_reduce_args a)
{
	SSW_DYN_LDS(lds);
	reduce_seg_body(a, (int)blockIdx.x, lds);
}

/* the reductions of several buckets in one grid (one workgroup per pair; a launch per bucket is as slow as ONE pair's scan -- 1.5 ms
   over a 5 Mb target -- however few pairs the bucket has, and 25 of them in a row were a tenth of a mixed-length batch's time) */
__global__ void __launch_bounds__(256) k_reducem(ssw_reducem_args m)
{
	SSW_DYN_LDS(lds);
	int s = 0;
	while (s + 1 < m.nsub && (int)blockIdx.x >= m.first_wg[s + 1]) ++s;
	const ssw_reduce_args a = m.sub[s];
	reduce_seg_body(a, (int)blockIdx.x - m.first_wg[s], lds);
}

/* ================================================================================================
 * k_capture: one chain per alignment; tracks the best cell (value, first column, smallest row) of a
 * window.  reverse == 0: columns [ref_end1 - halo, ref_end1] forward -> read_end1.
 * reverse == 1: reversed read prefix against columns ref_end1, ref_end1-1, ... -> begin position.
 * grid = ceil(nq / 4) workgroups of 64 threads (4 chains, each with its own profile).
 * ================================================================================================ */
template <int R>
__global__ void __launch_bounds__(64) k_capture(ssw_capture_args a)
{
	typedef ChainGeom<R> G;
	constexpr int C = G::C;
	SSW_DYN_LDS(lds);
	const int tid = (int)threadIdx.x, l16 = tid & 15, grp = tid >> 4;
	const u32 prof_bytes = (u32)(a.n + 1) * G::PSTRIDE;
	const u32 prof = (u32)grp * (prof_bytes + CHAIN_BYTES), ring = prof + prof_bytes, red = ring + RING_BYTES;
	const u32 nulloff = (u32)a.n * G::PSTRIDE;
	const int job = (int)blockIdx.x * 4 + grp;
	const int q = job < a.nq ? a.qlist[job] : -1;

	ssw_dres r;
	bool active = false;
	if (q >= 0) { r = a.res[q]; active = r.status == 0 && r.score1 > 0 && (a.reverse ? r.want_begin != 0 : !r.loc_done); }
	int qlen = 0, plen = 0, c_edge = 0, ncols = 0, P = 16;
	const int8_t* qc = a.qcodes;
	if (active) {
		const int qs = vm_query(a.vm, q);
		qc = a.qcodes + a.qoff[qs];
		qlen = (int)(a.qoff[qs + 1] - a.qoff[qs]);
		plen = a.reverse ? r.read_end1 + 1 : qlen;
		P = (plen + 15) & ~15;
		/* exact window for a KNOWN score (round 5).  A path that ends with score S has D <= P diagonal steps (<= max(mat) each) and G target-only
		   gap steps (>= gapE each, gapO > gapE): S <= D max(mat) - G gapE, so it spans D + G <= P + (P max(mat) - S) / gapE columns.  Both passes only
		   ask where score1 is reached: a cell whose true value is score1 has its whole path inside a window of that width, every other cell
		   only gets a lower bound of a value < score1 -- the same cells are found as with the score-free bound P + P max(mat) / gapE, which a
		   clean 150-bp read (score 280 of 300) undercuts 2.4 times (k_capture is 182 of a single ssw_align call's 510 us). */
		const long long lost = (long long)P * (a.maxmat > 0 ? a.maxmat : 0) - r.score1;
		long long w = (long long)P + (lost > 0 ? lost : 0) / (a.gapE > 0 ? a.gapE : 1) + 1;
		if (a.gapE <= 0 || w > r.ref_end1) w = r.ref_end1;
		ncols = (int)w + 1;
		c_edge = a.reverse ? r.ref_end1 : r.ref_end1 - (int)w;   /* traversal column 0 */
	}
	build_profile<R>(lds, prof, l16, 16, a.mat, a.n, qc, plen, a.reverse, (const int8_t*)0, 0);

	/* uniform step count for the 4 chains of the wave */
	int mc = ncols;
#pragma unroll
	for (int sh = 16; sh < 64; sh <<= 1) { const int o = (int)xl_shfl((u32)mc, (tid + sh) & 63); mc = o > mc ? o : mc; }
	const int nsteps = (mc + 16 + 15) & ~15;
	const int8_t* tg = active ? vm_target(a.vm, a.tgt, q) : a.tgt;
	const int dirstep = a.reverse ? -1 : 1;

	lds_st16(lds, ring + 2u * (48 + l16), nulloff);
	{
		int code = l16 < ncols ? tg[c_edge + dirstep * l16] : a.n;
		if (code < 0 || code > a.n) code = a.n;
		const u32 off = (u32)code * G::PSTRIDE;
		lds_st16(lds, ring + 2u * l16, off);
		lds_st16(lds, ring + 2u * (64 + l16), off);
	}
	u32 nxt;
	{
		const int tc = 16 + l16;
		int code = tc < ncols ? tg[c_edge + dirstep * tc] : a.n;
		if (code < 0 || code > a.n) code = a.n;
		nxt = (u32)code * G::PSTRIDE;
	}
	wave_lds_fence();

	u32 H[R], E[R];
#pragma unroll
	for (int k = 0; k < R; ++k) { H[k] = 0; E[k] = 0; }
	u32 Hlast = 0, Fout = 0, hsave = 0;
	int best = 0, btc = 0x7fffffff, brow = 0;
	const u32 lane_prof = prof + (u32)l16 * 16u;

	for (int s0 = 0; s0 < nsteps; s0 += 16) {
		{
			const int p = (s0 + 16 + l16) & 63;
			lds_st16(lds, ring + 2u * p, nxt);
			if (p < 16) lds_st16(lds, ring + 2u * (64 + p), nxt);
			const int tc = s0 + 32 + l16;
			int code = tc < ncols ? tg[c_edge + dirstep * tc] : a.n;
			if (code < 0 || code > a.n) code = a.n;
			nxt = (u32)code * G::PSTRIDE;
		}
		wave_lds_fence();
		const u32 rp = ring + 2u * (u32)((s0 - l16) & 63);
#pragma unroll 2
		for (int j = 0; j < 16; ++j) {
			const int tc = s0 + j - l16;
			const u32 paddr = lds_ld16(lds, rp + 2u * j) + lane_prof;
			u32x4 sc[C];
#pragma unroll
			for (int c = 0; c < C; ++c) sc[c] = lds_ld128(lds, paddr + 256u * c);
			const u32 hin = xl_row_shr1_zero(Hlast);
			u32 f = xl_row_shr1_zero(Fout);
			u32 cm = 0, ck = 0;
			chain_rows<R, false>(sc, H, E, hsave, f, cm, ck, a.gapO2, a.gapE2);
			hsave = hin; Hlast = H[R - 1]; Fout = f;
			const int m = (int)(cm & 0xffffu);     /* this lane's maximum in this column (low half = the query) */
			if (m > best && tc >= 0 && tc < ncols) {
				best = m; btc = tc;
#pragma unroll
				for (int k = R - 1; k >= 0; --k) if ((int)(H[k] & 0xffffu) == m) brow = l16 * R + k;
			}
		}
	}

	/* chain-wide winner: highest value, then first column, then smallest row */
	lds_st32(lds, red + 12u * l16, (u32)best);
	lds_st32(lds, red + 12u * l16 + 4, (u32)btc);
	lds_st32(lds, red + 12u * l16 + 8, (u32)brow);
	wave_lds_fence();
	if (l16 == 0 && active) {
		int bv = 0, bc = 0x7fffffff, br = 0;
		for (int k = 0; k < 16; ++k) {
			const int v = (int)lds_ld32(lds, red + 12u * k), c = (int)lds_ld32(lds, red + 12u * k + 4), w = (int)lds_ld32(lds, red + 12u * k + 8);
			if (v > bv || (v == bv && v > 0 && (c < bc || (c == bc && w < br)))) { bv = v; bc = c; br = w; }
		}
		if (!a.reverse) {
			/* ssw.c:342-351: smallest row holding the maximum, never beyond the last read base */
			a.res[q].read_end1 = (bv == r.score1) ? (br < qlen - 1 ? br : qlen - 1) : -1;
			if (bv != r.score1) a.res[q].status = 3;   /* internal error: window did not reproduce score1 */
		} else {
			if (bv != r.score1 && ncols <= r.ref_end1) {
				a.res[q].status = 3;   /* halo bound violated (cannot happen for gapO > gapE) */
			} else {
				const int rb = r.ref_end1 - bc, qb = r.read_end1 - (br < plen - 1 ? br : plen - 1);
				a.res[q].ref_begin1 = rb; a.res[q].read_begin1 = qb; a.res[q].rev_score = bv;
				if (r.score1 > bv) a.res[q].flag = 2;                       /* ssw.c:932-935 */
				const int skip = (7 & a.flag) == 0 || ((2 & a.flag) != 0 && r.score1 < a.filters) ||
				                 ((4 & a.flag) != 0 && (r.ref_end1 - rb > a.filterd || r.read_end1 - qb > a.filterd));   /* ssw.c:938 */
				a.res[q].want_cigar = !skip;
			}
		}
	}
}

/* ================================================================================================
 * k_chainx: generic chain kernel -- any query length (row strips), one profile per chain.  A chain is GL lanes:
 * GL = 16: one DPP row, 4 chains (jobs) per 64-thread workgroup; GL = 64: the whole wavefront is ONE chain
 * (hand-off with wave_shr:1), strips of 64*R rows -- 4x the waves for the same number of jobs and a quarter of the
 * LDS per wave, which is what long-read batches (few, long queries) need to fill the device.  CAPTURE = false: forward fill of (pair, tile) jobs, column maxima to cm16/cm8.
 * CAPTURE = true: locate / reverse window of one query, best cell to the result record (same contract as k_capture).
 * ================================================================================================ */
#define BND_RING_BYTES (64 * 16)
template <int R, int GL> struct StripGeom {
	static constexpr int C = (R + 3) / 4;
	static constexpr u32 CSTRIDE = (u32)GL * 16u;              /* one 16-byte chunk of every lane */
	static constexpr u32 PSTRIDE = (u32)C * CSTRIDE;           /* one residue */
	static constexpr int RB = GL == 64 ? 128 : 64;             /* target ring entries (> GL + 30), + 32 mirrored */
	static constexpr u32 RINGB = (u32)(RB + 32) * 2u;
	static constexpr u32 EXTRA = 2 * RINGB + 2 * BND_RING_BYTES + (u32)GL * 12u;   /* capture: one target ring per query half */
};

/* a boundary record { H, F, column maximum, 16-bit-rule column maximum }: the fourth word only where it is used -- a register of
   a vector load that nothing reads gets reused by hipcc at once, and overwriting it has to wait for the load to land (a full
   LDS latency per step in a kernel that runs two wavefronts per SIMD) */
template <bool ALL4> SSW_DEV u32x4 lds_ld_bnd(const unsigned char* lds, u32 off)
{
	if (ALL4) return lds_ld128(lds, off);
	const u32x2 hf = lds_ld64(lds, off);
	const u32x4 r = { hf[0], hf[1], lds_ld32(lds, off + 8u), 0u };
	return r;
}

/* profile of one strip: word = (score of query a's row, score of query b's row) against residue b; rows at or below a
   query's padded length are dead for that half */
/* PM 0: packed int16; 2: column frame, packed-sum entries (fr_pack: both halves look at the same target column); 3: column frame,
   per-half two's complement entries (window passes: every query half has its own target column; added with pk_addw) */
template <int R, int GL, int PM = 0>
SSW_DEV void build_profile_strip(unsigned char* lds, u32 base, int first, int nthreads, const int8_t* mat, int n,
                                 const int8_t* qa, int lena, int reva, int rowsa, const int8_t* qb, int lenb, int revb, int rowsb, int row0, int gapE = 0)
{
	constexpr int C = StripGeom<R, GL>::C;
	const int total = (n + 1) * C * GL * 4;
	for (int w = first; w < total; w += nthreads) {
		const int b = w / (C * GL * 4), rem = w - b * (C * GL * 4);
		const int c = rem / (GL * 4), l = (rem & (GL * 4 - 1)) >> 2, k = rem & 3;
		const int r = c * 4 + k, row = row0 + l * R + r;
		u32 v;
		if (r >= R) v = 0;
		else if (b == n) v = PM ? fr_pack(FR_DEAD, FR_DEAD) : DEAD2;                            /* null residue */
		else {
			int lo = -32768, hi = -32768;                      /* rows below the padded query */
			if (row < rowsa) lo = row < lena ? mat[b * n + (reva ? qa[lena - 1 - row] : qa[row])] : 0;
			if (row < rowsb) hi = qb && row < lenb ? mat[b * n + (revb ? qb[lenb - 1 - row] : qb[row])] : 0;
			if (PM == 2) v = fr_pack(lo == -32768 ? FR_DEAD : lo + gapE, hi == -32768 ? FR_DEAD : hi + gapE);
			else if (PM == 3) v = pk_make(lo == -32768 ? FR_DEAD : lo + gapE, hi == -32768 ? FR_DEAD : hi + gapE);
			else v = pk_make(lo, hi);
		}
		lds_st32(lds, base + (u32)w * 4u, v);
	}
}

template <int R> struct ChainState {
	u32 H[R], E[R];
	u32 Hlast, Fout, cmout, cm8out, hsave;
	/* capture mode: best cell of the window per query half (value, first column, smallest row) */
	int best[2], btc[2], brow[2];
	/* fill mode: best cell seen by this lane, per query half (value, first column, smallest row) */
	int tv[2], ttc[2], trow[2];
};

struct StripCtx {
	u32 prof, ring, ringb, bin, bout, nulloff;
	int l16, ncols, nsteps, c_edge, dirstep, store_from, row0;
	int ncols2[2], c_edge2[2];   /* capture: the two query halves have their own windows of the target */
	bool mine, first, last;
	const int8_t* tg;
	u32* bnd;          /* this job's boundary records */
	u32* o16; u32* o8; /* fill: column maxima of the last strip */
	u32* g16; u32* g8; /* fill, optional: maxima of every aligned group of 16 columns (as k_fill writes them): k_reduce_seg scans these */
	u32 gapO2, gapE2;
	int n;
	u32 bmask;         /* boundary-out ring: entries - 1 (63, or 31 in the LDS-trimmed queue kernel) */
	int bnd_avail;     /* traversal columns [0, bnd_avail) have a boundary record from the strip above (= ncols unless the strips walk a diagonal band) */
	int col_shift;     /* capture, banded reverse pass: this strip's traversal column 0 is column col_shift of the job's window */
	int fr_base, fr_kmask, gapEi;   /* column-frame form of the fill (run_strip<..., FR>) */
};

template <int PS> SSW_DEV u32 strip_code_off(const StripCtx& x, int h, int tc, bool capture)
{
	const int nc = capture ? x.ncols2[h] : x.ncols, ce = capture ? x.c_edge2[h] : x.c_edge;
	int code = tc < nc ? x.tg[ce + x.dirstep * tc] : x.n;
	if (code < 0 || code > x.n) code = x.n;
	return (u32)code * (u32)PS;
}

/* fill mode, best-cell tracking: `now` = the lane's running record after a step, `pre` = before it (this column's rows above folded
   in).  Only the lane whose OWN rows beat that -- the lane holding the new record cell, not every lane below it -- records: value,
   column, smallest row.  (Columns outside the target only decay; the test on tc keeps the hand-written contract explicit.  The
   branch-free form of k_filldb's db_record, one v_bfi per row, measured 13 ms slower on config 4's fill.) */
template <int R>
SSW_DEV void strip_record(u32 now, u32 pre, int tc, const StripCtx& x, const u32 (&H)[R], int (&sv)[2], int (&stc)[2], int (&srow)[2], int phi = 0)
{
	if (now != pre && x.mine && tc >= 0 && tc < x.ncols) {
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int nv = (int)((now >> (16 * h)) & 0xffffu), ov = (int)((pre >> (16 * h)) & 0xffffu);
			if (nv > ov) {
				sv[h] = nv - phi; stc[h] = tc; srow[h] = 0x7fffffff;      /* (frame form: now / pre / H carry + phi of that column) */
#pragma unroll
				for (int k = R - 1; k >= 0; --k) if ((int)((H[k] >> (16 * h)) & 0xffffu) == nv) srow[h] = x.row0 + x.l16 * R + k;
			}
		}
	}
}

/* last strip of a fill job: the maximum of the 16 finished columns [base, base + 16) that the staging lanes (one DPP row) have just
   flushed, for both padding rules -- what k_fill's fill_flush16 leaves for k_reduce_seg.  Executed by every lane of the wavefront (the
   butterfly stays inside a DPP row; only the chain's lane 0 stores). */
SSW_DEV void strip_group_max(const u32x4& rec, bool valid, int base, const StripCtx& x)
{
	u32 m16 = valid ? rec[2] : 0u, m8 = valid ? rec[3] : 0u;
	m16 = pk_max(m16, xl_row_ror<1>(m16)); m8 = pk_max(m8, xl_row_ror<1>(m8));
	m16 = pk_max(m16, xl_row_ror<2>(m16)); m8 = pk_max(m8, xl_row_ror<2>(m8));
	m16 = pk_max(m16, xl_row_ror<4>(m16)); m8 = pk_max(m8, xl_row_ror<4>(m8));
	m16 = pk_max(m16, xl_row_ror<8>(m16)); m8 = pk_max(m8, xl_row_ror<8>(m8));
	if (x.l16 == 0 && x.mine && base >= x.store_from && base < x.ncols) { x.g16[base >> 4] = m16; x.g8[base >> 4] = m8; }
}

/* FR (fill mode only): column-frame form of the recurrence (chain_rows_fr).  Boundary records travel between strips as TRUE values:
   the staging lanes add phi of the column when a record enters the boundary-in ring (lane 0 meets column tc at step tc) and
   subtract the parking lane's when it leaves the boundary-out ring (lane GL-1 finishes column tc at step tc + GL - 1). */
template <int R, bool CAPTURE, bool MASK8, int GL, bool CM3 = false, bool FR = false>
SSW_DEV void run_strip(unsigned char* lds, const StripCtx& x, ChainState<R>& st, const u32 (&m8)[R])
{
	typedef StripGeom<R, GL> G;
	constexpr int C = G::C, RB = G::RB;
	const int l16 = x.l16;             /* lane within the chain (0..GL-1) */
	const bool stg = l16 < 16;         /* the 16 lanes that stage the rings and flush the boundary records */
	const u32 fr_c1 = x.gapO2 - x.gapE2;
	auto bnd_in = [&](u32x4 rec, int tc) -> u32x4 {      /* a record of the strip above (true values) in the frame lane 0 has at column tc */
		if (FR) { const u32 p = pk_dup(fr_phi(tc, 0, GL, x.fr_base, x.fr_kmask, x.gapEi)); rec[0] += p; rec[1] += p; rec[2] += p; rec[3] += p; }
		return rec;
	};
	auto bnd_out = [&](u32x4 rec, int tc) -> u32x4 {     /* what lane GL-1 parked for column tc, as true values */
		if (FR) { const u32 p = pk_dup(fr_phi(tc + GL - 1, GL - 1, GL, x.fr_base, x.fr_kmask, x.gapEi)); rec[0] -= p; rec[1] -= p; rec[2] -= p; rec[3] -= p; }
		return rec;
	};
	/* rings: target columns -GL..-1 null, 0..15 now, 16..31 in flight; boundary-in likewise */
	lds_st16(lds, x.ring + 2u * (RB - GL + l16), x.nulloff);
	if (CAPTURE) lds_st16(lds, x.ringb + 2u * (RB - GL + l16), x.nulloff);
	u32 nxt = 0, nxtb = 0;
	if (stg) {
		const u32 off = strip_code_off<G::PSTRIDE>(x, 0, l16, CAPTURE);
		lds_st16(lds, x.ring + 2u * l16, off);
		lds_st16(lds, x.ring + 2u * (RB + l16), off);
		nxt = strip_code_off<G::PSTRIDE>(x, 0, 16 + l16, CAPTURE);
		if (CAPTURE) {
			const u32 offb = strip_code_off<G::PSTRIDE>(x, 1, l16, true);
			lds_st16(lds, x.ringb + 2u * l16, offb);
			lds_st16(lds, x.ringb + 2u * (RB + l16), offb);
			nxtb = strip_code_off<G::PSTRIDE>(x, 1, 16 + l16, true);
		}
	}
	const u32x4 zero4 = { 0u, 0u, 0u, 0u };
	const bool take = !x.first && x.mine && stg;
	if (stg) {
		u32x4 rec = zero4;
		if (take && l16 < x.bnd_avail) rec = *(const u32x4*)(x.bnd + 4 * (int64_t)l16);
		lds_st128(lds, x.bin + 16u * l16, bnd_in(rec, l16));
		lds_st128(lds, x.bin + 16u * (48 + l16), zero4);
	}
	u32x4 nb = zero4;
	if (take && 16 + l16 < x.bnd_avail) nb = *(const u32x4*)(x.bnd + 4 * (int64_t)(16 + l16));
	nb = bnd_in(nb, 16 + l16);
	/* frame form: the all-zero state of the column before the lane's first one; `fl` = phi one column ahead */
	const u32 zero0 = FR ? pk_dup(fr_phi(0, l16, GL, x.fr_base, x.fr_kmask, x.gapEi) - x.gapEi) : 0u;
	u32 fl = zero0 + x.gapE2;
#pragma unroll
	for (int r = 0; r < R; ++r) { st.H[r] = zero0; st.E[r] = FR ? fl : 0u; }
	st.Hlast = zero0; st.Fout = zero0; st.cmout = zero0; st.cm8out = zero0; st.hsave = zero0;
	const u32 lane_prof = x.prof + (u32)l16 * 16u;
	u32 sbest = zero0; int sv[2] = { 0, 0 }, stc[2] = { 0x7fffffff, 0x7fffffff }, srow[2] = { 0x7fffffff, 0x7fffffff };   /* this strip's tracking */
	if (CAPTURE) sbest = pk_subu(pk_make(st.best[0], st.best[1]), 0x00010001u) + zero0;      /* (frame form: in the frame of the lane's column, like everything it is compared with) */
	unsigned long long pend = 0ull; u32 prep = 0;   /* fill: lanes whose rows set a record in the step before, and that step's `pre` */
	wave_lds_fence();
	/* software pipeline of the LDS reads (two waves per SIMD do not hide their latency): the scores and the boundary record
	   of step s+1 and the ring entry of step s+2 are requested before step s computes.  Ring entries are staged a chunk ahead
	   and the mirror is 32 entries deep, so reading up to 17 entries past a chunk's first one is safe. */
	u32x4 sc_n[C], sb_n[C], rec_n;
	u32 pa_n, pb_n = 0;
	{
		const u32 r0 = 2u * (u32)((0 - l16) & (RB - 1));
		const u32 pa0 = lds_ld16(lds, x.ring + r0) + lane_prof;
#pragma unroll
		for (int c = 0; c < C; ++c) sc_n[c] = lds_ld128(lds, pa0 + G::CSTRIDE * c);
		if (CAPTURE) {
			const u32 pb0 = lds_ld16(lds, x.ringb + r0) + lane_prof;
#pragma unroll
			for (int c = 0; c < C; ++c) sb_n[c] = lds_ld128(lds, pb0 + G::CSTRIDE * c);
			pb_n = lds_ld16(lds, x.ringb + r0 + 2u) + lane_prof;
		}
		rec_n = lds_ld_bnd<MASK8>(lds, x.bin);
		pa_n = lds_ld16(lds, x.ring + r0 + 2u) + lane_prof;
	}

	for (int s0 = 0; s0 < x.nsteps; s0 += 16) {
		if (FR && s0 > 0 && (s0 & x.fr_kmask) == 0) {   /* renormalisation: every frame value drops by K x gapE */
			const u32 k = (u32)(x.fr_kmask + 1) * x.gapE2;
#pragma unroll
			for (int r = 0; r < R; ++r) { st.H[r] -= k; st.E[r] -= k; }
			st.Hlast -= k; st.Fout -= k; st.cmout -= k; st.cm8out -= k; st.hsave -= k; fl -= k; sbest -= k; prep -= k;
		}
		if (stg) {   /* stage [s0+16, s0+32), prefetch [s0+32, s0+48) */
			const int p = (s0 + 16 + l16) & (RB - 1);
			lds_st16(lds, x.ring + 2u * p, nxt);
			if (p < 32) lds_st16(lds, x.ring + 2u * (RB + p), nxt);
			if (CAPTURE) {
				lds_st16(lds, x.ringb + 2u * p, nxtb);
				if (p < 32) lds_st16(lds, x.ringb + 2u * (RB + p), nxtb);
			}
			lds_st128(lds, x.bin + 16u * (p & 63), nb);
			const int tc = s0 + 32 + l16;
			nxt = strip_code_off<G::PSTRIDE>(x, 0, tc, CAPTURE);
			if (CAPTURE) nxtb = strip_code_off<G::PSTRIDE>(x, 1, tc, true);
			nb = zero4;
			if (take && tc < x.bnd_avail) nb = *(const u32x4*)(x.bnd + 4 * (int64_t)tc);
			nb = bnd_in(nb, tc);
		}
		wave_lds_fence();
		if (s0 >= GL + 16) {   /* boundary-out records of columns [s0-GL-16, s0-GL) are complete */
			const int tc = s0 - GL - 16 + l16;
			u32x4 rec = zero4;
			const bool have = stg && x.mine && tc < x.ncols;
			if (have) {
				rec = bnd_out(lds_ld128(lds, x.bout + 16u * ((u32)tc & x.bmask)), tc);
				if (!x.last) *(u32x4*)(x.bnd + 4 * (int64_t)tc) = rec;
				else if (!CAPTURE && tc >= x.store_from) { x.o16[tc] = rec[2]; x.o8[tc] = rec[3]; }
			}
			if (!CAPTURE && x.last && x.g16) strip_group_max(rec, have && tc >= x.store_from, s0 - GL - 16, x);
		}
		wave_lds_fence();
		const u32 rpo = 2u * (u32)((s0 - l16) & (RB - 1));
#ifndef STRIP_UNROLL
#define STRIP_UNROLL 2
#endif
#pragma unroll STRIP_UNROLL
		for (int j = 0; j < 16; ++j) {
			const int s = s0 + j, tc = s - l16;
			u32x4 sc[C];
#pragma unroll
			for (int c = 0; c < C; ++c) sc[c] = sc_n[c];
			if (CAPTURE) {   /* the upper query half looks at its own target column */
#pragma unroll
				for (int c = 0; c < C; ++c)
#pragma unroll
					for (int k = 0; k < 4; ++k) sc[c][k] = (sc[c][k] & 0xffffu) | (sb_n[c][k] & 0xffff0000u);
			}
			const u32x4 rec = rec_n;       /* what lane 0 receives from the strip above */
			/* requests for the next steps */
			/* (the ring entries of step s+2 are requested FIRST: asked for after the score chunks, hipcc gave them a register of the last
			   chunk's unused rows -- R = 10 uses 10 of 12 -- and had to wait for every load in flight before it could issue them) */
			const u32 pa_next = lds_ld16(lds, x.ring + rpo + 2u * (j + 2));
			const u32 pb_next = CAPTURE ? lds_ld16(lds, x.ringb + rpo + 2u * (j + 2)) : 0u;
#pragma unroll
			for (int c = 0; c + 1 < C; ++c) sc_n[c] = lds_ld128(lds, pa_n + G::CSTRIDE * c);
			sc_n[C - 1] = lds_ld_rows<R - 4 * (C - 1)>(lds, pa_n + G::CSTRIDE * (C - 1));      /* the last chunk: only the rows that exist */
			if (CAPTURE) {
#pragma unroll
				for (int c = 0; c + 1 < C; ++c) sb_n[c] = lds_ld128(lds, pb_n + G::CSTRIDE * c);
				sb_n[C - 1] = lds_ld_rows<R - 4 * (C - 1)>(lds, pb_n + G::CSTRIDE * (C - 1));
				pb_n = pb_next + lane_prof;
			}
			rec_n = lds_ld_bnd<MASK8>(lds, x.bin + 16u * ((s + 1) & 63));
			pa_n = pa_next + lane_prof;
			const u32 hin = xl_chain_shr1_keep<GL>(rec[0], st.Hlast);
			u32 f = xl_chain_shr1_keep<GL>(rec[1], st.Fout);
			u32 cm = xl_chain_shr1_keep<GL>(rec[2], st.cmout);
			u32 cm8 = MASK8 ? xl_chain_shr1_keep<GL>(rec[3], st.cm8out) : 0u;
			u32 d = st.hsave;
			u32 lm = 0;   /* capture: this lane's own maximum in this column */
			/* fill: the record of the step BEFORE.  The scalar branch on "some lane set a record" is taken a step late, before the rows of
			   the next column run: the compare that feeds it is a whole step old by then, while at the end of its own step the scalar
			   unit would wait for the vector compare every time (config 4's fill: 1602 -> 1546 ms).  H still holds that column.
			   (k_filldb, where the branch is taken in most steps, got slower with the same change.) */
			if (!CAPTURE && pend != 0ull) strip_record<R>(sbest, prep, tc - 1, x, st.H, sv, stc, srow, FR ? (int)((fl - x.gapE2) & 0xffffu) : 0);
			if (FR) { fl += x.gapE2; sbest += x.gapE2; }      /* the lane's record follows the frame of its column */
			const u32 pre = pk_max(sbest, cm);   /* fill: the lane's running record and this column's rows above */
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const u32 hold = st.H[r];
				const u32 sv = sc[r >> 2][r & 3];
				u32 h;
				if (FR) {
					h = pk_max3_fr(CAPTURE ? pk_addw(d, sv) : d + sv, st.E[r], f);
					const u32 t = h - fr_c1;
					st.E[r] = pk_max3_fr(st.E[r], t, fl);
					f = pk_max(f, t) - x.gapE2;
				} else {
				const u32 h0 = pk_max(pk_adds(d, sv), st.E[r]);
				h = pk_max(h0, f);
				const u32 t0 = pk_subu(h0, x.gapO2);
				st.E[r] = pk_max(pk_subu(st.E[r], x.gapE2), t0);
				f = pk_max(pk_subu(f, x.gapE2), t0);
				}
				if (CAPTURE && FR) {   /* the lane's own maximum of this column, two rows per instruction */
					if (r & 1) lm = pk_max3_fr(lm, st.H[r - 1 >= 0 ? r - 1 : 0], h);
					else if (r == R - 1) lm = pk_max(lm, h);
				} else if (CAPTURE) lm = pk_max(lm, h);
				else if (CM3 && !MASK8) {   /* column maximum of two rows in one instruction (scores below 31744: pk_max3_nonneg) */
					if (r & 1) cm = pk_max3_nonneg(cm, st.H[r - 1 >= 0 ? r - 1 : 0], h);
					else if (r == R - 1) cm = pk_max(cm, h);
				} else cm = pk_max(cm, h);
				if (MASK8) cm8 = pk_max(cm8, h & m8[r]);
				st.H[r] = h;
				d = hold;
			}
			if (!CAPTURE) {
				/* best-cell tracking: `pre` = everything this lane knows of rows above and of earlier columns (its running
				   record and this column's maximum of the rows above).  Only the lane whose OWN rows beat that -- the lane
				   holding the new record cell, not every lane below it -- takes the branch. */
				sbest = pk_max(sbest, cm);      /* = max(pre, own rows); pre was taken before the rows */
				pend = wave_ballot(sbest != pre); prep = pre;
				sched_fence();
			}
			st.hsave = hin; st.Hlast = st.H[R - 1]; st.Fout = f; st.cmout = cm; st.cm8out = MASK8 ? cm8 : cm;
			if (l16 == GL - 1) {   /* (one 16-byte store; four dword stores instead -- no register copies to make the four consecutive -- measured 2.4 % slower) */
				const u32x4 o = { st.Hlast, st.Fout, st.cmout, st.cm8out };
				lds_st128(lds, x.bout + 16u * ((u32)(s - (GL - 1)) & x.bmask), o);
			}
			if (CAPTURE) {   /* rarely taken: some half reaches (at least) its best so far -- sbest holds best - 1 per half */
				if (pk_max(sbest, lm) != sbest && x.mine && tc >= 0) {
					const int phi = FR ? (int)((fl - x.gapE2) & 0xffffu) : 0;      /* lm, H and sbest carry the frame of this column */
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const int mf = (int)((lm >> (16 * h)) & 0xffffu), m = mf - phi;
						if (tc < x.ncols2[h] && (m > st.best[h] || (m == st.best[h] && m > 0 && tc + x.col_shift < st.btc[h]))) {
							st.best[h] = m; st.btc[h] = tc + x.col_shift;      /* (columns of the job's window: the strips of a banded pass start at different ones) */
#pragma unroll
							for (int k = R - 1; k >= 0; --k) if ((int)((st.H[k] >> (16 * h)) & 0xffffu) == mf) st.brow[h] = x.row0 + l16 * R + k;
						}
					}
					sbest = pk_subu(pk_make(st.best[0], st.best[1]), 0x00010001u) + (FR ? fl - x.gapE2 : 0u);
				}
			}
		}
	}
	if (!CAPTURE && pend != 0ull) strip_record<R>(sbest, prep, x.nsteps - 1 - l16, x, st.H, sv, stc, srow, FR ? (int)((fl - x.gapE2) & 0xffffu) : 0);
	wave_lds_fence();
	for (int base = x.nsteps - GL - 16; base < x.nsteps - GL + 16; base += 16) {
		const int tc = base + l16;
		u32x4 rec = zero4;
		const bool have = stg && x.mine && tc >= 0 && tc < x.ncols;
		if (have) {
			rec = bnd_out(lds_ld128(lds, x.bout + 16u * ((u32)tc & x.bmask)), tc);
			if (!x.last) *(u32x4*)(x.bnd + 4 * (int64_t)tc) = rec;
			else if (!CAPTURE && tc >= x.store_from) { x.o16[tc] = rec[2]; x.o8[tc] = rec[3]; }
		}
		if (!CAPTURE && x.last && x.g16 && base >= 0) strip_group_max(rec, have && tc >= x.store_from, base, x);
	}
	if (!CAPTURE) {   /* merge the strip's best cell into the lane's: higher value, then earlier column, then smaller row */
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int v = sv[h];      /* the last record this lane set itself in this strip */
			if (v > st.tv[h] || (v == st.tv[h] && v > 0 && (stc[h] < st.ttc[h] || (stc[h] == st.ttc[h] && srow[h] < st.trow[h])))) {
				st.tv[h] = v; st.ttc[h] = stc[h]; st.trow[h] = srow[h];
			}
		}
	}
	dev_fence();   /* the next strip of this chain re-reads the boundary records through HBM */
}

/* capture mode: one query half of a job (see k_chainx) */
struct CapHalf {
	int q, qlen, lena, rows, ncols, c_edge;
	int band;      /* > 0: capped reverse pass on a diagonal band -- a strip of rows [r0, r0 + n) only visits columns [r0 - band, r0 + n + band) */
	bool active, capped;
	ssw_dres r;
	const int8_t* qc;
};

SSW_DEV void cap_half_setup(CapHalf& h, const ssw_chainx_args& a, int q)
{
	h.q = q; h.active = false; h.capped = false; h.qlen = 0; h.lena = 0; h.rows = 0; h.ncols = 0; h.c_edge = 0; h.qc = a.qcodes; h.band = 0;
	if (q < 0) return;
	h.r = a.res[q];
	h.active = h.r.status == 0 && h.r.score1 > 0 && (a.reverse ? h.r.want_begin == 1 : !h.r.loc_done);
	if (!h.active) return;
	const int qs = vm_query(a.vm, q);
	h.qc = a.qcodes + a.qoff[qs];
	h.qlen = (int)(a.qoff[qs + 1] - a.qoff[qs]);
	h.lena = a.reverse ? h.r.read_end1 + 1 : h.qlen;
	h.rows = (h.lena + 15) & ~15;
	/* exact window for the known score1 (see k_capture): rows + (rows max(mat) - score1) / gapE + 1 columns */
	const long long lost0 = (long long)h.rows * (a.maxmat > 0 ? a.maxmat : 0) - h.r.score1;
	long long w = (long long)h.rows + (lost0 > 0 ? lost0 : 0) / (a.gapE > 0 ? a.gapE : 1) + 1;
	if (a.gapE <= 0 || w > h.r.ref_end1) w = h.r.ref_end1;
	if (a.reverse && a.window_extra >= 0) {   /* first try: the alignment rarely spans more than its rows + 25 % */
		const long long cap = (long long)h.rows + h.rows / 4 + a.window_extra;
		long long wfree = (long long)h.rows + ((long long)h.rows * (a.maxmat > 0 ? a.maxmat : 0) + a.gapE - 1) / (a.gapE > 0 ? a.gapE : 1) + 1;      /* the score-free bound */
		if (a.gapE <= 0 || wfree > h.r.ref_end1) wfree = h.r.ref_end1;
		if (cap < wfree) {      /* (as before round 5: the capped first try and its diagonal band are worth it; the window is the smaller of cap and the exact one) */
			if (cap < w) w = cap;
			h.capped = true;
			/* the diagonal band is only worth trying where its acceptance proof (cap_half_finish) can succeed: an alignment that scores
			   close to max(mat) per row.  A weak one -- an unrelated read's best local alignment, hundreds of rows of the linear regime --
			   would be found and then rerun with the exact window; it keeps the whole capped window instead. */
			/* how far a path of this score can stray from the diagonal: every residue of deviation costs >= gapE of the max(mat) * rows an
			   ungapped perfect alignment would score -- plus rows / 32 + window_extra of slack, at most the capped window's rows / 4 + extra */
			const long long mm = a.maxmat > 0 ? a.maxmat : 0;
			long long lost = (mm * h.rows - h.r.score1) / (a.gapE > 0 ? a.gapE : 1);
			if (lost < 0) lost = 0;
			long long band = lost + h.rows / 32 + a.window_extra;
			if (band > h.rows / 4 + a.window_extra) band = h.rows / 4 + a.window_extra;
			if (mm * (h.rows + h.rows / 64) - (long long)(a.gapO2 & 0xffffu) - (band - 1) * a.gapE < (long long)h.r.score1) h.band = (int)band;
		}
	}
	h.ncols = (int)w + 1;
	h.c_edge = a.reverse ? h.r.ref_end1 : h.r.ref_end1 - (int)w;
	if (a.vm.vq) h.c_edge += (int)a.vm.toff[a.vm.vt[q]];      /* pair jobs: the chain reads from the concatenated targets (the host keeps them below 2^31 residues) */
}

/* the window's best cell of one half -> the result record (same contract as k_capture) */
SSW_DEV void cap_half_finish(const CapHalf& h, const ssw_chainx_args& a, int bv, int bc, int br)
{
	const int q = h.q;
	if (!a.reverse) {
		a.res[q].read_end1 = (bv == h.r.score1) ? (br < h.qlen - 1 ? br : h.qlen - 1) : -1;
		if (bv != h.r.score1) a.res[q].status = 3;
	} else {
		/* A banded pass (a.banded: every strip only visited the columns within `band` of its rows) is accepted when no cell it skipped, at
		   or before the column found, can hold score1: a path through a cell (r, c) with |r - c| >= band and c <= bc has at most bc + 1
		   diagonal steps and gaps of >= band residues in one direction, i.e. scores <= max(mat) * (bc + 1) - gapO - (band - 1) * gapE.
		   Otherwise -- like a capped window that did not reproduce score1 -- the alignment is rerun with the exact window. */
		bool proven = true;
		if (a.banded && h.capped && h.band > 0 && bv == h.r.score1)      /* (band > 0: this job really walked the band, k_chainq) */
			proven = (long long)(a.maxmat > 0 ? a.maxmat : 0) * (bc + 1) - (long long)(a.gapO2 & 0xffffu) - (long long)(h.band - 1) * a.gapE < (long long)h.r.score1;
		if ((bv != h.r.score1 || !proven) && h.capped) { if (a.retry_count) atomicAdd(a.retry_count, 1); }   /* stays want_begin == 1: rerun uncapped */
		else if (bv != h.r.score1 && h.ncols <= h.r.ref_end1) a.res[q].status = 3;
		else {
			const int rb = h.r.ref_end1 - bc, qbeg = h.r.read_end1 - (br < h.lena - 1 ? br : h.lena - 1);
			a.res[q].want_begin = 2;   /* done */
			a.res[q].ref_begin1 = rb; a.res[q].read_begin1 = qbeg; a.res[q].rev_score = bv;
			if (h.r.score1 > bv) a.res[q].flag = 2;
			const int skip = (7 & a.flag) == 0 || ((2 & a.flag) != 0 && h.r.score1 < a.filters) ||
			                 ((4 & a.flag) != 0 && (h.r.ref_end1 - rb > a.filterd || h.r.read_end1 - qbeg > a.filterd));
			a.res[q].want_cigar = !skip;
		}
	}
}

/* CAPTURE = false: job = (pair, tile) of the forward fill.  CAPTURE = true: job = two queries of the list (2 job, 2 job + 1),
   one per 16-bit half, each with its own window of the target -- locate (forward) or begin position (reverse). */
template <int R, bool CAPTURE, int GL>
__global__ void __launch_bounds__(64) k_chainx(ssw_chainx_args a)
{
	typedef StripGeom<R, GL> G;
	SSW_DYN_LDS(lds);
	const int tid = (int)threadIdx.x, l16 = tid & (GL - 1), grp = tid / GL;
	const u32 prof_bytes = (u32)(a.n + 1) * G::PSTRIDE;
	StripCtx x;
	x.prof = (u32)grp * (prof_bytes + G::EXTRA); x.ring = x.prof + prof_bytes; x.ringb = x.ring + G::RINGB; x.bin = x.ringb + G::RINGB;
	x.bout = x.bin + BND_RING_BYTES; x.nulloff = (u32)a.n * G::PSTRIDE;
	const u32 red = x.bout + BND_RING_BYTES;
	x.l16 = l16; x.gapO2 = a.gapO2; x.gapE2 = a.gapE2; x.n = a.n; x.tg = CAPTURE && a.vm.vq ? a.vm.tcodes : a.tgt; x.bmask = 63u;
	x.fr_base = 0; x.fr_kmask = 0; x.gapEi = 0;
	const int job = (int)blockIdx.x * (64 / GL) + grp;
	const bool valid = job < a.njobs;

	const int8_t *qa = a.qcodes, *qb = 0;
	int lena = 0, lenb = 0, rev = 0, rowsa = 0, rowsb = 0, rows_total = 0, p8a = 0, p8b = 0;
	bool active = false;
	CapHalf ch[2];
	x.ncols = 0; x.c_edge = 0; x.dirstep = 1; x.store_from = 0; x.o16 = 0; x.o8 = 0; x.g16 = 0; x.g8 = 0; x.col_shift = 0;
	x.ncols2[0] = x.ncols2[1] = 0; x.c_edge2[0] = x.c_edge2[1] = 0;
	if (!CAPTURE) {
		if (valid) {
			const int pair = job / a.ntiles, t = job - pair * a.ntiles;
			const ssw_pair pr = a.pairs[pair];
			qa = a.qcodes + a.qoff[pr.qa]; lena = (int)(a.qoff[pr.qa + 1] - a.qoff[pr.qa]);
			if (pr.qb >= 0) { qb = a.qcodes + a.qoff[pr.qb]; lenb = (int)(a.qoff[pr.qb + 1] - a.qoff[pr.qb]); }
			rows_total = ((lena > lenb ? lena : lenb) + 15) & ~15;
			rowsa = (lena + 15) & ~15; rowsb = qb ? (lenb + 15) & ~15 : rows_total;      /* rows below a query's OWN padded length are dead for its half */
			p8a = (lena + 7) & ~7; p8b = qb ? (lenb + 7) & ~7 : rows_total;
			const int tile_lo = t * a.tile, tile_hi = tile_lo + a.tile < a.refLen ? tile_lo + a.tile : a.refLen;
			const int c_first = tile_lo - a.halo > 0 ? tile_lo - a.halo : 0;
			x.c_edge = c_first; x.ncols = tile_hi - c_first; x.store_from = tile_lo - c_first;
			x.o16 = a.cm16 + (int64_t)pair * a.cm_stride + c_first; x.o8 = a.cm8 + (int64_t)pair * a.cm_stride + c_first;
			if (a.sg16) { x.g16 = a.sg16 + (int64_t)pair * a.seg_stride + (c_first >> 4); x.g8 = a.sg8 + (int64_t)pair * a.seg_stride + (c_first >> 4); }      /* (tiles and halos are multiples of 16 columns) */
			active = true;
		}
	} else {
		cap_half_setup(ch[0], a, valid ? a.qlist[2 * job] : -1);
		cap_half_setup(ch[1], a, valid && 2 * job + 1 < a.nlist ? a.qlist[2 * job + 1] : -1);
		active = ch[0].active || ch[1].active;
		rev = a.reverse;
		x.dirstep = a.reverse ? -1 : 1;
		if (ch[0].active) { qa = ch[0].qc; lena = ch[0].lena; rowsa = ch[0].rows; }
		if (ch[1].active) { qb = ch[1].qc; lenb = ch[1].lena; rowsb = ch[1].rows; }
		rows_total = rowsa > rowsb ? rowsa : rowsb;
		for (int h = 0; h < 2; ++h) { x.ncols2[h] = ch[h].active ? ch[h].ncols : 0; x.c_edge2[h] = ch[h].c_edge; }
		x.ncols = x.ncols2[0] > x.ncols2[1] ? x.ncols2[0] : x.ncols2[1];
	}
	const int S = active ? (rows_total + GL * R - 1) / (GL * R) : 0;
	int maxS = S, mc = x.ncols;
#pragma unroll
	for (int sh = GL; sh < 64; sh <<= 1) {
		const int o1 = (int)xl_shfl((u32)maxS, (tid + sh) & 63), o2 = (int)xl_shfl((u32)mc, (tid + sh) & 63);
		maxS = o1 > maxS ? o1 : maxS; mc = o2 > mc ? o2 : mc;
	}
	x.nsteps = (mc + GL + 15) & ~15;
	x.bnd = a.bnd + (int64_t)(valid ? job : 0) * a.bnd_stride * 4;
	x.bnd_avail = x.ncols;

	ChainState<R> st;
	st.best[0] = st.best[1] = 0; st.btc[0] = st.btc[1] = 0x7fffffff; st.brow[0] = st.brow[1] = 0;
	st.tv[0] = st.tv[1] = 0; st.ttc[0] = st.ttc[1] = 0x7fffffff; st.trow[0] = st.trow[1] = 0x7fffffff;
	u32 m8[R];
	for (int sidx = 0; sidx < maxS; ++sidx) {
		x.mine = sidx < S; x.first = sidx == 0; x.last = sidx == S - 1; x.row0 = sidx * GL * R;
		build_profile_strip<R, GL>(lds, x.prof, l16, GL, a.mat, a.n, qa, lena, rev, x.mine ? rowsa : 0, qb, lenb, rev, x.mine ? rowsb : 0, x.row0);
		bool need_mask = false;
		if (!CAPTURE) {
			need_mask = x.mine && x.last && (p8a < rows_total || p8b < rows_total);
#pragma unroll
			for (int k = 0; k < R; ++k) {
				const int row = x.row0 + l16 * R + k;
				m8[k] = (row < p8a ? 0xffffu : 0u) | (row < p8b ? 0xffff0000u : 0u);
			}
		} else {
#pragma unroll
			for (int k = 0; k < R; ++k) m8[k] = 0;
		}
		if (!CAPTURE && wave_any(need_mask)) run_strip<R, CAPTURE, true, GL>(lds, x, st, m8);
		else run_strip<R, CAPTURE, false, GL>(lds, x, st, m8);
	}

	if (!CAPTURE && a.cand) {   /* best cell of this (pair, tile) job per query, for k_reduce (saves the locate pass) */
		for (int h = 0; h < 2; ++h) {
			lds_st32(lds, red + 12u * l16, (u32)st.tv[h]);
			lds_st32(lds, red + 12u * l16 + 4, (u32)st.ttc[h]);
			lds_st32(lds, red + 12u * l16 + 8, (u32)st.trow[h]);
			wave_lds_fence();
			if (l16 == 0 && valid) {
				int bv = 0, bc = 0x7fffffff, br = 0x7fffffff;
				for (int k = 0; k < GL; ++k) {
					const int v = (int)lds_ld32(lds, red + 12u * k), c = (int)lds_ld32(lds, red + 12u * k + 4), w = (int)lds_ld32(lds, red + 12u * k + 8);
					if (v > bv || (v == bv && v > 0 && (c < bc || (c == bc && w < br)))) { bv = v; bc = c; br = w; }
				}
				int32_t* cd = a.cand + ((int64_t)job * 2 + h) * 4;
				cd[0] = bv; cd[1] = bv > 0 ? x.c_edge + bc : -1; cd[2] = br; cd[3] = 0;
			}
			wave_lds_fence();
		}
	}
	if (CAPTURE) {
		for (int h = 0; h < 2; ++h) {
			lds_st32(lds, red + 12u * l16, (u32)st.best[h]);
			lds_st32(lds, red + 12u * l16 + 4, (u32)st.btc[h]);
			lds_st32(lds, red + 12u * l16 + 8, (u32)st.brow[h]);
			wave_lds_fence();
			if (l16 == 0 && ch[h].active) {
				int bv = 0, bc = 0x7fffffff, br = 0;
				for (int k = 0; k < GL; ++k) {
					const int v = (int)lds_ld32(lds, red + 12u * k), c = (int)lds_ld32(lds, red + 12u * k + 4), w = (int)lds_ld32(lds, red + 12u * k + 8);
					if (v > bv || (v == bv && v > 0 && (c < bc || (c == bc && w < br)))) { bv = v; bc = c; br = w; }
				}
				cap_half_finish(ch[h], a, bv, bc, br);
			}
			wave_lds_fence();
		}
	}
}

/* ================================================================================================
 * k_chainq: the 64-lane strip kernel behind a work queue.  k_chainx gives a job (a query pair x a target tile, or two
 * window passes) to one wavefront that walks the job's S row strips one after the other: a batch of J jobs is J
 * wavefronts of S strips each, which fills 2048 wavefront slots in ceil(J / 2048) rounds -- 5000 long-read pairs leave a
 * fifth of the device idle in the last round.  Here the unit of work is ONE STRIP: a persistent launch draws tickets
 * (item k = strip k / J of job k % J, strip-major) and a strip only waits for the completion flag of the strip above it,
 * which was drawn J tickets earlier and is normally long done.  What a wavefront carried in registers from strip to
 * strip (the best cell so far) travels through a 32-byte record per item instead; the boundary rows travel through HBM as
 * before.  LDS is trimmed (one target ring in fill mode, 32-entry boundary-out ring, reduction scratch aliased) so that
 * 12 rows per lane leave room for 8 wavefronts per CU.  FORM 3: column-frame form of the fill (run_strip<..., FR>), 0: plain int16.
 * ================================================================================================ */
template <int V> struct IntTag { static constexpr int value = V; };
template <int R, bool CAPTURE> struct QueueGeom {
	static constexpr int C = (R + 3) / 4;
	static constexpr u32 PSTRIDE = (u32)C * 1024u;
	static constexpr u32 RINGB = (128u + 32u) * 2u;
	static constexpr u32 BOUT = 32u * 16u;
	static constexpr u32 EXTRA = (CAPTURE ? 2u : 1u) * RINGB + BND_RING_BYTES + BOUT;   /* the 768-byte reduction scratch aliases the boundary-in ring */
};

template <int R, bool CAPTURE, int FORM>
__global__ void __launch_bounds__(64) k_chainq(ssw_chainx_args a)
{
	constexpr int GL = 64;
	typedef StripGeom<R, GL> G;
	typedef QueueGeom<R, CAPTURE> QG;
	SSW_DYN_LDS(lds);
	const int tid = (int)threadIdx.x, l16 = tid;
	const u32 prof_bytes = (u32)(a.n + 1) * G::PSTRIDE;
	StripCtx x;
	x.prof = 0; x.ring = prof_bytes; x.ringb = x.ring + (CAPTURE ? QG::RINGB : 0u); x.bin = x.ringb + QG::RINGB;
	x.bout = x.bin + BND_RING_BYTES; x.nulloff = (u32)a.n * G::PSTRIDE; x.bmask = 31u;
	const u32 red = x.bin;
	x.l16 = l16; x.gapO2 = a.gapO2; x.gapE2 = a.gapE2; x.n = a.n; x.tg = CAPTURE && a.vm.vq ? a.vm.tcodes : a.tgt;
	x.fr_base = a.fr_base; x.fr_kmask = a.fr_kmask; x.gapEi = (int)(a.gapE2 & 0xffffu);
	const int S = a.strips, nitems = a.njobs * S;
	int* const ticket = a.queue; int* const flags = a.queue + 1;

	/* a.whole_jobs: a ticket is a whole job whose strips this wavefront walks itself (no waiting on other wavefronts at all:
	   for batches with fewer jobs than wavefront slots a strip-level queue would only make wavefronts wait for each other) */
	const int nticket = a.whole_jobs ? a.njobs : nitems;
	for (;;) {
		int k = 0;
		if (tid == 0) k = dev_ticket(ticket);
		k = (int)xl_readlane((u32)k, 0);
		if (k >= nticket) break;
		const int s_first = a.whole_jobs ? 0 : k / a.njobs, s_end = a.whole_jobs ? S : s_first + 1;
		const int job = a.whole_jobs ? k : k - s_first * a.njobs;
		for (int sidx = s_first; sidx < s_end; ++sidx) {
		int* const my_flag = flags + (int64_t)job * S + sidx;
		int32_t* const my_cand = a.cand_strip + ((int64_t)job * S + sidx) * 8;

		const int8_t *qa = a.qcodes, *qb = 0;
		int lena = 0, lenb = 0, rev = 0, rowsa = 0, rowsb = 0, rows_total = 0, p8a = 0, p8b = 0;
		bool active = false;
		CapHalf ch[2];
		x.ncols = 0; x.c_edge = 0; x.dirstep = 1; x.store_from = 0; x.o16 = 0; x.o8 = 0; x.g16 = 0; x.g8 = 0; x.col_shift = 0;
		x.ncols2[0] = x.ncols2[1] = 0; x.c_edge2[0] = x.c_edge2[1] = 0;
		if (sidx > 0) {   /* everything the strip above wrote -- boundary records, its best cell, and (window passes) the records */
			if (!a.whole_jobs && tid == 0 && !dev_flag_wait(flags + (int64_t)job * S + sidx - 1)) atomicAdd(a.err, 1);   /* error word: the host fails the call */
			dev_fence();
		}
		if (!CAPTURE) {
			const int pair = job / a.ntiles, t = job - pair * a.ntiles;
			const ssw_pair pr = a.pairs[pair];
			qa = a.qcodes + a.qoff[pr.qa]; lena = (int)(a.qoff[pr.qa + 1] - a.qoff[pr.qa]);
			if (pr.qb >= 0) { qb = a.qcodes + a.qoff[pr.qb]; lenb = (int)(a.qoff[pr.qb + 1] - a.qoff[pr.qb]); }
			rows_total = ((lena > lenb ? lena : lenb) + 15) & ~15;
			rowsa = (lena + 15) & ~15; rowsb = qb ? (lenb + 15) & ~15 : rows_total;      /* rows below a query's OWN padded length are dead for its half */
			p8a = (lena + 7) & ~7; p8b = qb ? (lenb + 7) & ~7 : rows_total;
			const int tile_lo = t * a.tile, tile_hi = tile_lo + a.tile < a.refLen ? tile_lo + a.tile : a.refLen;
			const int c_first = tile_lo - a.halo > 0 ? tile_lo - a.halo : 0;
			x.c_edge = c_first; x.ncols = tile_hi - c_first; x.store_from = tile_lo - c_first;
			x.o16 = a.cm16 + (int64_t)pair * a.cm_stride + c_first; x.o8 = a.cm8 + (int64_t)pair * a.cm_stride + c_first;
			if (a.sg16) { x.g16 = a.sg16 + (int64_t)pair * a.seg_stride + (c_first >> 4); x.g8 = a.sg8 + (int64_t)pair * a.seg_stride + (c_first >> 4); }      /* (tiles and halos are multiples of 16 columns) */
			active = true;
		} else {
			cap_half_setup(ch[0], a, a.qlist[2 * job]);
			cap_half_setup(ch[1], a, 2 * job + 1 < a.nlist ? a.qlist[2 * job + 1] : -1);
			active = ch[0].active || ch[1].active;
			rev = a.reverse;
			x.dirstep = a.reverse ? -1 : 1;
			if (ch[0].active) { qa = ch[0].qc; lena = ch[0].lena; rowsa = ch[0].rows; }
			if (ch[1].active) { qb = ch[1].qc; lenb = ch[1].lena; rowsb = ch[1].rows; }
			rows_total = rowsa > rowsb ? rowsa : rowsb;
			for (int h = 0; h < 2; ++h) { x.ncols2[h] = ch[h].active ? ch[h].ncols : 0; x.c_edge2[h] = ch[h].c_edge; }
			x.ncols = x.ncols2[0] > x.ncols2[1] ? x.ncols2[0] : x.ncols2[1];
		}
		/* fill jobs of a launch with a.tail_R > 0 (all of the same padded length): S - 1 strips of R rows per lane and a LAST strip of
		   a.tail_R <= 4 rows per lane that takes the remainder (10 000 rows = 13 x 768 + 16: a fourteenth strip of 12 rows per lane
		   would compute 752 dead rows -- 7 % of the fill) */
		const bool tail_mode = !CAPTURE && a.tail_R > 0;
		const int Sjob = !active ? 0 : tail_mode ? S : (rows_total + GL * R - 1) / (GL * R);
		if (sidx >= Sjob) {   /* this job has fewer strips than the launch's S (window passes of short prefixes): nothing to do */
			if (tid == 0) dev_flag_set(my_flag);
			continue;
		}
		x.bnd = a.bnd + (int64_t)job * a.bnd_stride * 4;
		x.bnd_avail = x.ncols;
		if (CAPTURE && a.banded) {
			/* capped reverse pass on a diagonal band (round 4): the alignment leaves the end cell along the diagonal and strays from it by
			   at most its indel budget, so the strip of rows [r0, r0 + 64 R) only walks the columns [r0 - band, r0 + 64 R + band) of the
			   window -- 5 600 instead of 12 500 columns per strip of a 10-kb read.  Cells outside count as the zero boundary (a lower bound
			   of the true values); cap_half_finish accepts the result only with a proof that none of them could hold score1. */
			int band = 0;
			bool all_capped = true;
			for (int h = 0; h < 2; ++h) {
				if (ch[h].active && ch[h].capped && ch[h].band > band) band = ch[h].band;
				if (ch[h].active && (!ch[h].capped || ch[h].band <= 0)) all_capped = false;
			}
			if (band > 0 && all_capped) {
				const int r0 = sidx * GL * R;
				const int lo = r0 - band > 0 ? (r0 - band) & ~15 : 0;
				int hi_up = 0;
				x.ncols = 0;
				for (int h = 0; h < 2; ++h) {
					const int nc = ch[h].active ? ch[h].ncols : 0;
					int hi = r0 + GL * R + band; if (hi > nc) hi = nc;
					int hu = r0 + band; if (hu > nc) hu = nc;      /* where the strip above (rows [r0 - 64 R, r0)) stopped */
					x.ncols2[h] = hi - lo > 0 ? hi - lo : 0;
					x.c_edge2[h] = ch[h].c_edge + x.dirstep * lo;
					if (x.ncols2[h] > x.ncols) x.ncols = x.ncols2[h];
					if (hu > hi_up) hi_up = hu;
				}
				x.col_shift = lo;
				x.bnd += 4 * (int64_t)lo;
				x.bnd_avail = hi_up - lo > 0 ? hi_up - lo : 0;
				if (x.bnd_avail > x.ncols) x.bnd_avail = x.ncols;
			} else { ch[0].band = 0; ch[1].band = 0; }      /* whole windows for this job: nothing to prove */
		}
		x.nsteps = (x.ncols + GL + 15) & ~15;
		x.mine = true; x.first = sidx == 0; x.last = sidx == Sjob - 1; x.row0 = sidx * GL * R;

		/* the best cell so far of this job: what the strips above found (value, first column, smallest row per half) */
		int cv[2] = { 0, 0 }, cc[2] = { 0x7fffffff, 0x7fffffff }, cr[2] = { CAPTURE ? 0 : 0x7fffffff, CAPTURE ? 0 : 0x7fffffff };
		if (sidx > 0) {
			const int32_t* pc = a.cand_strip + ((int64_t)job * S + sidx - 1) * 8;
			for (int h = 0; h < 2; ++h) { cv[h] = pc[4 * h]; cc[h] = pc[4 * h + 1]; cr[h] = pc[4 * h + 2]; }
		}
		auto strip_body = [&](auto rr_tag) {
		constexpr int RR = decltype(rr_tag)::value;
		x.nulloff = (u32)a.n * StripGeom<RR, GL>::PSTRIDE;
		ChainState<RR> st;
		for (int h = 0; h < 2; ++h) {
			st.best[h] = CAPTURE ? cv[h] : 0; st.btc[h] = CAPTURE ? cc[h] : 0x7fffffff; st.brow[h] = CAPTURE ? cr[h] : 0;
			st.tv[h] = 0; st.ttc[h] = 0x7fffffff; st.trow[h] = 0x7fffffff;
		}
		if (CAPTURE) {
			/* A window pass only asks WHERE the known score1 is reached: cells below it need no record.  The lanes start from
			   score1 - 1 (no column, no row) wherever a window that does not hold score1 is not accepted anyway -- the forward
			   locate pass (status 3) and the capped first try of the reverse pass (rerun with the exact halo) -- so along a true
			   alignment, where a new best comes with every column, they stop recording.  The uncapped reverse pass keeps every
			   record: there a smaller best is an answer (flag 2 of the reference, ssw.c:932-935). */
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const int floorv = ch[h].active && (!a.reverse || ch[h].capped) ? ch[h].r.score1 - 1 : 0;
				if (floorv > st.best[h]) { st.best[h] = floorv; st.btc[h] = 0x7fffffff; st.brow[h] = 0; }
			}
		}
		build_profile_strip<RR, GL, FORM == 3 ? (CAPTURE ? 3 : 2) : 0>(lds, x.prof, l16, GL, a.mat, a.n, qa, lena, rev, rowsa, qb, lenb, rev, rowsb, x.row0, (int)(a.gapE2 & 0xffffu));
		u32 m8[RR];
		bool need_mask = false;
		if (!CAPTURE) {
			need_mask = x.last && (p8a < rows_total || p8b < rows_total);
#pragma unroll
			for (int q = 0; q < RR; ++q) {
				const int row = x.row0 + l16 * RR + q;
				m8[q] = (row < p8a ? 0xffffu : 0u) | (row < p8b ? 0xffff0000u : 0u);
			}
		} else {
#pragma unroll
			for (int q = 0; q < RR; ++q) m8[q] = 0;
		}
		if (!CAPTURE && need_mask) run_strip<RR, CAPTURE, true, GL, false, FORM == 3>(lds, x, st, m8);
		else run_strip<RR, CAPTURE, false, GL, FORM == 3, FORM == 3>(lds, x, st, m8);

		/* chain-wide winner of this strip merged with the strips above: value, then first column, then smallest row */
		for (int h = 0; h < 2; ++h) {
			lds_st32(lds, red + 12u * l16, (u32)(CAPTURE ? st.best[h] : st.tv[h]));
			lds_st32(lds, red + 12u * l16 + 4, (u32)(CAPTURE ? st.btc[h] : st.ttc[h]));
			lds_st32(lds, red + 12u * l16 + 8, (u32)(CAPTURE ? st.brow[h] : st.trow[h]));
			wave_lds_fence();
			if (l16 == 0) {
				int bv = cv[h], bc = cc[h], br = cr[h];
				for (int q = 0; q < GL; ++q) {
					const int v = (int)lds_ld32(lds, red + 12u * q), c = (int)lds_ld32(lds, red + 12u * q + 4), w = (int)lds_ld32(lds, red + 12u * q + 8);
					if (v > bv || (v == bv && v > 0 && (c < bc || (c == bc && w < br)))) { bv = v; bc = c; br = w; }
				}
				my_cand[4 * h] = bv; my_cand[4 * h + 1] = bc; my_cand[4 * h + 2] = br; my_cand[4 * h + 3] = 0;
				if (x.last) {
					if (!CAPTURE) {
						if (a.cand) { int32_t* cd = a.cand + ((int64_t)job * 2 + h) * 4; cd[0] = bv; cd[1] = bv > 0 ? x.c_edge + bc : -1; cd[2] = br; cd[3] = 0; }
					} else if (ch[h].active) cap_half_finish(ch[h], a, bv, bc, br);
				}
			}
			wave_lds_fence();
		}
		};      /* strip_body */
		if constexpr (!CAPTURE && R > 4) {
			if (tail_mode && x.last && a.tail_R == 1) strip_body(IntTag<1>());
			else if (tail_mode && x.last && a.tail_R == 2) strip_body(IntTag<2>());
			else if (tail_mode && x.last && a.tail_R <= 4) strip_body(IntTag<4>());
			else strip_body(IntTag<R>());
		} else strip_body(IntTag<R>());
		dev_fence();
		if (tid == 0) dev_flag_set(my_flag);
		}
	}
}

/* ================================================================================================
 * k_literal: the reference's two SSE2 kernels re-enacted lane for lane (see ssw_dev.h).  One DPP row = one __m128i;
 * 4 alignments per wavefront run in lockstep under per-row predicates.  Only used when gapO <= gapE.
 * Per alignment scratch (HBM, L1/L2 resident): H buffers x2, E, Hmax as [segment][16] int16, read codes, maxColumn.
 * ================================================================================================ */
struct LitOut { int score, ref, read, score2, ref2; };
#define LIT_U 8

SSW_DEV int lit_rowmax(int v)   /* maximum over the 16 lanes of a row, in every lane */
{
	u32 x = (u32)v;
	u32 y = xl_row_ror<1>(x); x = (int)y > (int)x ? y : x;
	y = xl_row_ror<2>(x); x = (int)y > (int)x ? y : x;
	y = xl_row_ror<4>(x); x = (int)y > (int)x ? y : x;
	y = xl_row_ror<8>(x); x = (int)y > (int)x ? y : x;
	return (int)x;
}
SSW_DEV int lit_rowmin(int v) { return -lit_rowmax(-v); }

/* one call == sw_sse2_byte (is_byte) / sw_sse2_word on one row of lanes; `on` = this row takes part */
SSW_DEV void literal_fill(bool on, bool is_byte, const int8_t* ref, int ref_dir, int refLen, const int8_t* read, int readLen, int rev_read,
                          const int8_t* mat, int n, int gapO, int gapE, int terminate, int bias, int maskLen,
                          unsigned char* state, uint16_t* mc, int tid, LitOut& out)
{
	/* `state`: the alignment's H / E / Hmax / code arrays -- every lane only touches its own column of them -- in LDS when they fit
	   (short reads: 16 alignments per workgroup), else in HBM scratch; `mc`: maxColumn, always in HBM */
	const int l16 = tid & 15, grp = (tid & 63) >> 4;
	const int L = is_byte ? 16 : 8;
	const bool lane_on = on && l16 < L;
	const int segLen = on ? (readLen + L - 1) / L : 0;
	int16_t* Hbuf0 = (int16_t*)state;
	int16_t* Hbuf1 = Hbuf0 + (size_t)segLen * 16;
	int16_t* Eb = Hbuf1 + (size_t)segLen * 16;
	int16_t* Hmx = Eb + (size_t)segLen * 16;
	int8_t* code = (int8_t*)(Hmx + (size_t)segLen * 16);
	/* uniform loop bounds for the 4 rows of the wavefront */
	int maxseg = segLen, maxcol = on ? refLen : 0;
#pragma unroll
	for (int sh = 16; sh < 64; sh <<= 1) {
		const int o1 = (int)xl_shfl((u32)maxseg, (tid + sh) & 63), o2 = (int)xl_shfl((u32)maxcol, (tid + sh) & 63);
		maxseg = o1 > maxseg ? o1 : maxseg; maxcol = o2 > maxcol ? o2 : maxcol;
	}
	for (int j = 0; j < segLen; ++j) {
		const int q = j + l16 * segLen;            /* striped layout: lane k holds rows k*segLen + j (ssw.c:169-186) */
		Hbuf0[j * 16 + l16] = 0; Hbuf1[j * 16 + l16] = 0; Eb[j * 16 + l16] = 0; Hmx[j * 16 + l16] = 0;
		code[j * 16 + l16] = (lane_on && q < readLen) ? (rev_read ? read[readLen - 1 - q] : read[q]) : (int8_t)-1;
	}
	for (int c = l16; c < (on ? refLen : 0); c += 16) mc[c] = 0;
	dev_fence();

	int max = 0, end_ref = is_byte ? -1 : 0, par = 0;
	bool alive = on;
	const int hi_sat = is_byte ? 255 : 32767;
	for (int it = 0; it < maxcol; ++it) {
		if (!wave_any(alive && it < refLen)) break;
		const bool colact = alive && it < refLen;
		const int i = ref_dir ? refLen - 1 - it : it;
		int16_t* Hst = par ? Hbuf0 : Hbuf1;       /* after the swap of ssw.c:269-271 */
		int16_t* Hld = par ? Hbuf1 : Hbuf0;
		const int8_t* mrow = mat + (colact ? (int)ref[i] : 0) * n;
		int prevlast = (colact && lane_on) ? (int)Hld[(segLen - 1) * 16 + l16] : 0;
		int vH = (int)xl_row_shr1_zero((u32)prevlast);
		int vF = 0, vMax = 0;
		/* the segments of a column in chunks of LIT_U: the chunk's inputs (read codes, E, the previous column's H -- addresses that do not
		   depend on the arithmetic) are requested together, then its scores, then the chunk is computed: one wavefront per SIMD (a batch of
		   a few thousand alignments is a few hundred wavefronts) cannot hide a memory round trip per segment, it can hide one per chunk */
		for (int j0 = 0; j0 < maxseg; j0 += LIT_U) {
			int cdv[LIT_U], ev[LIT_U], hlv[LIT_U], scv[LIT_U];
#pragma unroll
			for (int u = 0; u < LIT_U; ++u) {
				const int j = j0 + u;
				const bool act = colact && lane_on && j < segLen;
				cdv[u] = act ? (int)code[j * 16 + l16] : -1;
				ev[u] = act ? (int)Eb[j * 16 + l16] : 0;
				hlv[u] = act ? (int)Hld[j * 16 + l16] : 0;
			}
#pragma unroll
			for (int u = 0; u < LIT_U; ++u) scv[u] = cdv[u] >= 0 ? (int)mrow[cdv[u]] : 0;
#pragma unroll
			for (int u = 0; u < LIT_U; ++u) {
				const int j = j0 + u;
				const bool act = colact && lane_on && j < segLen;
				if (act) {
					const int sc = scv[u];
					int h;
					if (is_byte) { h = vH + sc + bias; if (h > 255) h = 255; h -= bias; if (h < 0) h = 0; }
					else { h = vH + sc; if (h > 32767) h = 32767; if (h < -32768) h = -32768; }
					int e = ev[u];
					h = h > e ? h : e; h = h > vF ? h : vF;
					vMax = vMax > h ? vMax : h;
					Hst[j * 16 + l16] = (int16_t)h;
					if (is_byte) { h = h - gapO; if (h < 0) h = 0; e = e - gapE; if (e < 0) e = 0; }
					else { const unsigned uh = (uint16_t)h, ue = (uint16_t)e; h = (int16_t)(uint16_t)(uh > (unsigned)gapO ? uh - gapO : 0); e = (int16_t)(uint16_t)(ue > (unsigned)gapE ? ue - gapE : 0); }
					e = e > h ? e : h;
					Eb[j * 16 + l16] = (int16_t)e;
					if (is_byte) { vF = vF - gapE; if (vF < 0) vF = 0; }
					else { const unsigned uf = (uint16_t)vF; vF = (int16_t)(uint16_t)(uf > (unsigned)gapE ? uf - gapE : 0); }
					vF = vF > h ? vF : h;
					vH = hlv[u];
				}
			}
		}
		/* lazy-F loop (ssw.c:302-315 / 509-520): stops as soon as no lane's F can still raise an H */
		bool lazy = colact;
		for (int k = 0; k < 16; ++k) {
			if (!wave_any(lazy && k < L)) break;
			const bool kact = lazy && k < L;
			vF = (int)xl_row_shr1_zero((u32)vF);
			for (int j = 0; j < maxseg; ++j) {      /* (reading a chunk of segments ahead here, like the sweep above does, measured 12 % slower: the loop usually ends within a segment or two) */
				if (!wave_any(lazy && k < L && j < segLen)) break;
				const bool act = kact && lazy && j < segLen;
				bool more = false;
				if (act && lane_on) {
					int h = Hst[j * 16 + l16];
					h = h > vF ? h : vF;
					vMax = vMax > h ? vMax : h;
					Hst[j * 16 + l16] = (int16_t)h;
					if (is_byte) { h = h - gapO; if (h < 0) h = 0; vF = vF - gapE; if (vF < 0) vF = 0; }
					else { const unsigned uh = (uint16_t)h, uf = (uint16_t)vF; h = (int16_t)(uint16_t)(uh > (unsigned)gapO ? uh - gapO : 0); vF = (int16_t)(uint16_t)(uf > (unsigned)gapE ? uf - gapE : 0); }
					more = vF > h;
				}
				const unsigned long long b = wave_ballot(more);
				if (act && ((b >> (16 * grp)) & 0xffffull) == 0) lazy = false;
			}
		}
		(void)hi_sat;
		const int cm = lit_rowmax(lane_on ? vMax : 0);
		bool brk = false;
		if (colact) {
			if (cm > max) {
				max = cm;
				if (is_byte && max + bias >= 255) brk = true;
				else {
					end_ref = i;
					for (int j = 0; j < segLen; ++j) Hmx[j * 16 + l16] = Hst[j * 16 + l16];
				}
			}
			if (!brk) { if (l16 == 0) mc[i] = (uint16_t)cm; if (cm == terminate) brk = true; }
			if (brk) alive = false;
		}
		par ^= 1;
	}
	dev_fence();
	/* read end: smallest row holding the maximum in the saved column (ssw.c:342-351) */
	int end_read = readLen - 1;
	if (lane_on)
		for (int j = 0; j < segLen; ++j)
			if ((int)(is_byte ? (uint16_t)Hmx[j * 16 + l16] : Hmx[j * 16 + l16]) == max) { const int row = j + l16 * segLen; if (row < end_read) end_read = row; }
	end_read = lit_rowmin(on ? end_read : 0x7fffffff);
	/* second best (ssw.c:368-381 / 570-583) */
	int s2 = 0, i2 = 0x7fffffff;
	if (on) {
		const int lo_edge = end_ref - maskLen > 0 ? end_ref - maskLen : 0;
		const int hi_edge = end_ref + maskLen > refLen ? refLen : end_ref + maskLen;
		const int up_from = is_byte ? hi_edge + 1 : hi_edge;
		for (int c = l16; c < refLen; c += 16)
			if (c < lo_edge || c >= up_from) { const int v = mc[c]; if (v > s2) { s2 = v; i2 = c; } }
	}
	const int s2m = lit_rowmax(s2);
	const int i2m = lit_rowmin(s2 == s2m && s2m > 0 ? i2 : 0x7fffffff);
	out.score = (is_byte && max + bias >= 255) ? 255 : max;
	out.ref = end_ref; out.read = end_read; out.score2 = s2m; out.ref2 = s2m > 0 ? i2m : 0;
}

__global__ void __launch_bounds__(256) k_literal(ssw_literal_args a)
{
	SSW_DYN_LDS(lds);
	const int tid = (int)threadIdx.x, l16 = tid & 15, grp = tid >> 4;
	const int job = (int)blockIdx.x * ((int)blockDim.x >> 4) + grp;
	/* a.spec (forward pass of a batch that leaves most of the device idle, score_size 2): BOTH rule sets of every query at once -- jobs
	   [0, nb) run the 8-bit kernel, jobs [nb, nb + nq) the 16-bit kernel of query job - nb, nb = nq rounded up to whole wavefronts -- instead
	   of the 16-bit kernel after the 8-bit one has overflowed; the later of a query's two jobs writes the record (spec_cnt).  For DNA reads
	   under 2/-2 nearly every 150-bp read overflows: the forward pass takes the time of one kernel instead of two. */
	const bool spec = a.pass == 0 && a.spec_cnt != (int32_t*)0;
	const int nb = spec ? (a.nq + 3) & ~3 : a.nq;
	const int kind = !spec ? 0 : job < nb ? 1 : 2;                  /* 0: as score_size says; 1: 8-bit rules only; 2: 16-bit rules only */
	const int jq = kind == 2 ? job - nb : job;
	const int njobs = spec ? nb + a.nq : a.nq;
	const int q = jq < a.nq && job < njobs ? a.qlist[jq] : -1;
	unsigned char* const gscr = a.scratch + (int64_t)(job < njobs ? job : 0) * a.scratch_stride;
	unsigned char* const scratch = a.lds_stride > 0 ? lds + (size_t)grp * (size_t)a.lds_stride : gscr;
	uint16_t* const mc = (uint16_t*)(gscr + a.mc_off);
	/* the scoring matrix in LDS, behind the per-alignment state regions: mat[ref][read] is a dependent look-up in every segment */
	int8_t* const lmat = (int8_t*)(lds + (size_t)((int)blockDim.x >> 4) * (size_t)a.lds_stride);
	for (int k = tid; k < a.n * a.n; k += (int)blockDim.x) lmat[k] = a.mat[k];
	__syncthreads();
	const int8_t* read = q >= 0 ? a.qcodes + a.qoff[q] : a.qcodes;
	const int readLen = q >= 0 ? (int)(a.qoff[q + 1] - a.qoff[q]) : 0;
	const int maskLen = a.maskLen >= 0 ? a.maskLen : readLen / 2;
	const bool have_byte = a.score_size == 0 || a.score_size == 2, have_word = a.score_size == 1 || a.score_size == 2;
	LitOut o;
	if (a.pass == 0) {
		ssw_dres r;
		r.score1 = 0; r.score2 = 0; r.ref_begin1 = -1; r.ref_end1 = 0; r.read_begin1 = -1; r.read_end1 = 0;
		r.ref_end2 = 0; r.cigarLen = 0; r.flag = 0; r.status = 0; r.word = 0; r.want_begin = 0; r.want_cigar = 0;
		r.rev_score = 0; r.loc_done = 0; r.nm = 0; r.cigar_off = 0;
		bool need_word = q >= 0 && (kind == 2 || (kind == 0 && !have_byte));
		bool done = q < 0;
		const bool do_byte = q >= 0 && (kind == 1 || (kind == 0 && have_byte));
		if (wave_any(do_byte)) {
			literal_fill(do_byte, true, a.tgt, 0, a.refLen, read, readLen, 0, lmat, a.n, a.gapO, a.gapE, 255, a.bias, maskLen,
			             scratch, mc, tid, o);
			if (do_byte && kind == 0) {
				if (o.score == 255) { if (have_word) need_word = true; else { r.status = 1; done = true; } }
			}
		}
		if (wave_any(need_word)) {
			LitOut w;
			literal_fill(need_word, false, a.tgt, 0, a.refLen, read, readLen, 0, lmat, a.n, a.gapO, a.gapE, 65535, 0, maskLen, scratch, mc, tid, w);
			if (need_word) { o = w; r.word = 1; }
		}
		if (spec) {
			/* both rule sets ran side by side: each job parks its outcome, the later one decides like ssw_align does (src/ssw.c:881-893: the 16-bit
			   kernel's answer iff the 8-bit one saturated) and writes the record */
			int32_t* const mine = a.spec_out + ((int64_t)(q >= 0 ? q : 0) * 2 + (kind == 2 ? 1 : 0)) * 8;
			if (q >= 0 && l16 == 0) { mine[0] = o.score; mine[1] = o.ref; mine[2] = o.read; mine[3] = o.score2; mine[4] = o.ref2; }
			dev_fence();
			int second = 0;
			if (q >= 0 && l16 == 0) second = atomicAdd(a.spec_cnt + q, 1) == 1;
			dev_fence();
			if (!second) done = true;
			else {
				const int32_t* const ob = a.spec_out + ((int64_t)q * 2) * 8;
				const int32_t* const ow = ob + 8;
				const int32_t* const pick = ob[0] == 255 ? ow : ob;
				o.score = pick[0]; o.ref = pick[1]; o.read = pick[2]; o.score2 = pick[3]; o.ref2 = pick[4];
				r.word = ob[0] == 255 ? 1 : 0;
			}
			if (l16 != 0) done = true;
		}
		if (q >= 0 && !done && o.score > 0) {
			r.score1 = o.score; r.ref_end1 = o.ref; r.read_end1 = o.read;
			if (maskLen >= 15) { r.score2 = o.score2; r.ref_end2 = o.ref2; } else { r.score2 = 0; r.ref_end2 = -1; }
			r.want_begin = !(a.flag == 0 || (a.flag == 2 && o.score < a.filters));
		}
		if (q >= 0 && l16 == 0 && (!spec || !done)) a.res[q] = r;
	} else {
		ssw_dres r;
		bool act = false;
		if (q >= 0) { r = a.res[q]; act = r.status == 0 && r.score1 > 0 && r.want_begin != 0; }
		const bool actb = act && !r.word, actw = act && r.word;
		const int plen = act ? r.read_end1 + 1 : 0, cols = act ? r.ref_end1 + 1 : 0;
		if (wave_any(actb)) {
			LitOut w;
			literal_fill(actb, true, a.tgt, 1, cols, read, plen, 1, lmat, a.n, a.gapO, a.gapE, r.score1 & 0xff, a.bias, maskLen, scratch, mc, tid, w);
			if (actb) o = w;
		}
		if (wave_any(actw)) {
			LitOut w;
			literal_fill(actw, false, a.tgt, 1, cols, read, plen, 1, lmat, a.n, a.gapO, a.gapE, r.score1 & 0xffff, 0, maskLen, scratch, mc, tid, w);
			if (actw) o = w;
		}
		if (act && l16 == 0) {
			const int rb = o.ref, qb = r.read_end1 - o.read;
			a.res[q].ref_begin1 = rb; a.res[q].read_begin1 = qb; a.res[q].rev_score = o.score;
			if (r.score1 > o.score) a.res[q].flag = 2;
			const int skip = (7 & a.flag) == 0 || ((2 & a.flag) != 0 && r.score1 < a.filters) ||
			                 ((4 & a.flag) != 0 && (r.ref_end1 - rb > a.filterd || r.read_end1 - qb > a.filterd));
			a.res[q].want_cigar = !skip;
		}
	}
}

/* ================================================================================================
 * k_trace: banded_sw + cigar re-score + band retry, one thread per alignment (scalar int32, exactly the
 * reference's control flow; the band of short reads is a handful of cells wide).
 * ================================================================================================ */
SSW_DEV int band_u(int w, int i, int j) { int x = i - w; if (x < 0) x = 0; return j - x + 1; }       /* ssw.c:92 */
SSW_DEV int band_d(int w, int i, int j, int p) { int x = i - w; if (x < 0) x = 0; return (j - x) * 3 + p; }  /* ssw.c:95 */

SSW_DEV int trace_one(const int8_t* ref, const int8_t* read, int refLen, int readLen, int score, int gapO, int gapE,
                      int band_width, const int8_t* mat, int n, unsigned char* scratch, int64_t cap,
                      u32* cig, int cigcap, int64_t* need)
{
	const int NEG = -1073741824;   /* INT32_MIN / 2 */
	const int len = refLen > readLen ? refLen : readLen;
	int best = 0, best_i = 0, best_j = 0, width, width_d, i, j;
	int *hb, *eb, *hc; int8_t* dir;
	do {
		width = band_width * 2 + 3; width_d = band_width * 2 + 1;
		const int64_t rowbytes = (((int64_t)width + 1) * 4 + 15) & ~(int64_t)15;
		const int64_t want = 3 * rowbytes + (int64_t)width_d * readLen * 3 + 16;
		if (want > cap) { *need = want; return -2; }
		hb = (int*)scratch; eb = (int*)(scratch + rowbytes); hc = (int*)(scratch + 2 * rowbytes);
		dir = (int8_t*)(scratch + 3 * rowbytes);
		for (j = 1; j < width - 1; ++j) hb[j] = 0;
		for (i = 0; i < readLen; ++i) {
			const int beg = i - band_width > 0 ? i - band_width : 0;
			const int end = i + band_width < refLen - 1 ? i + band_width : refLen - 1;
			const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
			int f = NEG, u = 0;
			int8_t* line = dir + (int64_t)width_d * i * 3;
			hb[0] = 0; hb[edge] = 0; hc[0] = 0;
			eb[0] = NEG; eb[edge] = NEG;
			for (j = beg; j <= end; ++j) {
				u = band_u(band_width, i, j);
				const int up = band_u(band_width, i - 1, j), lf = band_u(band_width, i, j - 1), dg = band_u(band_width, i - 1, j - 1);
				int open = i == 0 ? -gapO : hb[up] - gapO;
				int ext = i == 0 ? NEG : eb[up] - gapE;
				const int e = open > ext ? open : ext;
				const int8_t de = open > ext ? 3 : 2;
				eb[u] = e;
				line[band_d(band_width, i, j, 0)] = de;
				open = hc[lf] - gapO; ext = f - gapE;
				f = open > ext ? open : ext;
				const int8_t df = open > ext ? 5 : 4;
				line[band_d(band_width, i, j, 1)] = df;
				const int e1 = e > 0 ? e : 0, f1 = f > 0 ? f : 0;
				const int gap = e1 > f1 ? e1 : f1;
				const int dia = hb[dg] + mat[(int)ref[j] * n + read[i]];
				const int h = gap > dia ? gap : dia;
				hc[u] = h;
				if (h > best) { best = h; best_i = i; best_j = j; }
				line[band_d(band_width, i, j, 2)] = gap <= dia ? (int8_t)1 : (e1 > f1 ? de : df);
			}
			for (j = 1; j <= u; ++j) hb[j] = hc[j];
		}
		band_width *= 2;
	} while (best < score && band_width <= len);
	band_width /= 2;

	/* walk back from the best cell, emitting run-length ops last-to-first (ssw.c:682-762) */
	int nops = 0, run = 0, state = 2, cur = 0, prev = 0;
	i = best_i; j = best_j;
	while (i >= 0 && j > 0) {
		const int8_t d = dir[(int64_t)width_d * i * 3 + band_d(band_width, i, j, state)];
		if (d == 1) { --i; --j; state = 2; cur = 0; }
		else if (d == 2) { --i; state = 0; cur = 1; }
		else if (d == 3) { --i; state = 2; cur = 1; }
		else if (d == 4) { --j; state = 1; cur = 2; }
		else if (d == 5) { --j; state = 2; cur = 2; }
		else return -1;
		if (cur == prev) ++run;
		else { if (nops < cigcap) cig[nops] = ((u32)run << 4) | (u32)prev; ++nops; prev = cur; run = 1; }
	}
	if (cur == 0) { if (nops < cigcap) cig[nops] = ((u32)(run + 1) << 4); ++nops; }
	else {
		if (nops < cigcap) cig[nops] = ((u32)run << 4) | (u32)cur; ++nops;
		if (nops < cigcap) cig[nops] = (1u << 4); ++nops;
	}
	if (nops > cigcap) { *need = -(int64_t)nops; return -2; }
	for (int x = 0, y = nops - 1; x < y; ++x, --y) { const u32 t = cig[x]; cig[x] = cig[y]; cig[y] = t; }
	return nops;
}

SSW_DEV int cigar_score(const u32* cig, int n_ops, const int8_t* ref, const int8_t* read, const int8_t* mat, int n, int gapO, int gapE)
{
	int score = 0, rp = 0, qp = 0;
	for (int i = 0; i < n_ops; ++i) {
		const int len = (int)(cig[i] >> 4), op = (int)(cig[i] & 0xf);
		if (op == 0) { for (int k = 0; k < len; ++k) score += mat[(int)ref[rp++] * n + read[qp++]]; }
		else { score -= gapO + (len > 1 ? (len - 1) * gapE : 0); if (op == 1) qp += len; else if (op == 2) rp += len; }
	}
	return score;
}

/* cigar_alignment_score (src/ssw.c:785-811) by one wavefront: the residues of an M run spread over the 64 lanes (matrix in LDS) */
SSW_DEV int cigar_score_wave(const u32* cig, int n_ops, const int8_t* ref, const int8_t* read, const unsigned char* lds, int n, int gapO, int gapE, int lane)
{
	int part = 0, gaps = 0, rp = 0, qp = 0;
	for (int i = 0; i < n_ops; ++i) {
		const int len = (int)(cig[i] >> 4), op = (int)(cig[i] & 0xf);
		if (op == 0) { for (int x = lane; x < len; x += 64) part += lds_ld8s(lds, (u32)((int)ref[rp + x] * n + read[qp + x])); rp += len; qp += len; }
		else { gaps += gapO + (len > 1 ? (len - 1) * gapE : 0); if (op == 1) qp += len; else if (op == 2) rp += len; }
	}
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) part += (int)xl_shfl((u32)part, lane ^ d);
	return part - gaps;
}

/* ------------------------------------------------------------------------------------------------
 * trace_wave: the same banded_sw, one WAVEFRONT per alignment (long reads: bands of hundreds of cells, 10^4 rows).
 * The cells of a band row are spread over the 64 lanes in chunks; E and the diagonal term only need the previous
 * row; the horizontal dependency F[u] = max(h[u-1] - gapO, F[u-1] - gapE) with h = max(e, dia, 0, F) unrolls to the
 * max-plus recurrence F[u] = max(A[u-1] - gapO, F[u-1] - min(gapO, gapE)), A = max(e, dia, 0), which is a prefix scan
 * (6 shuffle steps per chunk).  All values are the reference's exact integers, so every direction byte, the running
 * best cell (row-major, strict >) and hence the CIGAR are identical to the scalar walk.
 * ------------------------------------------------------------------------------------------------ */
SSW_DEV int wave_bcast(int v, int src) { return (int)xl_shfl((u32)v, src); }

/* LDS of the wavefront / workgroup traceback: [matrix 1024][exchange 512][h_b][e_b][h_c][target window ring] */
#define TRACE_LDS_FIXED 1536u
SSW_HD u32 trace_ring_size(int band_width, int nthreads) { u32 r = 256; while (r < (u32)(2 * band_width + 2 * nthreads + 2)) r <<= 1; return r; }
/* cells per thread of trace_band_blocked for a row of width_d cells on nthreads threads (0: the row takes the other form) */
SSW_HD int trace_cpt_class(int width_d, int nthreads)
{
	const int cpt = (width_d + nthreads - 1) / nthreads;
	if (cpt <= 1 && nthreads == 64) return 1;        /* one wavefront, one cell per lane (rows need no padding: the stride between lanes is one dword) */
	return cpt <= 2 ? 2 : cpt <= 4 ? 4 : nthreads < 256 ? 0 : cpt <= 8 ? 8 : cpt <= 12 ? 12 : 0;
}
/* One band row in LDS.  trace_band_blocked keeps a thread's C cells C + 1 entries apart (entry of cell u >= 1: u + 1 + (u - 1) / C; cell 0
   at entry 0): the threads of a wavefront read and write their k-th cells at a stride of C + 1 dwords -- odd, hence one LDS bank each; at a
   stride of C = 8 dwords 32 lanes share four banks and every access costs eight LDS cycles (which, not the arithmetic, bounded a row).
   + 12 cells: a thread reads up to 11 cells past its last one. */
SSW_HD int64_t trace_rowbytes(int band_width, int nthreads)
{
	const int64_t cells = (int64_t)(band_width * 2 + 3) + 1 + 12;
	const int C = trace_cpt_class(band_width * 2 + 1, nthreads);
	return ((cells + (C > 1 ? cells / C + 3 : 0)) * 4 + 15) & ~(int64_t)15;
}
SSW_HD int64_t trace_lds_need(int band_width, int nthreads)
{
	const int64_t rowbytes = trace_rowbytes(band_width, nthreads);
	/* one (h, F) slot per thread: trace_band_blocked */
	return TRACE_LDS_FIXED + 8 * (int64_t)nthreads + 3 * rowbytes + (int64_t)trace_ring_size(band_width, nthreads);
}

/* row storage: LDS offsets (L) or the scratch arrays in HBM */
template <bool L> struct TraceRows {
	unsigned char* lds; u32 ohb, oeb, ohc;
	int *hb, *eb, *hc;
	SSW_DEVM int ldhb(int k) const { return L ? (int)lds_ld32(lds, ohb + 4u * (u32)k) : hb[k]; }
	SSW_DEVM int ldeb(int k) const { return L ? (int)lds_ld32(lds, oeb + 4u * (u32)k) : eb[k]; }
	SSW_DEVM int ldhc(int k) const { return L ? (int)lds_ld32(lds, ohc + 4u * (u32)k) : hc[k]; }
	SSW_DEVM void sthb(int k, int v) const { if (L) lds_st32(lds, ohb + 4u * (u32)k, (u32)v); else hb[k] = v; }
	SSW_DEVM void steb(int k, int v) const { if (L) lds_st32(lds, oeb + 4u * (u32)k, (u32)v); else eb[k] = v; }
	SSW_DEVM void sthc(int k, int v) const { if (L) lds_st32(lds, ohc + 4u * (u32)k, (u32)v); else hc[k] = v; }
};
/* all threads of the alignment's team (NW wavefronts) see each other's LDS / scratch writes afterwards */
template <bool L, int NW> SSW_DEV void trace_sync()
{
	if (NW > 1) { if (!L) { wg_fence(); __syncthreads(); } else lds_barrier(); }
	else if (L) wave_lds_fence();
	else wg_fence();
}

struct TraceBest { int best, i, j; };

/* exchange area (LDS offset 1024): [0,64) wave totals, [64,128) last h, [128,192) last F, 192.. carries F A H, 208.. broadcast,
   256.. per-wave best (3 ints each) */
#define TX_T 1024u
#define TX_H 1088u
#define TX_F 1152u
#define TX_CARRY 1216u
#define TX_BCAST 1232u
#define TX_BEST 1280u
#define TX_READ 1472u      /* 64 bytes: the read's codes of the 64 rows being walked (after the 16 x 12 bytes of TX_BEST) */

/* one band width: fills the direction bytes, updates the running best cell (row-major, strict >).  NW wavefronts
   (64 * NW cells per chunk) work on one alignment; the horizontal dependency is a two-level max-plus scan. */
template <bool L, int NW>
SSW_DEV void trace_band(const TraceRows<L>& R, const int8_t* ref, const int8_t* read, int refLen, int readLen,
                        int gapO, int gapE, int band_width, const int8_t* mat, int n, int8_t* dir, u32 oring,
                        u32 ring_mask, TraceBest& tb, int tid)
{
	constexpr int NT = 64 * NW;
	const int NEG = -1073741824;
	const int m = gapO < gapE ? gapO : gapE;
	const int lane = tid & 63, wv = tid >> 6;
	const int width = band_width * 2 + 3, width_d = band_width * 2 + 1;
	unsigned char* lds = R.lds;
	for (int j = 1 + tid; j < width - 1; j += NT) R.sthb(j, 0);
	int staged = 0;
	int lb = tb.best, li = 0, lj = 0;      /* this thread's first cell above everything it saw before */
	trace_sync<L, NW>();
	int rd = read[0];
	for (int i = 0; i < readLen; ++i) {
		/* (a global load in the row loop waits, on this target, for the row's direction-byte STORES as well -- one vmcnt for both: the rows that
		   live in LDS take the read's codes from an LDS window of 64 rows, refilled with the target window) */
		const int rdn = L ? 0 : i + 1 < readLen ? read[i + 1] : 0;             /* HBM rows: next row's base, in flight during this row */
		const int xi = i - band_width > 0 ? i - band_width : 0;
		const int xp = i - 1 - band_width > 0 ? i - 1 - band_width : 0;
		const int sft = xi - xp;                                   /* 0 or 1: how far the band slid against the previous row */
		const int beg = xi;
		const int end = i + band_width < refLen - 1 ? i + band_width : refLen - 1;
		const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
		const int ncell = end - beg + 1;
		int8_t* line = dir + (int64_t)(band_width + 1) * i;     /* one nibble per cell, a row of 2 b + 1 cells in b + 1 bytes */
		if (L && (i & 63) == 0) {   /* target window: everything the next 64 rows can touch */
			int64_t hi = (int64_t)i + band_width + 65; if (hi > refLen) hi = refLen;
			for (int j = staged + tid; j < (int)hi; j += NT) lds_st8(lds, oring + ((u32)j & ring_mask), (u32)(unsigned char)ref[j]);
			if ((int)hi > staged) staged = (int)hi;
			if (tid < 64) lds_st8(lds, TX_READ + (u32)tid, (u32)(unsigned char)read[i + tid < readLen ? i + tid : readLen - 1]);
		}
		if (tid == 0) { R.sthb(0, 0); R.sthb(edge, 0); R.sthc(0, 0); R.steb(0, NEG); R.steb(edge, NEG); }
		trace_sync<L, NW>();
		if (L) rd = lds_ld8s(lds, TX_READ + (u32)(i & 63));
		int carryF = NEG, carryA = 0, carryH = 0;                  /* F, A and h of the cell left of the chunk (h_c[0] = 0) */
		for (int c0 = 1; c0 <= ncell; c0 += NT) {
			const int u = c0 + tid;
			const bool ok = u <= ncell;
			const int j = beg + u - 1;
			int e = NEG, dia = NEG; int8_t de = 2;
			if (ok) {
				const int up = u + sft;
				const int open = i == 0 ? -gapO : R.ldhb(up) - gapO;
				const int ext = i == 0 ? NEG : R.ldeb(up) - gapE;
				e = open > ext ? open : ext; de = open > ext ? 3 : 2;
				const int sc = L ? lds_ld8s(lds, (u32)((int)lds_ld8s(lds, oring + ((u32)j & ring_mask)) * n + rd))
				                 : (int)mat[(int)ref[j] * n + rd];
				dia = R.ldhb(up - 1) + sc;
			}
			int A = e > dia ? e : dia; if (A < 0) A = 0;
			/* s[p] = max_{k <= p} (A[k] - (p - k) m): inclusive max-plus scan, first within the wavefront ... */
			int s = A;
			{   /* DPP scan: four steps inside the 16-lane rows, then lane 15 -> next row, lane 31 -> upper half */
				int o;
				o = (int)xl_row_shr_keep<1>((u32)NEG, (u32)s) - m; s = o > s ? o : s;
				o = (int)xl_row_shr_keep<2>((u32)NEG, (u32)s) - 2 * m; s = o > s ? o : s;
				o = (int)xl_row_shr_keep<4>((u32)NEG, (u32)s) - 4 * m; s = o > s ? o : s;
				o = (int)xl_row_shr_keep<8>((u32)NEG, (u32)s) - 8 * m; s = o > s ? o : s;
				o = (int)xl_row_bcast15_keep((u32)NEG, (u32)s) - ((lane & 15) + 1) * m; s = o > s ? o : s;
				o = (int)xl_row_bcast31_keep((u32)NEG, (u32)s) - ((lane & 31) + 1) * m; s = o > s ? o : s;
			}
			int P = NEG;                                            /* ... then across the wavefronts to the left: s of the cell before lane 0 */
			if (NW > 1) {
				if (lane == 63) lds_st32(lds, TX_T + 4u * (u32)wv, (u32)s);
				lds_barrier();
				{   /* every wavefront scans the (at most 16) wave totals itself: lane v holds T_v, decay 64 m per wavefront */
					int t = lane < NW ? (int)lds_ld32(lds, TX_T + 4u * (u32)lane) : NEG, o;
					o = (int)xl_row_shr_keep<1>((u32)NEG, (u32)t) - 64 * m; t = o > t ? o : t;
					o = (int)xl_row_shr_keep<2>((u32)NEG, (u32)t) - 128 * m; t = o > t ? o : t;
					if (NW > 4) {
						o = (int)xl_row_shr_keep<4>((u32)NEG, (u32)t) - 256 * m; t = o > t ? o : t;
						o = (int)xl_row_shr_keep<8>((u32)NEG, (u32)t) - 512 * m; t = o > t ? o : t;
					}
					if (wv > 0) P = (int)xl_readlane((u32)t, wv - 1);
				}
				const int sp = P - (lane + 1) * m;
				s = sp > s ? sp : s;
			}
			const int sleft = (int)xl_wave_shr1_keep((u32)P, (u32)s);   /* s[p-1]; nothing (NEG) left of the chunk's first cell */
			int F = sleft - gapO;
			{ const int ca = carryA - gapO - tid * m, cf = carryF - (tid + 1) * m; F = ca > F ? ca : F; F = cf > F ? cf : F; }
			const int e1 = e > 0 ? e : 0, f1 = F > 0 ? F : 0;
			const int gap = e1 > f1 ? e1 : f1;
			const int h = gap > dia ? gap : dia;
			/* direction of F: needs h and F of the cell to the left; hand-over of the chunk's last cell to the next chunk */
			const int last = ncell - c0 < NT - 1 ? ncell - c0 : NT - 1;     /* thread holding the chunk's last valid cell */
			int hleft, Fleft;
			if (NW > 1) {
				if (lane == 63) { lds_st32(lds, TX_H + 4u * (u32)wv, (u32)h); lds_st32(lds, TX_F + 4u * (u32)wv, (u32)F); }
				if (tid == last) { lds_st32(lds, TX_CARRY, (u32)F); lds_st32(lds, TX_CARRY + 4, (u32)A); lds_st32(lds, TX_CARRY + 8, (u32)h); }
				lds_barrier();
				int kh = carryH, kf = carryF;
				if (wv > 0) { kh = (int)lds_ld32(lds, TX_H + 4u * (u32)(wv - 1)); kf = (int)lds_ld32(lds, TX_F + 4u * (u32)(wv - 1)); }
				hleft = (int)xl_wave_shr1_keep((u32)kh, (u32)h); Fleft = (int)xl_wave_shr1_keep((u32)kf, (u32)F);
				carryF = (int)lds_ld32(lds, TX_CARRY); carryA = (int)lds_ld32(lds, TX_CARRY + 4); carryH = (int)lds_ld32(lds, TX_CARRY + 8);
			} else {
				hleft = (int)xl_wave_shr1_keep((u32)carryH, (u32)h); Fleft = (int)xl_wave_shr1_keep((u32)carryF, (u32)F);
				carryF = (int)xl_readlane((u32)F, last); carryA = (int)xl_readlane((u32)A, last); carryH = (int)xl_readlane((u32)h, last);
			}
			const int8_t df = (hleft - gapO) > (Fleft - gapE) ? 5 : 4;
			/* direction NIBBLE of the cell: bit 0 = E opened, bit 1 = F opened, bits 2-3 = H's source (0 diagonal, 1 E, 2 F); cells 2m and 2m + 1 of
			   a row share a byte (the even lane stores it: 32 contiguous bytes per wavefront) */
			const u32 nb = ok ? (u32)((de == 3 ? 1 : 0) | (df == 5 ? 2 : 0) | ((gap <= dia ? 0 : (e1 > f1 ? 1 : 2)) << 2)) : 0u;
			const u32 nb_hi = xl_wave_shl1_keep(0u, nb);
			if (ok) {
				R.steb(u, e); R.sthc(u, h);
				if (!(tid & 1)) line[(u - 1) >> 1] = (int8_t)(nb | (nb_hi << 4));
				if (h > lb) { lb = h; li = i; lj = j; }      /* rows and chunks come in row-major order: strict > keeps the first */
			}
		}
		if (NW == 1) trace_sync<L, NW>();    /* NW > 1: every thread is past the chunk's second barrier, and copies the cells it wrote itself */
		for (int u = 1 + tid; u <= ncell; u += NT) R.sthb(u, R.ldhc(u));
		trace_sync<L, NW>();
		if (!L) rd = rdn;
	}
	/* the scalar walk's best cell: highest h above the carried best; among equals the first in row-major order */
	if (lb <= tb.best) { lb = NEG; li = 0x7fffffff; lj = 0x7fffffff; }
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) {
		const int oh = wave_bcast(lb, lane ^ d), oi = wave_bcast(li, lane ^ d), oj = wave_bcast(lj, lane ^ d);
		if (oh > lb || (oh == lb && (oi < li || (oi == li && oj < lj)))) { lb = oh; li = oi; lj = oj; }
	}
	if (NW > 1) {
		if (lane == 0) { lds_st32(lds, TX_BEST + 12u * (u32)wv, (u32)lb); lds_st32(lds, TX_BEST + 12u * (u32)wv + 4, (u32)li); lds_st32(lds, TX_BEST + 12u * (u32)wv + 8, (u32)lj); }
		lds_barrier();
		for (int v = 0; v < NW; ++v) {
			const int oh = (int)lds_ld32(lds, TX_BEST + 12u * (u32)v), oi = (int)lds_ld32(lds, TX_BEST + 12u * (u32)v + 4), oj = (int)lds_ld32(lds, TX_BEST + 12u * (u32)v + 8);
			if (oh > lb || (oh == lb && (oi < li || (oi == li && oj < lj)))) { lb = oh; li = oi; lj = oj; }
		}
		lds_barrier();
	}
	if (lb > tb.best) { tb.best = lb; tb.i = li; tb.j = lj; }
}

/* ------------------------------------------------------------------------------------------------
 * trace_band_blocked: the same band fill with rows in LDS and TWO workgroup barriers per row instead of two per 64*NW cells
 * (NW = 1: fences).  Every thread owns CPT consecutive cells of the row: it evaluates E / diagonal / A for them, scans its own
 * cells serially, the threads' totals are scanned across the team (decay CPT*m per thread: DPP inside a wavefront, wave totals
 * through LDS -- barrier 1), then every thread finishes its cells (F, H, directions) and writes the row; the direction of F
 * in a thread's first cell needs h and F of the cell to its left, which the neighbour leaves in an LDS slot (barrier 2).
 * The rows of H ping-pong between two buffers (no copy), and what the reference forces at a row's start (h_b[edge] = 0,
 * e_b[edge] = -inf, src/ssw.c:636-639) is applied by whoever writes that index at the end of the row before.  Values,
 * directions and the row-major strict-> best cell are the scalar walk's: the long rows of wide bands (thousands of cells x 10^4
 * rows per alignment) were bound by barrier latency, not by arithmetic.
 * ------------------------------------------------------------------------------------------------ */
#define TX_SLOT 1536u      /* blocked form: h and F of every thread's last cell (two arrays of one dword per thread), placed after the fixed area */
SSW_HD u32 trace_lds_fixed_blocked(int nthreads) { return TX_SLOT + 8u * (u32)nthreads; }

/* CPT = cells per thread (trace_cpt_class: 1 for a single wavefront, 2, 4, and 8 or 12 for teams -- the smallest that covers the row;
   threads beyond the row idle, wavefronts beyond it only keep the barriers).  What a thread reads of the previous row -- H at
   up-1 .. up+CPT-1, E at up .. up+CPT-1, the target codes of its cells -- is requested in batches of four cells without a branch (cells
   past the row's end read the rows' padding), then the scores: two LDS round trips per batch.  (The first form of this function walked
   its cells under `if (k < cnt)`: hipcc made each an exec-masked block with its own three dependent round trips, 12 x 3 per row.)
   The rows are laid out with CPT + 1 entries between two threads' cells (trace_rowbytes): one LDS bank per lane. */
template <int NW, int CPT>
SSW_DEV void trace_band_blocked(unsigned char* lds, u32 oh0, u32 oh1, u32 oeb, const int8_t* ref, const int8_t* read, int refLen, int readLen,
                                int gapO, int gapE, int band_width, int n, int8_t* dir, u32 oring, u32 ring_mask, TraceBest& tb, int tid)
{
	constexpr int NT = 64 * NW;
	const int NEG = -1073741824;
	const int m = gapO < gapE ? gapO : gapE;
	const int lane = tid & 63, wv = tid >> 6;
	const int width = band_width * 2 + 3, width_d = band_width * 2 + 1;
	const int D = CPT * m;                            /* decay of the scan from one thread's last cell to the next one's */
	/* rows start as the reference's do before row 0 (H = 0; E = -inf makes `open` win: ssw.c:644-646), padding included; the target ring
	   starts as code 0, so that a cell past the row's end looks up a real score (nothing of it is kept) */
	constexpr bool PAD = CPT > 1;
	constexpr u32 BS = PAD ? CPT + 1 : 1;              /* entries between two threads' cells (see trace_rowbytes) */
	auto entry = [](int u) -> u32 { return u <= 0 ? 0u : PAD ? (u32)(u + 1 + (u - 1) / CPT) : (u32)u; };
	for (u32 j = (u32)tid; j <= entry(width + 12); j += NT) { lds_st32(lds, oh0 + 4u * j, 0u); lds_st32(lds, oh1 + 4u * j, 0u); lds_st32(lds, oeb + 4u * j, (u32)NEG); }
	for (u32 j = 4u * (u32)tid; j <= ring_mask; j += 4u * NT) lds_st32(lds, oring + j, 0u);
	int staged = 0;
	int lb = tb.best, li = 0, lj = 0;
	trace_sync<true, NW>();
	int rd = 0;
	const int u0 = tid * CPT + 1;
	for (int i = 0; i < readLen; ++i) {
		const int xi = i - band_width > 0 ? i - band_width : 0;
		const int xp = i - 1 - band_width > 0 ? i - 1 - band_width : 0;
		const int sft = xi - xp;
		const int beg = xi;
		const int end = i + band_width < refLen - 1 ? i + band_width : refLen - 1;
		const int ncell = end - beg + 1;
		const int endn = i + 1 + band_width < refLen - 1 ? i + 1 + band_width : refLen - 1;      /* the next row's forced index */
		const int edgen = endn + 1 < width - 1 ? endn + 1 : width - 1;
		const u32 hp = (i & 1) ? oh0 : oh1, hcur = (i & 1) ? oh1 : oh0;      /* previous row's H, this row's H */
		int8_t* line = dir + (int64_t)(band_width + 1) * i;      /* direction nibbles (trace_band): a row of 2 b + 1 cells in b + 1 bytes */
		if ((i & 63) == 0) {   /* target window: everything the next 64 rows can touch */
			int64_t hi = (int64_t)i + band_width + 65; if (hi > refLen) hi = refLen;
			for (int j = staged + tid; j < (int)hi; j += NT) lds_st8(lds, oring + ((u32)j & ring_mask), (u32)(unsigned char)ref[j]);
			if ((int)hi > staged) staged = (int)hi;
			/* the read's codes of these 64 rows too: a global load inside the row loop would wait for the rows' direction-byte stores (one
			   vmcnt for loads and stores on this target) -- a store round trip per row */
			if (tid < 64) lds_st8(lds, TX_READ + (u32)tid, (u32)(unsigned char)read[i + tid < readLen ? i + tid : readLen - 1]);
			trace_sync<true, NW>();
		}
		rd = lds_ld8s(lds, TX_READ + (u32)(i & 63));
		/* ---- this thread's cells: u0 .. u0 + cnt - 1 */
		const int cnt = u0 > ncell ? 0 : (ncell - u0 + 1 < CPT ? ncell - u0 + 1 : CPT);
		/* a wavefront whose 64 CPT cells all lie past the row's end (the class rounds the cells per thread up; a band's first and last rows are
		   short) only keeps the barriers: its issue slots go to the wavefronts of its SIMD that have cells */
		const bool has = NW == 1 || wv * 64 * CPT < ncell;
		int e[CPT], dia[CPT], sl[CPT];
		u32 deb = 0;                                   /* bit k: E of cell k was opened (direction 3) */
		int run = NEG;
		if (has) {
			/* Cells past the row's end (k >= cnt) are evaluated like the others -- on the rows' padding or, for a thread with no cell at all, on
			   the row's first entries -- with A forced to -inf: the scan then decays over them exactly as the total V wants, and nothing else of
			   them is stored.  Row 0 needs no case of its own: the rows start as H = 0 / E = -inf. */
			/* cell up0 - 1 + c of the previous row, c = 0 .. CPT + 1 (up0 = u0 + sft = tid CPT + 1 + sft): entry tid (CPT + 1) + off(c + sft), off(0) = 0,
			   off(x) = x + 1 for 1 <= x <= CPT, off(CPT + 1) = CPT + 3 -- uniform offsets on a per-thread base */
			const u32 base4 = cnt > 0 ? 4u * BS * (u32)tid : 0u;
			auto off4 = [](int x) -> u32 { return 4u * (u32)(x <= 0 ? 0 : !PAD ? x : x <= CPT ? x + 1 : x + 2); };
			constexpr int KB = CPT < 4 ? CPT : 4;       /* cells per batch of loads (register budget: 128 per thread in a team of 1024) */
			int hprev = (int)lds_ld32(lds, hp + base4 + off4(sft));
#pragma unroll
			for (int kb = 0; kb < CPT; kb += KB) {
				int hv[KB], ev[KB], cd[KB], sc[KB];
#pragma unroll
				for (int k = 0; k < KB; ++k) {
					const u32 x4 = base4 + off4(kb + k + 1 + sft);
					hv[k] = (int)lds_ld32(lds, hp + x4);
					ev[k] = (int)lds_ld32(lds, oeb + x4);
					cd[k] = lds_ld8s(lds, oring + ((u32)(beg + u0 + kb + k - 1) & ring_mask));
				}
#pragma unroll
				for (int k = 0; k < KB; ++k) sc[k] = lds_ld8s(lds, (u32)(cd[k] * n + rd));
#pragma unroll
				for (int k = 0; k < KB; ++k) {
					const int open = hv[k] - gapO;
					const int ext = ev[k] - gapE;
					const int ek = open > ext ? open : ext;
					const int dk = hprev + sc[k];
					hprev = hv[k];
					int A = ek > dk ? ek : dk; if (A < 0) A = 0;
					if (kb + k >= cnt) A = NEG;
					run = run - m > A ? run - m : A;               /* inclusive max-plus scan inside the thread */
					e[kb + k] = ek; dia[kb + k] = dk; sl[kb + k] = run;
					deb |= open > ext ? 1u << (kb + k) : 0u;
				}
				sched_fence();
			}
		}
		/* ---- scan of the threads' totals: V = S at the thread's last cell (threads without cells carry -inf: nothing follows them) */
		int V = cnt > 0 ? run : NEG;                   /* (a partial last thread: decayed to a full block's end by its cells past the row) */
		if (has) {
			int o;
			o = (int)xl_row_shr_keep<1>((u32)NEG, (u32)V) - D; V = o > V ? o : V;
			o = (int)xl_row_shr_keep<2>((u32)NEG, (u32)V) - 2 * D; V = o > V ? o : V;
			o = (int)xl_row_shr_keep<4>((u32)NEG, (u32)V) - 4 * D; V = o > V ? o : V;
			o = (int)xl_row_shr_keep<8>((u32)NEG, (u32)V) - 8 * D; V = o > V ? o : V;
			o = (int)xl_row_bcast15_keep((u32)NEG, (u32)V) - ((lane & 15) + 1) * D; V = o > V ? o : V;
			o = (int)xl_row_bcast31_keep((u32)NEG, (u32)V) - ((lane & 31) + 1) * D; V = o > V ? o : V;
		}
		int Pw = wv == 0 ? 0 : NEG;                   /* S at the cell before this wavefront's first one; cell 0 of the row holds h_c[0] = 0 */
		if (NW > 1) {
			if (lane == 63) lds_st32(lds, TX_T + 4u * (u32)wv, (u32)V);
			lds_barrier();                                /* barrier 1: wave totals; every phase-1 read of the previous row is done */
			if (has) {
			int t = lane < NW ? (int)lds_ld32(lds, TX_T + 4u * (u32)lane) : NEG, o;
			if (lane == 0) { const int z = 0 - 64 * D; t = z > t ? z : t; }      /* the row's cell 0 decayed to the end of wavefront 0 */
			o = (int)xl_row_shr_keep<1>((u32)NEG, (u32)t) - 64 * D; t = o > t ? o : t;
			o = (int)xl_row_shr_keep<2>((u32)NEG, (u32)t) - 128 * D; t = o > t ? o : t;
			if (NW > 4) {
				o = (int)xl_row_shr_keep<4>((u32)NEG, (u32)t) - 256 * D; t = o > t ? o : t;
				o = (int)xl_row_shr_keep<8>((u32)NEG, (u32)t) - 512 * D; t = o > t ? o : t;
			}
			if (wv > 0) Pw = (int)xl_readlane((u32)t, wv - 1);
			}
		} else wave_lds_fence();                          /* (one wavefront: its reads of the previous row are issued before the writes below) */
		/* ---- finish the cells */
		int hl = 0, Fl = NEG;                          /* h and F of the cell to the left (cell 0: h_c[0] = 0, no F); a thread's first cell learns them after barrier 2 */
		u32 byte_first = 0;
		unsigned long long nibbles = 0;                /* direction nibbles of this thread's cells of the row (cell k at bits 4k..4k+3) */
		if (has) {
		{ const int sp = Pw - (lane + 1) * D; V = sp > V ? sp : V; }
		const int P = (int)xl_wave_shr1_keep((u32)Pw, (u32)V);      /* S at the cell before this thread's first one */
		int Sprev = P;
		int hm = 0;                                    /* highest h of this thread's cells in this row */
#pragma unroll
		for (int k = 0; k < CPT; ++k) {
			const int F = Sprev - gapO;
			const int sfin = P - (k + 1) * m > sl[k] ? P - (k + 1) * m : sl[k];
			Sprev = sfin;
			const int e1 = e[k] > 0 ? e[k] : 0, f1 = F > 0 ? F : 0;
			const int gap = e1 > f1 ? e1 : f1;
			const int h = gap > dia[k] ? gap : dia[k];
			const int de3 = (int)((deb >> k) & 1u);
			const u32 hs = gap <= dia[k] ? 0u : (e1 > f1 ? 1u : 2u);      /* H's source: diagonal / E / F */
			if (k == 0) byte_first = (u32)de3 | (hs << 2);                 /* (whether F was opened into the first cell is known after barrier 2) */
			else {
				const int df5 = (hl - gapO) > (Fl - gapE) ? 1 : 0;
				if (k < cnt) nibbles |= (unsigned long long)((u32)de3 | ((u32)df5 << 1) | (hs << 2)) << (4 * k);
			}
			if (k < cnt) {      /* (stores and moves only: nothing in here waits) */
				lds_st32(lds, oeb + 4u * (BS * (u32)tid + (u32)k + (PAD ? 2u : 1u)), (u32)e[k]);      /* = entry(u) */
				lds_st32(lds, hcur + 4u * (BS * (u32)tid + (u32)k + (PAD ? 2u : 1u)), (u32)h);
				hm = h > hm ? h : hm;
				hl = h; Fl = F;
			}
		}
		if (hm > lb) {      /* a new best cell (rare: bests of narrower bands carry over): the first of this thread's cells that holds it */
			int kk = -1;
			for (int k = 0; k < cnt; ++k) if (kk < 0 && (int)lds_ld32(lds, hcur + 4u * (BS * (u32)tid + (u32)k + (PAD ? 2u : 1u))) == hm) kk = k;
			lb = hm; li = i; lj = beg + u0 + kk - 1;
		}
		/* the row as the next one will read it: its forced index holds 0 / -inf whatever was computed there (after the lookup above) */
		if (edgen >= u0 && edgen < u0 + cnt) { lds_st32(lds, oeb + 4u * entry(edgen), (u32)NEG); lds_st32(lds, hcur + 4u * entry(edgen), 0u); }
		}      /* has */
		if (NW > 1 && cnt > 0) { lds_st32(lds, TX_SLOT + 4u * (u32)tid, (u32)hl); lds_st32(lds, TX_SLOT + 4u * (u32)(NT + tid), (u32)Fl); }
		if (tid == 0 && edgen > ncell) { lds_st32(lds, oeb + 4u * entry(edgen), (u32)NEG); lds_st32(lds, hcur + 4u * entry(edgen), 0u); }
		trace_sync<true, NW>();                         /* barrier 2: the row is written; neighbours' last cells are in the slots */
		int hleft = 0, Fleft = NEG;
		if (NW == 1) { hleft = (int)xl_wave_shr1_keep(0u, (u32)hl); Fleft = (int)xl_wave_shr1_keep((u32)NEG, (u32)Fl); }      /* (one wavefront: the neighbour's last cell by DPP) */
		if (cnt > 0) {
			if (NW > 1 && tid > 0) { hleft = (int)lds_ld32(lds, TX_SLOT + 4u * (u32)(tid - 1)); Fleft = (int)lds_ld32(lds, TX_SLOT + 4u * (u32)(NT + tid - 1)); }
			const int df5 = (hleft - gapO) > (Fleft - gapE) ? 1 : 0;
			nibbles |= (unsigned long long)(byte_first | ((u32)df5 << 1));
		}
		/* the thread's direction nibbles of this row, two cells per byte: its first cell has an even index in the row (tid x CPT, CPT even);
		   one cell per lane (CPT = 1): the even lane takes its neighbour's nibble */
		if (CPT == 1) {
			const u32 nb = cnt > 0 ? (u32)nibbles : 0u, nb_hi = xl_wave_shl1_keep(0u, nb);
			if (cnt > 0 && !(tid & 1)) line[(u0 - 1) >> 1] = (int8_t)(nb | (nb_hi << 4));
		} else {
#pragma unroll
			for (int kb = 0; kb < CPT / 2; ++kb)
				if (2 * kb < cnt) line[((u0 - 1) >> 1) + kb] = (int8_t)((nibbles >> (8 * kb)) & 0xffu);
		}
	}
	/* the scalar walk's best cell: highest h above the carried best; among equals the first in row-major order */
	if (lb <= tb.best) { lb = NEG; li = 0x7fffffff; lj = 0x7fffffff; }
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) {
		const int oh = wave_bcast(lb, lane ^ d), oi = wave_bcast(li, lane ^ d), oj = wave_bcast(lj, lane ^ d);
		if (oh > lb || (oh == lb && (oi < li || (oi == li && oj < lj)))) { lb = oh; li = oi; lj = oj; }
	}
	if (NW > 1) {
		lds_barrier();
		if (lane == 0) { lds_st32(lds, TX_BEST + 12u * (u32)wv, (u32)lb); lds_st32(lds, TX_BEST + 12u * (u32)wv + 4, (u32)li); lds_st32(lds, TX_BEST + 12u * (u32)wv + 8, (u32)lj); }
		lds_barrier();
		for (int v = 0; v < NW; ++v) {
			const int oh = (int)lds_ld32(lds, TX_BEST + 12u * (u32)v), oi = (int)lds_ld32(lds, TX_BEST + 12u * (u32)v + 4), oj = (int)lds_ld32(lds, TX_BEST + 12u * (u32)v + 8);
			if (oh > lb || (oh == lb && (oi < li || (oi == li && oj < lj)))) { lb = oh; li = oi; lj = oj; }
		}
		lds_barrier();
	} else wave_lds_fence();
	if (lb > tb.best) { tb.best = lb; tb.i = li; tb.j = lj; }
}

/* value of thread 0 in every thread of the team */
template <int NW> SSW_DEV int team_bcast0(unsigned char* lds, int v, int tid)
{
	if (NW == 1) return wave_bcast(v, 0);
	if (tid == 0) lds_st32(lds, TX_BCAST, (u32)v);
	lds_barrier();
	const int r = (int)lds_ld32(lds, TX_BCAST);
	lds_barrier();
	return r;
}

/* returns the number of CIGAR operations, -1 (traceback failed), or -2: *need holds the scratch bytes wanted and
   *band_io / tb the state to resume from (the narrower bands need not be walked again: they are deterministic) */
template <int NW>
SSW_DEV int trace_team(const int8_t* ref, const int8_t* read, int refLen, int readLen, int score, int gapO, int gapE,
                       int* band_io, TraceBest& tb, const int8_t* mat, int n, unsigned char* scratch, int64_t cap,
                       unsigned char* lds, int64_t lds_cap, u32* cig, int cigcap, int64_t* need, int tid, bool trace_unblocked)
{
	const int len = refLen > readLen ? refLen : readLen;
	int band_width = *band_io, width_d;
	int8_t* dir;
	do {
		const int width = band_width * 2 + 3; width_d = band_width * 2 + 1;
		const int64_t rowbytes = trace_rowbytes(band_width, 64 * NW);
		const int64_t want = 3 * rowbytes + (int64_t)(band_width + 1) * readLen + 16;      /* direction nibbles: b + 1 bytes per row of 2 b + 1 cells */
		if (want > cap) { *need = want; *band_io = band_width; return -2; }
		dir = (int8_t*)(scratch + 3 * rowbytes);
		/* (8 and 12 cells per thread for the teams, not for single wavefronts: their kernel keeps five wavefronts per SIMD) */
		const int cpt = trace_cpt_class(width_d, 64 * NW);
		if (trace_lds_need(band_width, 64 * NW) <= lds_cap && cpt > 0 && !trace_unblocked) {
			const u32 oh0 = trace_lds_fixed_blocked(64 * NW), oh1 = oh0 + (u32)rowbytes, oeb = oh1 + (u32)rowbytes, oring = oeb + (u32)rowbytes;
			const u32 rmask = trace_ring_size(band_width, 64 * NW) - 1;
			if (cpt <= 1) { if constexpr (NW == 1) trace_band_blocked<NW, 1>(lds, oh0, oh1, oeb, ref, read, refLen, readLen, gapO, gapE, band_width, n, dir, oring, rmask, tb, tid); }
			else if (cpt <= 2) trace_band_blocked<NW, 2>(lds, oh0, oh1, oeb, ref, read, refLen, readLen, gapO, gapE, band_width, n, dir, oring, rmask, tb, tid);
			else if (cpt <= 4) trace_band_blocked<NW, 4>(lds, oh0, oh1, oeb, ref, read, refLen, readLen, gapO, gapE, band_width, n, dir, oring, rmask, tb, tid);
			else if constexpr (NW >= 4) {
				if (cpt <= 8) trace_band_blocked<NW, 8>(lds, oh0, oh1, oeb, ref, read, refLen, readLen, gapO, gapE, band_width, n, dir, oring, rmask, tb, tid);
				else trace_band_blocked<NW, 12>(lds, oh0, oh1, oeb, ref, read, refLen, readLen, gapO, gapE, band_width, n, dir, oring, rmask, tb, tid);
			}
		} else
		if (trace_lds_need(band_width, 64 * NW) <= lds_cap) {
			TraceRows<true> R; R.lds = lds; R.ohb = TRACE_LDS_FIXED; R.oeb = TRACE_LDS_FIXED + (u32)rowbytes; R.ohc = TRACE_LDS_FIXED + 2 * (u32)rowbytes;
			R.hb = R.eb = R.hc = 0;
			const u32 oring = TRACE_LDS_FIXED + 3 * (u32)rowbytes;
			trace_band<true, NW>(R, ref, read, refLen, readLen, gapO, gapE, band_width, mat, n, dir, oring, trace_ring_size(band_width, 64 * NW) - 1, tb, tid);
		} else {
			TraceRows<false> R; R.lds = lds; R.ohb = R.oeb = R.ohc = 0;
			R.hb = (int*)scratch; R.eb = (int*)(scratch + rowbytes); R.hc = (int*)(scratch + 2 * rowbytes);
			trace_band<false, NW>(R, ref, read, refLen, readLen, gapO, gapE, band_width, mat, n, dir, 0, 0, tb, tid);
		}
		band_width *= 2;
	} while (tb.best < score && band_width <= len);
	band_width /= 2;
	const int best_i = tb.i, best_j = tb.j;

	if (NW > 1) { wg_fence(); __syncthreads(); } else wg_fence();      /* (workgroup scope: writers and readers are one workgroup on one compute unit; an agent-scope fence writes L2 back, 10^5 times per batch) */
	int nops = 0;
	if (tid < 64) {
		/* The walk back (src/ssw.c:682-762) by the team's first wavefront.  It used to be thread 0's loop: 10^4 dependent one-byte loads for a
		   10-kb read, a memory latency each -- tens of milliseconds per alignment, after the passes, with the other 1023 threads waiting.  Now all 64 lanes follow
		   the same walk and the bytes come in batches: lane t fetches twelve bytes of row (base - t) around the band index the path has if
		   it keeps to its diagonal, a step takes its byte from the lane that holds its row with one cross-lane move; a path that drifts
		   out of a row's window (a long run of gap steps) or leaves the 64 rows of the batch starts a new batch where it stands. */
		int run = 0, state = 2, cur = 0, prev = 0, i = best_i, j = best_j, failed = 0, wbase = -1, jbase = 0;
		u32 ww0 = 0, ww1 = 0, ww2 = 0;
		const int64_t RB = band_width + 1;      /* bytes per row of direction nibbles */
		(void)width_d;
		while (i >= 0 && j > 0) {
			const int x0 = i - band_width > 0 ? i - band_width : 0;
			const int64_t pos = RB * i + ((j - x0) >> 1);      /* the byte that holds the cell's nibble */
			int t = wbase - i, o;
			{
				int64_t P = (RB * i + ((((jbase - t) - x0) >> 1)) - 4) & ~(int64_t)3; if (P < 0) P = 0;
				const int64_t oo = pos - P;
				o = oo < 0 || oo >= 12 ? -1 : (int)oo;
			}
			if (t < 0 || t >= 64 || o < 0) {      /* (uniform: every lane holds the same walk state) */
				wbase = i; jbase = j; t = 0;
				const int r = wbase - tid, jr = jbase - tid, xr = r - band_width > 0 ? r - band_width : 0;
				int64_t P = (RB * r + ((jr - xr) >> 1) - 4) & ~(int64_t)3; if (P < 0) P = 0;
				if (r >= 0) { const u32* src = (const u32*)(dir + P); ww0 = src[0]; ww1 = src[1]; ww2 = src[2]; }
				int64_t P0 = (pos - 4) & ~(int64_t)3; if (P0 < 0) P0 = 0;
				o = (int)(pos - P0);
			}
			const int sel = o >> 2;
			const u32 word = xl_shfl(sel <= 0 ? ww0 : sel == 1 ? ww1 : ww2, t);
			const int pk = (int)((word >> (8 * (o & 3) + 4 * ((j - x0) & 1))) & 0xfu);
			const int hs = pk >> 2;
			const int d = state == 0 ? 2 + (pk & 1) : state == 1 ? 4 + ((pk >> 1) & 1) : hs == 0 ? 1 : hs == 1 ? 2 + (pk & 1) : hs == 2 ? 4 + ((pk >> 1) & 1) : 0;
			if (d == 1) { --i; --j; state = 2; cur = 0; }
			else if (d == 2) { --i; state = 0; cur = 1; }
			else if (d == 3) { --i; state = 2; cur = 1; }
			else if (d == 4) { --j; state = 1; cur = 2; }
			else if (d == 5) { --j; state = 2; cur = 2; }
			else { failed = 1; break; }
			if (cur == prev) ++run;
			else { if (tid == 0 && nops < cigcap) cig[nops] = ((u32)run << 4) | (u32)prev; ++nops; prev = cur; run = 1; }
		}
		if (failed) nops = -1;
		else {
			if (cur == 0) { if (tid == 0 && nops < cigcap) cig[nops] = ((u32)(run + 1) << 4); ++nops; }
			else {
				if (tid == 0 && nops < cigcap) cig[nops] = ((u32)run << 4) | (u32)cur; ++nops;
				if (tid == 0 && nops < cigcap) cig[nops] = (1u << 4); ++nops;
			}
			if (nops > cigcap) { *need = -(int64_t)nops; nops = -2; }
			else {
				wg_fence();
				for (int x = tid; x < nops / 2; x += 64) { const u32 tt = cig[x]; cig[x] = cig[nops - 1 - x]; cig[nops - 1 - x] = tt; }
			}
		}
	}
	wg_fence();
	return team_bcast0<NW>(lds, nops, tid);
}

/* one team of NW wavefronts (one workgroup) per alignment; same contract as k_trace.  a.resume[q] = {band, best, best_i,
   best_j, stage} carries an alignment that ran out of scratch to the next negotiation round. */
template <int NW>
/* (register budget: a single wavefront per alignment keeps five wavefronts per SIMD -- at 64 registers the row loop spills, at 128 config 4 loses 10 ms;
   the teams have 128 each: four wavefronts per SIMD is a team of 1024 threads on one compute unit) */
__global__ void __launch_bounds__(64 * NW) SSW_WAVES_PER_EU(NW == 1 ? 5 : 4, 8) k_trace_wave(ssw_trace_args a)
{
	SSW_DYN_LDS(lds);
	const int job = (int)blockIdx.x, tid = (int)threadIdx.x;
	const int q = a.qlist[job];
	ssw_dres r = a.res[q];
	if (tid == 0) a.need[job] = 0;
	if (!r.want_cigar || r.status != 0) return;
	const int8_t* ref = vm_target(a.vm, a.tgt, q) + r.ref_begin1;
	const int8_t* read = a.qcodes + a.qoff[vm_query(a.vm, q)] + r.read_begin1;
	const int refLen = r.ref_end1 - r.ref_begin1 + 1, readLen = r.read_end1 - r.read_begin1 + 1;
	const int d = refLen - readLen;
	const int band0 = (d < 0 ? -d : d) + 1;
	int band = band0;
	const int full = refLen > readLen ? refLen : readLen;
	u32* cig = a.cigar + (int64_t)q * a.cigar_stride;
	unsigned char* scratch = a.soff ? a.scratch + a.soff[job] : a.scratch + (int64_t)job * a.scratch_stride;
	const int64_t scap = a.soff ? a.soff[job + 1] - a.soff[job] : a.scratch_stride;
	for (int k = tid; k < a.n * a.n && k < 1024; k += 64 * NW) lds_st8(lds, (u32)k, (u32)(unsigned char)a.mat[k]);
	trace_sync<true, NW>();
	const int64_t lds_cap = a.n * a.n <= 1024 ? (int64_t)a.lds_bytes : 0;
	TraceBest tb; tb.best = 0; tb.i = 0; tb.j = 0;
	int stage = 0;                         /* 1: the single full-band retry after a CIGAR that does not re-score (ssw.c:1000-1010) */
	int32_t* rs = a.resume + (int64_t)q * 8;
	if (rs[0] > 0) { band = rs[0]; tb.best = rs[1]; tb.i = rs[2]; tb.j = rs[3]; stage = rs[4]; }
	int nops;
	for (;;) {
		int64_t need = 0;
		int bio = band;
		nops = trace_team<NW>(ref, read, refLen, readLen, r.score1, a.gapO, a.gapE, &bio, tb, a.mat, a.n, scratch, scap,
		                      lds, lds_cap, cig, (int)a.cigar_stride, &need, tid, a.unblocked != 0);
		if (nops == -2) {
			const int64_t need0 = ((int64_t)team_bcast0<NW>(lds, (int)(need >> 32), tid) << 32) | (u32)team_bcast0<NW>(lds, (int)(need & 0xffffffff), tid);
			if (tid == 0) {
				a.need[job] = need0 > 0 ? (int)((need0 + 4095) >> 12) : -1;
				a.need[a.nq + job] = bio;
				rs[0] = bio; rs[1] = tb.best; rs[2] = tb.i; rs[3] = tb.j; rs[4] = stage;
			}
			return;
		}
		if (nops < 0) break;
		int sc = 0;
		if (tid < 64) sc = cigar_score_wave(cig, nops, ref, read, lds, a.n, a.gapO, a.gapE, tid);      /* (the matrix is in the first KiB of LDS) */
		sc = team_bcast0<NW>(lds, sc, tid);
		if (sc == r.score1) break;
		if (stage || band0 >= full) { nops = -1; break; }
		band = full; stage = 1;
		tb.best = 0; tb.i = 0; tb.j = 0;
	}
	if (tid == 0) {
		if (nops < 0) { a.res[q].flag = 1; a.res[q].cigarLen = 0; }
		else { a.res[q].cigarLen = nops; a.res[q].cigar_off = (int64_t)q * a.cigar_stride; }
	}
}

/* ------------------------------------------------------------------------------------------------
 * k_trace_diag<TEAM>: the same banded_sw for NARROW bands, SEVERAL alignments per wavefront (round 5).
 *
 * What the row kernels above cost a narrow band: a row of 9 .. 31 cells occupies one wavefront for ~110 vector instructions whatever its
 * width, and 72 % of config 4's alignments (all of a protein search's survivors) never leave bands <= 15.  Here a TEAM of 16 (32) lanes
 * owns one alignment -- four (two) alignments per wavefront -- and walks the band by ANTI-DIAGONALS, so that no scan is needed:
 * lane k of a team owns the two band diagonals d = 2k - w and 2k + 1 - w (d = column - row, |d| <= w), and in "pair" s it evaluates the
 * cells A = (i, i + 2k - w) and then B = (i, i + 2k + 1 - w) of row i = s - k.  Every neighbour is then already there:
 *     A: left  = lane k-1's B of the pair before (one DPP move each for h and F), up = this lane's B of the pair before, diagonal = its A;
 *     B: left  = A of this pair, diagonal = this lane's B of the pair before, up = lane k+1's A of THIS pair (one DPP move each for h and E).
 * A pass over a band of width w takes readLen + w pairs; cells outside the matrix or the band hand on h = 0, E = F = -inf, which is what
 * the reference's row arrays hold at those places (h_b[0], h_c[0], the forced index `edge` of src/ssw.c:636-639 one past a row's end).
 * The one place where `edge` lands on a REAL cell -- target coordinates applied as band coordinates: rows 1 .. w + 1 of a band that
 * spans the whole target (refLen <= 2w + 2) see the last column of the row above as 0 / -inf -- is reproduced explicitly (quirk below).
 * Values, direction bytes (same packing and [row][band index] layout as k_trace_wave), the row-major strict-> best cell (per lane in
 * visiting order = row-major within its two diagonals, then max / smallest (row, column) across the team), the doubling loop with its
 * persistent best, the walk, the re-score and the single full-band retry are the reference's.  A team whose band outgrows it
 * (w > TEAM - 1) hands over exactly like an alignment that ran out of scratch: need / band / resume state, and a row kernel continues.
 *
 * The teams of a wavefront are in different phases at the same time (passes of different lengths, the serial walk back, the re-score):
 * one flat loop, every iteration = one pair of the teams that are in a pass + one walk step / a few re-score steps of the others; all
 * cross-lane moves are executed by all lanes in uniform control flow.
 * ------------------------------------------------------------------------------------------------ */
template <int TEAM> SSW_DEV u32 team_from_below(u32 keep, u32 v, int k)     /* value of lane k - 1 of the team; lane 0 keeps */
{
	if (TEAM == 16) return xl_row_shr1_keep(keep, v);
	const u32 r = xl_wave_shr1_keep(keep, v);
	return k == 0 ? keep : r;
}
template <int TEAM> SSW_DEV u32 team_from_above(u32 keep, u32 v, int k)     /* value of lane k + 1 of the team; the last lane keeps */
{
	if (TEAM == 16) return xl_row_shl1_keep(keep, v);
	const u32 r = xl_wave_shl1_keep(keep, v);
	return k == TEAM - 1 ? keep : r;
}

#define TD_PASS 1
#define TD_WALK 2
#define TD_SCORE 3
#define TD_RETRY 4
#define TD_REVERSE 5
#define TD_DONE 0

template <int TEAM>
__global__ void __launch_bounds__(64) SSW_WAVES_PER_EU(4, 8) k_trace_diag(ssw_trace_args a)
{
	SSW_DYN_LDS(lds);
	constexpr int TPW = 64 / TEAM;
	const int NEG = -1073741824;
	const int lane = (int)threadIdx.x, k = lane % TEAM, lead = lane - k;
	const int job = (int)blockIdx.x * TPW + lane / TEAM;
	const int n = a.n, gapO = a.gapO, gapE = a.gapE;
	for (int x = lane; x < n * n && x < 1024; x += 64) lds_st8(lds, (u32)x, (u32)(unsigned char)a.mat[x]);
	/* The codes a pair needs come from two 256-byte LDS rings per team (read rows, target columns), refilled 64 entries at a time.  A global
	   load inside the pair loop would make every pair wait for the direction-byte STORES of the pair before it: loads and stores share one
	   counter (vmcnt) on this target and return out of order with each other, so a wait for a load is a wait for everything. */
	const u32 ring_rd = 1024u + 512u * (u32)(lane / TEAM), ring_rf = ring_rd + 256u;
	for (u32 x = 4u * (u32)lane; x < 512u * TPW; x += 256u) lds_st32(lds, 1024u + x, 0u);
	wave_lds_fence();
	constexpr int FILL = 64 / TEAM;      /* ring entries a lane fetches per refill of 64 */
	auto refill = [&](int base, const int8_t* rdp, int rdl, const int8_t* rfp, int rfl) {      /* entries [base, base + 64) of both rings */
		u32 wr = 0, wf = 0;
#pragma unroll
		for (int b = 0; b < FILL; ++b) {
			const int x = base + FILL * k + b;
			wr |= (u32)(unsigned char)rdp[x < 0 ? 0 : x >= rdl ? rdl - 1 : x] << (8 * b);
			wf |= (u32)(unsigned char)rfp[x < 0 ? 0 : x >= rfl ? rfl - 1 : x] << (8 * b);
		}
		const u32 at = (u32)(base + FILL * k) & 255u;
		if (FILL == 4) { lds_st32(lds, ring_rd + at, wr); lds_st32(lds, ring_rf + at, wf); }
		else { lds_st16(lds, ring_rd + at, wr); lds_st16(lds, ring_rf + at, wf); }
	};

	/* ---- the team's alignment */
	int state = TD_DONE, q = 0, refLen = 1, readLen = 1, score = 0, band0 = 1, full = 1, stage = 0;
	const int8_t* ref = a.tgt; const int8_t* read = a.qcodes;
	u32* cig = a.cigar; int8_t* dir = (int8_t*)a.scratch; int64_t scap = 0; int32_t* rs = a.resume;
	int tb_best = 0, tb_i = 0, tb_j = 0;
	int w = 1;
	if (job < a.nq) {
		q = a.qlist[job];
		const ssw_dres r = a.res[q];
		if (k == 0) a.need[job] = 0;
		if (r.want_cigar && r.status == 0) {
			ref = vm_target(a.vm, a.tgt, q) + r.ref_begin1;
			read = a.qcodes + a.qoff[vm_query(a.vm, q)] + r.read_begin1;
			refLen = r.ref_end1 - r.ref_begin1 + 1; readLen = r.read_end1 - r.read_begin1 + 1;
			score = r.score1;
			const int d = refLen - readLen;
			band0 = (d < 0 ? -d : d) + 1; full = refLen > readLen ? refLen : readLen;
			cig = a.cigar + (int64_t)q * a.cigar_stride;
			dir = (int8_t*)(a.soff ? a.scratch + a.soff[job] : a.scratch + (int64_t)job * a.scratch_stride);
			scap = a.soff ? a.soff[job + 1] - a.soff[job] : a.scratch_stride;
			rs = a.resume + (int64_t)q * 8;
			w = band0;
			if (rs[0] > 0) { w = rs[0]; tb_best = rs[1]; tb_i = rs[2]; tb_j = rs[3]; stage = rs[4]; }
			state = TD_RETRY;      /* (enters the loop through the common "start a pass at band w" code) */
		}
	}
	const int cigcap = (int)a.cigar_stride;
	/* pass state */
	int s = 0, S = 0, width_d = 3;
	int HAp = 0, HBp = 0, EBp = NEG, FBp = NEG;
	int lb = 0, li = 0, lj = 0;
	/* walk / re-score state (the same in every lane of the team), and this lane's twelve direction bytes of the current batch */
	int wi = 0, wj = 0, wst = 2, cur = 0, prev = 0, run = 0, nops = 0, wbase = -1, jbase = 0;
	u32 ww0 = 0, ww1 = 0, ww2 = 0;
	int oi = 0, rp = 0, qp = 0, sc = 0;
	bool fresh = state == TD_RETRY;      /* first entry: the band is the resumed / first one, not the full-band retry */

	while (wave_any(state != TD_DONE)) {
		/* ---- start a pass at band w (first entry, next doubling, full-band retry): team-uniform */
		if (wave_any(state == TD_RETRY)) {
			if (state == TD_RETRY) {
				if (!fresh) { stage = 1; w = full; tb_best = 0; tb_i = 0; tb_j = 0; }
				fresh = false;
				const int64_t want = (int64_t)(2 * w + 1) * readLen + 16;
				if (w > TEAM - 1 || want > scap) {      /* hand over to a row kernel: the band that did not fit and what it needs there (three padded rows + directions) */
					if (k == 0) {
						const int64_t need = 3 * ((int64_t)(2 * w + 16) * 6 + 64) + (int64_t)(2 * w + 1) * readLen + 64;
						a.need[job] = (int)((need + 4095) >> 12); a.need[a.nq + job] = w;
						rs[0] = w; rs[1] = tb_best; rs[2] = tb_i; rs[3] = tb_j; rs[4] = stage;
					}
					state = TD_DONE;
				} else {
					width_d = 2 * w + 1; s = 0; S = readLen + w;
					HAp = 0; HBp = 0; EBp = NEG; FBp = NEG;
					lb = tb_best; li = 0; lj = 0;
					refill(0, read, readLen, ref, refLen); refill(64, read, readLen, ref, refLen); refill(128, read, readLen, ref, refLen);
					state = TD_PASS;
				}
			}
			wave_lds_fence();
		}
		/* ---- one pair of every team that is in a pass */
		if (wave_any(state == TD_PASS)) {
			const bool P = state == TD_PASS;
			if (wave_any(P && s > 0 && (s & 63) == 0)) {      /* entries [s + 128, s + 192): needed from pair s + 113 on */
				if (P && s > 0 && (s & 63) == 0) refill(s + 128, read, readLen, ref, refLen);
				wave_lds_fence();
			}
			const int i = s - k, jA = s + k - w, jB = jA + 1;
			const int cr = lds_ld8s(lds, ring_rd + ((u32)i & 255u)), cA = lds_ld8s(lds, ring_rf + ((u32)jA & 255u)), cB = lds_ld8s(lds, ring_rf + ((u32)jB & 255u));
			const bool rowok = P && i >= 0 && i < readLen;
			const bool vA = rowok && k <= w && jA >= 0 && jA < refLen;
			const bool vB = rowok && k < w && jB >= 0 && jB < refLen;
			/* the reference's forced index on a real cell (see above): rows 1 .. w + 1, last target column, band spanning the target */
			const bool qrow = i >= 1 && i <= w + 1 && refLen <= 2 * w + 2;
			const int x0 = i - w > 0 ? i - w : 0;
			int8_t* line = dir + (int64_t)width_d * i - x0;
			/* cell A */
			const int hl = (int)team_from_below<TEAM>(0u, (u32)HBp, k), fl = (int)team_from_below<TEAM>((u32)NEG, (u32)FBp, k);
			int hA, eA, fA;
			{
				const bool qk = qrow && jA == refLen - 1;
				const int hu = qk ? 0 : HBp, eu = qk ? NEG : EBp;
				const int open = hu - gapO, ext = eu - gapE;
				const int e = open > ext ? open : ext, de3 = open > ext ? 1 : 0;
				const int fo = hl - gapO, fe = fl - gapE;
				const int f = fo > fe ? fo : fe, df5 = fo > fe ? 1 : 0;
				const int e1 = e > 0 ? e : 0, f1 = f > 0 ? f : 0, gap = e1 > f1 ? e1 : f1;
				const int dia = HAp + lds_ld8s(lds, (u32)(cA * n + cr));
				const int h = gap > dia ? gap : dia;
				const int dh = gap <= dia ? 1 : (e1 > f1 ? 2 + de3 : 4 + df5);
				if (vA) {
					line[jA] = (int8_t)(de3 | (df5 << 1) | (dh << 2));
					if (h > lb) { lb = h; li = i; lj = jA; }
				}
				hA = vA ? h : 0; eA = vA ? e : NEG; fA = vA ? f : NEG;
			}
			/* cell B */
			const int hu2 = (int)team_from_above<TEAM>(0u, (u32)hA, k), eu2 = (int)team_from_above<TEAM>((u32)NEG, (u32)eA, k);
			int hB, eB, fB;
			{
				const bool qk = qrow && jB == refLen - 1;
				const int hu = qk ? 0 : hu2, eu = qk ? NEG : eu2;
				const int open = hu - gapO, ext = eu - gapE;
				const int e = open > ext ? open : ext, de3 = open > ext ? 1 : 0;
				const int fo = hA - gapO, fe = fA - gapE;
				const int f = fo > fe ? fo : fe, df5 = fo > fe ? 1 : 0;
				const int e1 = e > 0 ? e : 0, f1 = f > 0 ? f : 0, gap = e1 > f1 ? e1 : f1;
				const int dia = HBp + lds_ld8s(lds, (u32)(cB * n + cr));
				const int h = gap > dia ? gap : dia;
				const int dh = gap <= dia ? 1 : (e1 > f1 ? 2 + de3 : 4 + df5);
				if (vB) {
					line[jB] = (int8_t)(de3 | (df5 << 1) | (dh << 2));
					if (h > lb) { lb = h; li = i; lj = jB; }
				}
				hB = vB ? h : 0; eB = vB ? e : NEG; fB = vB ? f : NEG;
			}
			if (P) { HAp = hA; HBp = hB; EBp = eB; FBp = fB; ++s; }
			/* ---- end of a pass: the team's best cell, then the reference's loop condition (src/ssw.c:679) */
			if (wave_any(P && s == S)) {
				const bool E = P && s == S;
				int rb = lb, ri = li, rj = lj;
				if (rb <= tb_best) { rb = NEG; ri = 0x7fffffff; rj = 0x7fffffff; }
#pragma unroll
				for (int d = 1; d < TEAM; d <<= 1) {
					const int oh = (int)xl_shfl((u32)rb, lane ^ d), oi2 = (int)xl_shfl((u32)ri, lane ^ d), oj2 = (int)xl_shfl((u32)rj, lane ^ d);
					if (oh > rb || (oh == rb && (oi2 < ri || (oi2 == ri && oj2 < rj)))) { rb = oh; ri = oi2; rj = oj2; }
				}
				wg_fence();      /* the direction bytes the team's lanes stored are visible to its lane 0 (a full-band retry rewrites lines an earlier walk has read) */
				if (E) {
					if (rb > tb_best) { tb_best = rb; tb_i = ri; tb_j = rj; }
					if (tb_best < score && 2 * w <= full) { w *= 2; fresh = true; state = TD_RETRY; }      /* next doubling (may hand over) */
					else {
						wi = tb_i; wj = tb_j; wst = 2; cur = 0; prev = 0; run = 0; nops = 0; wbase = -1; jbase = 0;
						state = TD_WALK;
					}
				}
			}
		}
		/* ---- walk back (src/ssw.c:682-762).  The walk is a chain of 10^4 dependent one-byte loads for a 10-kb read -- at one memory latency
		   each it took longer than the passes.  Here every lane of the team follows the same walk (team-uniform state), and the bytes come in
		   BATCHES: lane t fetches twelve bytes of row (base - t) around the band index the path has if it keeps to its diagonal; a step then
		   takes its byte from the lane that holds its row with one cross-lane move.  A path that drifts out of a row's window (a long
		   run of gap steps) or leaves the TEAM rows of the batch starts a new batch where it stands. */
		if (wave_any(state == TD_WALK)) {
			const bool Wk = state == TD_WALK;
			const bool stepping = Wk && wi >= 0 && wj > 0;
			int64_t pos; int o, t;
			{
				const int x0 = wi - w > 0 ? wi - w : 0;
				pos = (int64_t)width_d * wi + (wj - x0);
				t = wbase - wi;
				const int jr = jbase - t, xr = wi - w > 0 ? wi - w : 0;
				int64_t P = ((int64_t)width_d * wi + (jr - xr) - 4) & ~(int64_t)3; if (P < 0) P = 0;
				const int64_t oo = pos - P;
				o = oo < 0 || oo >= 12 ? -1 : (int)oo;
			}
			const bool reload = stepping && (t < 0 || t >= TEAM || o < 0);
			if (wave_any(reload)) {
				if (reload) {
					wbase = wi; jbase = wj; t = 0;
					const int r = wbase - k, jr = jbase - k, xr = r - w > 0 ? r - w : 0;
					int64_t P = ((int64_t)width_d * r + (jr - xr) - 4) & ~(int64_t)3; if (P < 0) P = 0;
					if (r >= 0) { const u32* src = (const u32*)(dir + P); ww0 = src[0]; ww1 = src[1]; ww2 = src[2]; }
					int64_t P0 = ((int64_t)width_d * wi + (wj - (wi - w > 0 ? wi - w : 0)) - 4) & ~(int64_t)3; if (P0 < 0) P0 = 0;
					o = (int)(pos - P0);
				}
			}
			const int sel = o >> 2;
			const u32 mine = sel <= 0 ? ww0 : sel == 1 ? ww1 : ww2;
			const u32 word = xl_shfl(mine, lead + (stepping ? t : 0));
			if (stepping) {
				const int pk = (int)((word >> (8 * (o & 3))) & 0xffu);
				const int d = wst == 0 ? 2 + (pk & 1) : wst == 1 ? 4 + ((pk >> 1) & 1) : pk >> 2;
				bool bad = false;
				if (d == 1) { --wi; --wj; wst = 2; cur = 0; }
				else if (d == 2) { --wi; wst = 0; cur = 1; }
				else if (d == 3) { --wi; wst = 2; cur = 1; }
				else if (d == 4) { --wj; wst = 1; cur = 2; }
				else if (d == 5) { --wj; wst = 2; cur = 2; }
				else bad = true;
				if (bad) { if (k == 0) { a.res[q].flag = 1; a.res[q].cigarLen = 0; } state = TD_DONE; }      /* banded_sw returns NULL: no retry (src/ssw.c:947) */
				else if (cur == prev) ++run;
				else { if (k == 0 && nops < cigcap) cig[nops] = ((u32)run << 4) | (u32)prev; ++nops; prev = cur; run = 1; }
			} else if (Wk) {
				if (cur == 0) { if (k == 0 && nops < cigcap) cig[nops] = ((u32)(run + 1) << 4); ++nops; }
				else {
					if (k == 0 && nops < cigcap) cig[nops] = ((u32)run << 4) | (u32)cur; ++nops;
					if (k == 0 && nops < cigcap) cig[nops] = (1u << 4); ++nops;
				}
				if (nops > cigcap) { if (k == 0) { a.need[job] = -1; a.need[a.nq + job] = w; } state = TD_DONE; }      /* "CIGAR slot too small": the host reports it */
				else { oi = 0; rp = 0; qp = 0; sc = 0; state = TD_REVERSE; }
			}
			if (wave_any(state == TD_REVERSE)) {      /* the operations were emitted last to first: reversed in place by the whole team, then re-scored */
				wg_fence();
				if (state == TD_REVERSE) {
					for (int x = k; x < nops / 2; x += TEAM) { const u32 tt = cig[x]; cig[x] = cig[nops - 1 - x]; cig[nops - 1 - x] = tt; }
				}
				wg_fence();
				if (state == TD_REVERSE) state = TD_SCORE;
			}
		}
		/* ---- cigar_alignment_score (src/ssw.c:785-811): one operation per iteration, the residues of an M run spread over the team's lanes */
		if (wave_any(state == TD_SCORE)) {
			const bool Sc = state == TD_SCORE, fin = Sc && oi == nops;
			int part = 0, olen = 0, oop = 0;
			if (Sc && !fin) {
				const u32 c = cig[oi];
				olen = (int)(c >> 4); oop = (int)(c & 0xf);
				if (oop == 0) for (int x = k; x < olen; x += TEAM) part += lds_ld8s(lds, (u32)((int)ref[rp + x] * n + read[qp + x]));
			}
#pragma unroll
			for (int d = 1; d < TEAM; d <<= 1) part += (int)xl_shfl((u32)part, lane ^ d);
			if (Sc && !fin) {
				if (oop == 0) { sc += part; rp += olen; qp += olen; }
				else { sc -= gapO + (olen > 1 ? (olen - 1) * gapE : 0); if (oop == 1) qp += olen; else if (oop == 2) rp += olen; }
				++oi;
			} else if (fin) {
				if (sc == score) { if (k == 0) { a.res[q].cigarLen = nops; a.res[q].cigar_off = (int64_t)q * a.cigar_stride; } state = TD_DONE; }
				else if (stage || band0 >= full) { if (k == 0) { a.res[q].flag = 1; a.res[q].cigarLen = 0; } state = TD_DONE; }
				else state = TD_RETRY;
			}
		}
	}
}

__global__ void __launch_bounds__(64) k_trace(ssw_trace_args a)
{
	const int job = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (job >= a.nq) return;
	const int q = a.qlist[job];
	ssw_dres r = a.res[q];
	a.need[job] = 0;
	if (!r.want_cigar || r.status != 0) return;
	const int8_t* ref = vm_target(a.vm, a.tgt, q) + r.ref_begin1;
	const int8_t* read = a.qcodes + a.qoff[vm_query(a.vm, q)] + r.read_begin1;
	const int refLen = r.ref_end1 - r.ref_begin1 + 1, readLen = r.read_end1 - r.read_begin1 + 1;
	const int d = refLen - readLen;
	int band = (d < 0 ? -d : d) + 1;
	const int full = refLen > readLen ? refLen : readLen;
	u32* cig = a.cigar + (int64_t)q * a.cigar_stride;   /* CIGAR slots are indexed by query, scratch by launch position */
	unsigned char* scratch = a.soff ? a.scratch + a.soff[job] : a.scratch + (int64_t)job * a.scratch_stride;
	const int64_t scap = a.soff ? a.soff[job + 1] - a.soff[job] : a.scratch_stride;
	int nops;
	for (;;) {   /* ssw.c:945-957 */
		int64_t need = 0;
		nops = trace_one(ref, read, refLen, readLen, r.score1, a.gapO, a.gapE, band, a.mat, a.n, scratch, scap,
		                 cig, (int)a.cigar_stride, &need);
		if (nops == -2) { a.need[job] = need > 0 ? (int)((need + 4095) >> 12) : -1; a.need[a.nq + job] = band; return; }   /* in 4-KiB units; the band that did not fit (as k_trace_wave) */
		if (nops < 0) break;
		if (cigar_score(cig, nops, ref, read, a.mat, a.n, a.gapO, a.gapE) == r.score1) break;
		if (band >= full) { nops = -1; break; }
		band = full;
	}
	if (nops < 0) { a.res[q].flag = 1; a.res[q].cigarLen = 0; }
	else { a.res[q].cigarLen = nops; a.res[q].cigar_off = (int64_t)q * a.cigar_stride; }
}

/* sequence preparation (reference src/main.c:84-116, 476-481, 504): mode 0 translates ASCII through the caller's
   128-entry table; mode 1 writes the reverse complement of every code sequence (codes 0..3 -> 3 - code, others unchanged:
   what the reference's rc_table + nt_table produce together); mode 2 writes the sequences AND, behind them, their reverse
   complements (one set of 2 x count sequences: `ssw_test -r` as one batch) */
__global__ void __launch_bounds__(256) k_prep(ssw_prep_args a)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.total) return;
	if (a.mode == 0) { a.out[i] = a.table[a.text[i] & 127]; return; }
	int lo = 0, hi = a.count;                      /* sequence s with off[s] <= i < off[s+1] */
	while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.off[mid] <= i) lo = mid; else hi = mid; }
	const int64_t b = a.off[lo], e = a.off[lo + 1];
	const int8_t c = a.codes_in[i];
	if (a.mode == 2) { a.out[i] = c; a.out[a.total + b + (e - 1 - i)] = (c >= 0 && c < 4) ? (int8_t)(3 - c) : c; return; }      /* originals, then their reverse complements */
	a.out[b + (e - 1 - i)] = (c >= 0 && c < 4) ? (int8_t)(3 - c) : c;
}

/* mark_mismatch on the device (reference src/ssw.c:1019-1074), one thread per alignment: leading soft clip, every M run split
   into '=' (7) / 'X' (8) runs by comparing residue codes, I and D kept, trailing soft clip; nm = mismatches + gap bases */
__global__ void __launch_bounds__(64) k_mark(ssw_mark_args a)
{
	const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (q >= a.nq) return;
	ssw_dres r = a.res[q];
	if (r.cigarLen <= 0 || r.status != 0) return;
	const u32* cig = a.cigar + r.cigar_off;
	u32* out = a.out + (int64_t)q * a.out_stride;
	const int8_t* t = vm_target(a.vm, a.tgt, q) + r.ref_begin1;
	const int qs = vm_query(a.vm, q);
	const int8_t* rd = a.qcodes + a.qoff[qs] + r.read_begin1;
	const int readLen = (int)(a.qoff[qs + 1] - a.qoff[qs]);
	int p = 0, nm = 0; u32 eq = 0, ne = 0;
	if (r.read_begin1 > 0) out[p++] = ((u32)r.read_begin1 << 4) | 4u;
	for (int i = 0; i < r.cigarLen; ++i) {
		const u32 len = cig[i] >> 4, op = cig[i] & 0xfu;
		if (op == 0 || op > 8) {
			for (u32 k = 0; k < len; ++k, ++t, ++rd) {
				if (*t != *rd) { ++nm; if (eq) { out[p++] = (eq << 4) | 7u; eq = 0; } ++ne; }
				else { if (ne) { out[p++] = (ne << 4) | 8u; ne = 0; } ++eq; }
			}
		} else if (op == 1 || op == 2) {
			if (op == 1) rd += len; else t += len;
			nm += (int)len;
			if (eq) { out[p++] = (eq << 4) | 7u; eq = 0; } else if (ne) { out[p++] = (ne << 4) | 8u; ne = 0; }
			out[p++] = (len << 4) | op;
		}
	}
	if (eq) out[p++] = (eq << 4) | 7u; else if (ne) out[p++] = (ne << 4) | 8u;
	if (readLen - r.read_end1 - 1 > 0) out[p++] = ((u32)(readLen - r.read_end1 - 1) << 4) | 4u;
	a.res[q].cigarLen = p; a.res[q].cigar_off = (int64_t)q * a.out_stride; a.res[q].nm = nm;
}

/* pack the used part of every CIGAR slot into one contiguous pool (one thread per query) */
__global__ void __launch_bounds__(256) k_gather(ssw_gather_args a)
{
	const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (q >= a.nq) return;
	const int len = a.res[q].cigarLen;
	if (len <= 0) return;
	const u32* src = a.src + a.res[q].cigar_off;
	u32* dst = a.dst + a.dst_off[q];
	for (int i = 0; i < len; ++i) dst[i] = src[i];
}

/* k_select (ssw_select_args): the pairs of a database-search chunk that go on to the reverse pass, compacted in (bucket-ordered
   query, target) order.  256 pairs per block; linear index i = k * nt + t, query order[k], record out[order[k] * nt + t]. */
SSW_DEV bool select_pred(const ssw_select_args& a, int64_t i, int64_t n, int& q, int& t, ssw_out_rec& o)
{
	if (i >= n) return false;
	const int k = (int)(i / a.nt);
	t = (int)(i - (int64_t)k * a.nt);
	q = a.order[k];
	o = a.out[(int64_t)q * a.nt + t];
	return (o.status & 0xffu) == 0 && o.score1 > 0 && !(a.flag == 0 || (a.flag == 2 && (int)o.score1 < a.filters));      /* ssw.c:900-903, 916 */
}

__global__ void __launch_bounds__(256) k_select(ssw_select_args a)
{
	SSW_DYN_LDS(lds);
	const int tid = (int)threadIdx.x;
	const int64_t n = (int64_t)a.nk * a.nt;
	if (a.pass == 1) {      /* one workgroup: exclusive scan of the block counts (each thread a contiguous run of blocks) */
		const int per = (a.nblk + 255) / 256;
		const int b0 = tid * per, b1 = b0 + per < a.nblk ? b0 + per : a.nblk;
		int sum = 0;
		for (int b = b0; b < b1; ++b) sum += a.blk[b];
		lds_st32(lds, 4u * tid, (u32)sum);
		__syncthreads();
		if (tid == 0) {
			int run = 0;
			for (int k = 0; k < 256; ++k) { const int v = (int)lds_ld32(lds, 4u * k); lds_st32(lds, 4u * k, (u32)run); run += v; }
			a.blk[a.nblk] = run;
			a.bucket_first[a.nbk] = run;
		}
		__syncthreads();
		int run = (int)lds_ld32(lds, 4u * tid);
		for (int b = b0; b < b1; ++b) { const int v = a.blk[b]; a.blk[b] = run; run += v; }
		return;
	}
	const int64_t i = (int64_t)blockIdx.x * 256 + tid;
	int q = 0, t = 0; ssw_out_rec o;
	const bool keep = select_pred(a, i, n, q, t, o);
	/* rank of this thread among the block's survivors: ballots of the four wavefronts through LDS */
	const unsigned long long bal = wave_ballot(keep);
	const int wave = tid >> 6, lane = tid & 63;
	if (lane == 0) lds_st32(lds, 4u * wave, (u32)__builtin_popcountll(bal));
	__syncthreads();
	int before = 0, total = 0;
	for (int w = 0; w < 4; ++w) { const int cnt = (int)lds_ld32(lds, 4u * w); if (w < wave) before += cnt; total += cnt; }
	if (a.pass == 0) { if (tid == 0) a.blk[blockIdx.x] = total; return; }
	const int pos = a.blk[blockIdx.x] + before + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
	for (int b = 0; b < a.nbk; ++b) if (a.bucket_lin[b] == i) a.bucket_first[b] = pos;      /* (pos = survivors before i, whether i survives or not) */
	if (i < n && (o.status & SSW_OUT_WORD)) a.out[(int64_t)q * a.nt + t].status = (uint16_t)(o.status & 0xffu);
	if (keep && pos < a.cap) {
		ssw_dres r;
		r.score1 = o.score1; r.score2 = o.score2; r.ref_begin1 = -1; r.ref_end1 = o.ref_end1; r.read_begin1 = -1; r.read_end1 = o.read_end1;
		r.ref_end2 = o.ref_end2; r.cigarLen = 0; r.flag = 0; r.status = 0; r.word = (o.status & SSW_OUT_WORD) ? 1 : 0; r.want_begin = 1; r.want_cigar = 0;
		r.rev_score = 0; r.loc_done = 1; r.nm = 0; r.cigar_off = 0;
		a.sres[pos] = r; a.svq[pos] = q; a.svt[pos] = a.tbase + t; a.vlist[pos] = pos;
	}
}

/* ------------------------------------------------------------------------------------------------
 * k_selftest: (a) the cross-lane primitives applied to the lane id, so that tests can pin the DPP semantics the
 * chains rely on (and that the CPU emulator assumes) on real hardware; (b) a packed-int16 VALU issue-rate probe
 * (8 independent v_pk_add_i16/v_pk_max_i16/v_pk_sub_u16 chains) whose measured rate is the roofline's "peak".
 * ------------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(256) k_selftest(ssw_selftest_args a)
{
	const u32 tid = (u32)threadIdx.x, gid = (u32)(blockIdx.x * blockDim.x + threadIdx.x);
	if (a.lanes_out && blockIdx.x == 0 && tid < 64) {
		const u32 v = 100u + tid;
		a.lanes_out[0 * 64 + tid] = xl_row_shr1_zero(v);
		a.lanes_out[1 * 64 + tid] = xl_row_shr1_keep(7000u + tid, v);
		a.lanes_out[2 * 64 + tid] = xl_row_ror<1>(v);
		a.lanes_out[3 * 64 + tid] = xl_row_ror<2>(v);
		a.lanes_out[4 * 64 + tid] = xl_row_ror<8>(v);
		a.lanes_out[5 * 64 + tid] = xl_shfl(v, (int)((tid + 16) & 63));
		a.lanes_out[6 * 64 + tid] = pk_adds(pk_make(30000, -30000), pk_make((int)tid * 100, -(int)tid * 100));
		a.lanes_out[7 * 64 + tid] = pk_subu(pk_make((int)tid, 5), pk_make(10, (int)tid));
		a.lanes_out[8 * 64 + tid] = pk_max(pk_make((int)tid - 32, 3), pk_make(0, (int)tid - 60));
		a.lanes_out[9 * 64 + tid] = xl_wave_shr1_keep(7000u + tid, v);
		a.lanes_out[10 * 64 + tid] = xl_row_shr_keep<2>(7000u + tid, v);
		a.lanes_out[11 * 64 + tid] = xl_row_shr_keep<4>(7000u + tid, v);
		a.lanes_out[12 * 64 + tid] = xl_row_shr_keep<8>(7000u + tid, v);
		a.lanes_out[13 * 64 + tid] = xl_row_bcast15_keep(7000u + tid, v);
		a.lanes_out[14 * 64 + tid] = xl_row_bcast31_keep(7000u + tid, v);
		a.lanes_out[15 * 64 + tid] = xl_readlane(v, 37);
	}
	if (a.iters > 0) {
		u32 x0 = gid, x1 = gid * 3u, x2 = gid * 5u, x3 = gid * 7u, x4 = gid * 11u, x5 = gid * 13u, x6 = gid * 17u, x7 = gid * 19u;
		const u32 g = a.seed;
		for (int it = 0; it < a.iters; ++it) {
#pragma unroll
			for (int k = 0; k < 4; ++k) {   /* 8 chains x 3 packed ops x 4 = 96 VOP3P per iteration */
				x0 = pk_max(pk_subu(pk_adds(x0, g), g), x0); x1 = pk_max(pk_subu(pk_adds(x1, g), g), x1);
				x2 = pk_max(pk_subu(pk_adds(x2, g), g), x2); x3 = pk_max(pk_subu(pk_adds(x3, g), g), x3);
				x4 = pk_max(pk_subu(pk_adds(x4, g), g), x4); x5 = pk_max(pk_subu(pk_adds(x5, g), g), x5);
				x6 = pk_max(pk_subu(pk_adds(x6, g), g), x6); x7 = pk_max(pk_subu(pk_adds(x7, g), g), x7);
			}
		}
		if (a.sink) a.sink[gid] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
	}
}

/* ================================================================================================
 * launchers (the thin C shim of SURVEY 8b: host code stays C)
 * ================================================================================================ */
#ifdef SSW_SIMT_EMU
template <class A, void (*K)(A)> static void emu_thunk(void* p) { K(*(A*)p); }
#define SSW_LAUNCH(kern, A, args, grid, block, ldsbytes, stream) \
	do { (void)(stream); emu::launch(&emu_thunk<A, kern>, (void*)&(args), (unsigned)(grid), (unsigned)(block), (size_t)(ldsbytes)); } while (0)
#define SSW_LAUNCH_OK() 0
#else
static thread_local char g_shim_err[256];
static int shim_check(hipError_t e, const char* what)
{
	if (e == hipSuccess) return 0;
	snprintf(g_shim_err, sizeof g_shim_err, "%s: %s", what, hipGetErrorString(e));
	return -1;
}
/* dynamic LDS above 64 KiB (large protein profiles: up to 160 KiB per workgroup on gfx950) needs an opt-in per kernel; a
   refused opt-in (or a request beyond the device limit) makes the launch fail with a message instead of a late HIP error */
#define SSW_LDS_MAX (160u * 1024u)
static thread_local int g_lds_refused;
template <class K> static void shim_allow_lds(K kern, size_t bytes)
{
	if (bytes <= 65536) return;
	if (bytes > SSW_LDS_MAX) { g_lds_refused = 1; snprintf(g_shim_err, sizeof g_shim_err, "kernel needs %zu bytes of LDS (limit %u)", bytes, SSW_LDS_MAX); return; }
	const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
	if (e != hipSuccess) { g_lds_refused = 1; snprintf(g_shim_err, sizeof g_shim_err, "hipFuncSetAttribute(%zu bytes of LDS): %s", bytes, hipGetErrorString(e)); }
}
#define SSW_LAUNCH(kern, A, args, grid, block, ldsbytes, stream) \
	do { g_lds_refused = 0; shim_allow_lds(kern, (size_t)(ldsbytes)); \
	     if (!g_lds_refused) hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (size_t)(ldsbytes), (hipStream_t)(stream), args); } while (0)
#define SSW_LAUNCH_OK() (g_lds_refused ? -1 : shim_check(hipGetLastError(), "kernel launch"))
#endif

#define FOR_EACH_R(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
                      X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)

extern "C" int ssw_shim_launch_fill(int R, const ssw_fill_args* a, void* stream)
{
	ssw_fill_args args = *a;
	const int64_t grid = (int64_t)args.npairs * args.bpp;
	if (grid <= 0) return 0;
	switch (R) {
#define X(r) case r: { const size_t ldsb = (size_t)(args.n + 1) * ChainGeom<r>::PSTRIDE + 16 * CHAIN_BYTES; \
		if (args.form == 3) SSW_LAUNCH((k_fill<r, 3>), ssw_fill_args, args, grid, 256, ldsb, stream); \
		else SSW_LAUNCH((k_fill<r, 0>), ssw_fill_args, args, grid, 256, ldsb, stream); } break;
		FOR_EACH_R(X)
#undef X
		default: return -2;
	}
	return SSW_LAUNCH_OK();
}

/* rows per lane of the fused database-search kernel: 1..24 (queries up to 384 residues, shared with k_fill) and 25..40
   (385..640 residues: still one strip, exact ceil(len / 16) like the short classes) */
#define FOR_EACH_DBR_LONG(X) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40)

extern "C" int ssw_shim_launch_filldb(int R, const ssw_filldb_args* a, void* stream)
{
	ssw_filldb_args args = *a;
	const int nch = SSW_DB_NCH;
	const int64_t grid = (int64_t)args.npairs * ((args.ntl + nch - 1) / nch);
	if (grid <= 0) return 0;
	switch (R) {
#define X(r) case r: { const size_t ldsb = (size_t)(args.n + 1) * ChainGeom<r>::PSTRIDE + (size_t)nch * DB_CHAIN_BYTES + 16; \
		if (args.form) SSW_LAUNCH((k_filldb<r, SSW_DB_NCH, true>), ssw_filldb_args, args, grid, 16 * SSW_DB_NCH, ldsb, stream); \
		else SSW_LAUNCH((k_filldb<r, SSW_DB_NCH, false>), ssw_filldb_args, args, grid, 16 * SSW_DB_NCH, ldsb, stream); } break;
		FOR_EACH_R(X)
		FOR_EACH_DBR_LONG(X)
#undef X
		default: return -2;
	}
	return SSW_LAUNCH_OK();
}

/* several buckets in one grid: hR = the rows per lane of the nsub sub-launches (host copy: sizes the LDS), all of one register class and form */
extern "C" int ssw_shim_fill_class(int R) { return R <= 10 ? 0 : R <= 16 ? 1 : 2; }
extern "C" int ssw_shim_launch_fillm(const ssw_fillm_args* a, const int32_t* hR, int n, int form, int64_t total_wgs, void* stream)
{
	ssw_fillm_args args = *a;
	if (total_wgs <= 0 || args.nsub <= 0) return 0;
	const int cls = ssw_shim_fill_class(hR[0]);
	size_t ldsb = 0;
	for (int i = 0; i < args.nsub; ++i) {
		if (ssw_shim_fill_class(hR[i]) != cls || hR[i] < 1 || hR[i] > SSW_RMAX) return -2;
		const size_t b = (size_t)(n + 1) * (size_t)((hR[i] + 3) / 4) * 256 + 16 * CHAIN_BYTES;
		if (b > ldsb) ldsb = b;
	}
#define X(c) if (cls == c) { if (form == 3) SSW_LAUNCH((k_fillm<c, 3>), ssw_fillm_args, args, total_wgs, 256, ldsb, stream); \
                             else SSW_LAUNCH((k_fillm<c, 0>), ssw_fillm_args, args, total_wgs, 256, ldsb, stream); }
	X(0) X(1) X(2)
#undef X
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_launch_reduce(const ssw_reduce_args* a, void* stream)
{
	ssw_reduce_args args = *a;
	if (args.npairs <= 0) return 0;
	if (args.sg16) {
		/* a handful of pairs against a long target (one ssw_align call: 62 500 groups of a 1 Mb target): 1024 threads share the scan -- 64 -> ~20 us */
		const int nthr = args.npairs <= 8 && args.refLen >= (1 << 17) ? 1024 : 256;
		SSW_LAUNCH(k_reduce_seg, ssw_reduce_args, args, args.npairs, nthr, nthr * 8, stream);
	}
	else SSW_LAUNCH(k_reduce, ssw_reduce_args, args, args.npairs, 256, 256 * 8, stream);
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_launch_reducem(const ssw_reducem_args* a, int64_t total_pairs, void* stream)
{
	ssw_reducem_args args = *a;
	if (total_pairs <= 0 || args.nsub <= 0) return 0;
	SSW_LAUNCH(k_reducem, ssw_reducem_args, args, total_pairs, 256, 256 * 8, stream);
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_launch_capture(int R, const ssw_capture_args* a, void* stream)
{
	ssw_capture_args args = *a;
	if (args.nq <= 0) return 0;
	const int grid = (args.nq + 3) / 4;
	switch (R) {
#define X(r) case r: { const size_t ldsb = 4 * ((size_t)(args.n + 1) * ChainGeom<r>::PSTRIDE + CHAIN_BYTES); \
		SSW_LAUNCH(k_capture<r>, ssw_capture_args, args, grid, 64, ldsb, stream); } break;
		FOR_EACH_R(X)
#undef X
		default: return -2;
	}
	return SSW_LAUNCH_OK();
}

/* dynamic LDS one k_capture<R> workgroup asks for (4 chains, each with its own profile): the host routes buckets whose
   request exceeds the device limit (wide alphabets x many rows per lane) through k_chainx, one profile per wavefront */
extern "C" int64_t ssw_shim_capture_lds_need(int R, int n)
{
	if (R < 1 || R > SSW_RMAX) return -1;
	const int64_t C = (R + 3) / 4;
	return 4 * ((int64_t)(n + 1) * C * 256 + CHAIN_BYTES);
}

extern "C" int ssw_shim_launch_chainx(int R, int capture, const ssw_chainx_args* a, void* stream)
{
	ssw_chainx_args args = *a;
	if (capture) { args.nlist = args.njobs; args.njobs = (args.njobs + 1) / 2; }   /* two queries of the list per job */
	if (args.njobs <= 0) return 0;
	if (args.lanes != 16) return -2;      /* 64-lane chains run behind the work queue: ssw_shim_launch_chainq */
	const int grid = (args.njobs + 3) / 4;
	switch (R) {
#define X(r) case r: { const size_t ldsb = 4 * ((size_t)(args.n + 1) * StripGeom<r, 16>::PSTRIDE + StripGeom<r, 16>::EXTRA); \
		if (capture) SSW_LAUNCH((k_chainx<r, true, 16>), ssw_chainx_args, args, grid, 64, ldsb, stream); \
		else SSW_LAUNCH((k_chainx<r, false, 16>), ssw_chainx_args, args, grid, 64, ldsb, stream); } break;
		FOR_EACH_R(X)
#undef X
		default: return -2;
	}
	return SSW_LAUNCH_OK();
}

#define FOR_EACH_QR(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

/* 64-lane chains behind the work queue: a persistent launch of at most max_workgroups wavefronts (0: one per item) that
   draws njobs x strips items; args.queue must be zeroed (1 + items + 1 words: ticket counter, flags, error word),
   args.cand_strip holds 8 words per item */
extern "C" int ssw_shim_launch_chainq(int R, int capture, const ssw_chainx_args* a, int max_workgroups, void* stream)
{
	ssw_chainx_args args = *a;
	if (capture) { args.nlist = args.njobs; args.njobs = (args.njobs + 1) / 2; }
	if (args.njobs <= 0 || args.strips <= 0) return 0;
	const int64_t items = (int64_t)args.njobs * args.strips;
	if (items > 0x7ffffff0) return -2;
	int64_t grid = args.whole_jobs ? args.njobs : items;
	if (max_workgroups > 0 && grid > max_workgroups) grid = max_workgroups;
	switch (R) {
#define X(r) case r: { const size_t ldsb = (size_t)(args.n + 1) * StripGeom<r, 64>::PSTRIDE + (capture ? QueueGeom<r, true>::EXTRA : QueueGeom<r, false>::EXTRA); \
		if (capture && args.form == 3) SSW_LAUNCH((k_chainq<r, true, 3>), ssw_chainx_args, args, grid, 64, ldsb, stream); \
		else if (capture) SSW_LAUNCH((k_chainq<r, true, 0>), ssw_chainx_args, args, grid, 64, ldsb, stream); \
		else if (args.form == 3) SSW_LAUNCH((k_chainq<r, false, 3>), ssw_chainx_args, args, grid, 64, ldsb, stream); \
		else SSW_LAUNCH((k_chainq<r, false, 0>), ssw_chainx_args, args, grid, 64, ldsb, stream); } break;
		FOR_EACH_QR(X)
#undef X
		default: return -2;
	}
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_chainq_resident(int R, int capture, int n)
{
#ifdef SSW_SIMT_EMU
	(void)R; (void)capture; (void)n; return 0;
#else
	static int cache[17][2][SSW_MAX_N + 1];      /* the occupancy query is not free and the answer does not change */
	if (R < 1 || R > 16 || n < 1 || n > SSW_MAX_N) return 0;
	if (cache[R][capture ? 1 : 0][n] > 0) return cache[R][capture ? 1 : 0][n];
	int per_cu = 0, dev = 0, cus = 0;
	switch (R) {
#define X(r) case r: { const size_t ldsb = (size_t)(n + 1) * StripGeom<r, 64>::PSTRIDE + (capture ? QueueGeom<r, true>::EXTRA : QueueGeom<r, false>::EXTRA); \
		if (capture) { shim_allow_lds(k_chainq<r, true, 0>, ldsb); if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chainq<r, true, 0>, 64, ldsb) != hipSuccess) per_cu = 0; } \
		else { shim_allow_lds(k_chainq<r, false, 0>, ldsb); if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chainq<r, false, 0>, 64, ldsb) != hipSuccess) per_cu = 0; } } break;
		FOR_EACH_QR(X)
#undef X
		default: return 0;
	}
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
	if (per_cu <= 0) {   /* the occupancy query gave nothing (seen on this stack for these kernels): 160 KiB of LDS per CU bound the
	                        one-wavefront workgroups, the register file (at least two wavefronts per SIMD up to 256 VGPRs) does the rest */
		const size_t ldsb = (size_t)(n + 1) * ((size_t)((R + 3) / 4) * 1024) + (capture ? 2176 : 1856);
		per_cu = (int)(SSW_LDS_MAX / ldsb);
		if (per_cu > 12) per_cu = 12;
		if (per_cu < 1) per_cu = 1;
	}
	cache[R][capture ? 1 : 0][n] = per_cu * cus;
	return per_cu * cus;
#endif
}

extern "C" int ssw_shim_launch_literal(const ssw_literal_args* a, void* stream)
{
	ssw_literal_args args = *a;
	if (args.nq <= 0) return 0;
	/* short reads: the per-alignment state lives in LDS, 16 alignments per 256-thread workgroup (up to 64 KiB), or 4 per wavefront-sized
	   workgroup (up to 160 KiB); longer reads keep it in the HBM scratch */
	const int64_t st = args.state_bytes;
	int threads = 64;
	args.lds_stride = 0;
	if (args.n < 1 || args.n > SSW_MAX_N_WIDE) return -1;
	const int64_t matb = ((int64_t)args.n * args.n + 15) / 16 * 16 < 1024 ? 1024 : ((int64_t)args.n * args.n + 15) / 16 * 16;      /* the matrix in LDS: 1 KiB up to 32 letters, 16 KiB at 128 */
	/* (an alignment is one DPP row -- 16 lanes -- whatever the workgroup: small batches take wavefront-sized workgroups so that
	   they spread over more CUs) */
	if (st > 0 && st * 16 + matb <= 65536 + 1024 && args.nq >= 16 * 2048) { threads = 256; args.lds_stride = (int32_t)st; }
	else if (st > 0 && st * 4 + matb <= (int64_t)SSW_LDS_LIMIT) { threads = 64; args.lds_stride = (int32_t)st; }
	const int per = threads / 16;
	if (args.pass != 0 || args.score_size != 2) { args.spec_cnt = 0; args.spec_out = 0; }
	const int njobs = args.spec_cnt ? ((args.nq + 3) & ~3) + args.nq : args.nq;      /* both rule sets of every query side by side (k_literal) */
	SSW_LAUNCH(k_literal, ssw_literal_args, args, (njobs + per - 1) / per, threads, (size_t)args.lds_stride * per + (size_t)matb, stream);
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_launch_trace(const ssw_trace_args* a, void* stream)
{
	ssw_trace_args args = *a;
	if (args.nq <= 0) return 0;
	SSW_LAUNCH(k_trace, ssw_trace_args, args, (args.nq + 63) / 64, 64, 0, stream);
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_launch_trace_diag(int team, const ssw_trace_args* a, void* stream)
{
	ssw_trace_args args = *a;
	if (args.nq <= 0) return 0;
	if (args.n * args.n > 1024) return -1;      /* (the matrix lives in 1 KiB of LDS; SSW_MAX_N = 32) */
	if (team == 32) SSW_LAUNCH((k_trace_diag<32>), ssw_trace_args, args, (args.nq + 1) / 2, 64, 1024 + 512 * 2, stream);
	else SSW_LAUNCH((k_trace_diag<16>), ssw_trace_args, args, (args.nq + 3) / 4, 64, 1024 + 512 * 4, stream);
	return SSW_LAUNCH_OK();
}

extern "C" int64_t ssw_shim_trace_lds_need(int band_width, int waves) { return trace_lds_need(band_width, 64 * waves); }

extern "C" int ssw_shim_launch_trace_wave(const ssw_trace_args* a, void* stream)
{
	ssw_trace_args args = *a;
	if (args.nq <= 0) return 0;
	if (args.lds_bytes < (int32_t)TRACE_LDS_FIXED) args.lds_bytes = (int32_t)TRACE_LDS_FIXED;
	switch (args.waves) {
		case 16: SSW_LAUNCH((k_trace_wave<16>), ssw_trace_args, args, args.nq, 1024, (size_t)args.lds_bytes, stream); break;
		case 4: SSW_LAUNCH((k_trace_wave<4>), ssw_trace_args, args, args.nq, 256, (size_t)args.lds_bytes, stream); break;
		default: SSW_LAUNCH((k_trace_wave<1>), ssw_trace_args, args, args.nq, 64, (size_t)args.lds_bytes, stream); break;
	}
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_launch_selftest(const ssw_selftest_args* a, int blocks, void* stream)
{
	ssw_selftest_args args = *a;
	SSW_LAUNCH(k_selftest, ssw_selftest_args, args, blocks, 256, 0, stream);
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_launch_prep(const ssw_prep_args* a, void* stream)
{
	ssw_prep_args args = *a;
	if (args.total <= 0) return 0;
	SSW_LAUNCH(k_prep, ssw_prep_args, args, (args.total + 255) / 256, 256, 0, stream);
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_launch_mark(const ssw_mark_args* a, void* stream)
{
	ssw_mark_args args = *a;
	if (args.nq <= 0) return 0;
	SSW_LAUNCH(k_mark, ssw_mark_args, args, (args.nq + 63) / 64, 64, 0, stream);
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_launch_select(const ssw_select_args* a, void* stream)
{
	ssw_select_args args = *a;
	const int64_t n = (int64_t)args.nk * args.nt;
	const int grid = args.pass == 1 ? 1 : (int)((n + 255) / 256);
	if (grid <= 0) return 0;
	SSW_LAUNCH(k_select, ssw_select_args, args, grid, 256, 1024, stream);
	return SSW_LAUNCH_OK();
}

extern "C" int ssw_shim_launch_gather(const ssw_gather_args* a, void* stream)
{
	ssw_gather_args args = *a;
	if (args.nq <= 0) return 0;
	SSW_LAUNCH(k_gather, ssw_gather_args, args, (args.nq + 255) / 256, 256, 0, stream);
	return SSW_LAUNCH_OK();
}

#ifndef SSW_SIMT_EMU
/* ---- HIP runtime part of the shim ---- */
extern "C" const char* ssw_shim_last_error(void) { return g_shim_err; }
extern "C" int ssw_shim_device_count(void)
{
	int n = 0;
	if (shim_check(hipGetDeviceCount(&n), "hipGetDeviceCount")) return 0;
	return n;
}
extern "C" int ssw_shim_set_device(int dev) { return shim_check(hipSetDevice(dev), "hipSetDevice"); }
extern "C" void* ssw_shim_stream_create(void)
{
	hipStream_t s = 0;
	if (shim_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate")) return 0;
	return (void*)s;
}
/* a stream whose kernels the command processor dispatches only when the normal-priority queues have nothing to dispatch (the filler of a
   pipelined series of fill launches: ssw_host.c "pipe") */
extern "C" void* ssw_shim_stream_create_low(void)
{
	hipStream_t s = 0;
	int least = 0, greatest = 0;
	if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = 0; (void)hipGetLastError(); }
	if (shim_check(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least), "hipStreamCreateWithPriority")) return 0;
	return (void*)s;
}
extern "C" void ssw_shim_stream_destroy(void* s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
extern "C" int ssw_shim_stream_sync(void* s) { return shim_check(hipStreamSynchronize((hipStream_t)s), "hipStreamSynchronize"); }
extern "C" void* ssw_shim_malloc(size_t bytes)
{
	void* p = 0;
	if (shim_check(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc")) {
		(void)hipGetLastError();      /* the runtime keeps the last error until it is read: a caller that recovers from this failure (smaller
		                                 budget, ssw_host.c SSW_ALLOC_RETRY) must not meet it again as "kernel launch: out of memory" */
		return 0;
	}
	return p;
}
extern "C" void ssw_shim_free(void* p) { if (p) (void)hipFree(p); }
extern "C" void* ssw_shim_host_alloc(size_t bytes)
{
	void* p = 0;
	if (shim_check(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault), "hipHostMalloc")) { (void)hipGetLastError(); return 0; }
	return p;
}
extern "C" void ssw_shim_host_free(void* p) { if (p) (void)hipHostFree(p); }
extern "C" int ssw_shim_h2d(void* d, const void* s, size_t n, void* st) { return n ? shim_check(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st), "hipMemcpy H2D") : 0; }
extern "C" int ssw_shim_d2h(void* d, const void* s, size_t n, void* st) { return n ? shim_check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st), "hipMemcpy D2H") : 0; }
extern "C" int ssw_shim_memset(void* d, int v, size_t n, void* st) { return n ? shim_check(hipMemsetAsync(d, v, n, (hipStream_t)st), "hipMemset") : 0; }
extern "C" size_t ssw_shim_mem_free_bytes(void)
{
	size_t f = 0, t = 0;
	if (hipMemGetInfo(&f, &t) != hipSuccess) return 0;
	return f;
}
extern "C" int ssw_shim_device_props(int* compute_units, int* waves_per_cu)
{
	int dev = 0;
	hipDeviceProp_t pr;
	if (shim_check(hipGetDevice(&dev), "hipGetDevice") || shim_check(hipGetDeviceProperties(&pr, dev), "hipGetDeviceProperties")) return -1;
	*compute_units = pr.multiProcessorCount;
	*waves_per_cu = pr.maxThreadsPerMultiProcessor / 64;
	return 0;
}
extern "C" void* ssw_shim_event_create(void)
{
	hipEvent_t e = 0;
	if (shim_check(hipEventCreate(&e), "hipEventCreate")) return 0;
	return (void*)e;
}
extern "C" void ssw_shim_event_destroy(void* e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
extern "C" int ssw_shim_event_record(void* e, void* s) { return shim_check(hipEventRecord((hipEvent_t)e, (hipStream_t)s), "hipEventRecord"); }
extern "C" int ssw_shim_event_sync(void* e) { return shim_check(hipEventSynchronize((hipEvent_t)e), "hipEventSynchronize"); }
extern "C" int ssw_shim_stream_wait_event(void* s, void* e) { return shim_check(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0), "hipStreamWaitEvent"); }
extern "C" float ssw_shim_event_elapsed_ms(void* a, void* b)
{
	float ms = 0.f;
	if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.f;
	return ms;
}
#endif
