"""Shared comparison helpers: batch results of the library (product or emulated) vs the checkers."""
import numpy as np

from sswutil import RES_FIELDS, cigar_str, oracle_align, ref_align, ref_lib
from sswutil import dna_matrix as _dna_matrix, random_ref as _random_ref


def expected(read, mat, n, ref, gapO, gapE, flag, filters, filterd, maskLen, score_size, use_ref=True):
    """The reference's answer: the compiled reference when oracle/_ref is present, else the pinned oracle."""
    if use_ref and ref_lib() is not None:
        return ref_align(read, mat, n, ref, gapO, gapE, flag, filters, filterd, maskLen, score_size)
    d, cig = oracle_align(read, mat, n, ref, gapO, gapE, flag, filters, filterd, maskLen, score_size, 0)
    return (None if d is None else {k: d[k] for k in RES_FIELDS}), cig


def compare_batch(res, cig, reads, refs, mat, n, gapO, gapE, flag, filters, filterd, maskLen, score_size, use_ref=True,
                  max_report=3):
    """res: [nq, nt] record array from Context.align_batch.  Returns the list of mismatch descriptions."""
    bad = []
    for qi, rd in enumerate(reads):
        ml = maskLen if maskLen >= 0 else len(rd) // 2
        for ti, rf in enumerate(refs):
            exp, ecig = expected(rd, mat, n, rf, gapO, gapE, flag, filters, filterd, ml, score_size, use_ref)
            g = res[qi, ti]
            if exp is None:
                ok = int(g["status"]) == 1
                got, gc = {"status": int(g["status"])}, []
            else:
                got = {k: int(g[k]) for k in RES_FIELDS}
                off, ln = int(g["cigar_off"]), int(g["cigarLen"])
                gc = [int(x) for x in cig[off:off + ln]] if ln > 0 else []
                ok = int(g["status"]) == 0 and got == exp and gc == ecig
            if not ok and len(bad) < max_report:
                bad.append("q%d(len %d) x t%d(len %d): expected %s %s got %s %s" %
                           (qi, len(rd), ti, len(rf), exp, cigar_str(ecig), got, cigar_str(gc)))
            elif not ok:
                bad.append("...")
    return bad


def make_reads(rng, ref, nq, lens, nc, sub=0.06, ins=0.02, dele=0.02, frac_random=0.2):
    from sswutil import mutate
    reads = []
    for i in range(nq):
        rl = int(lens[i % len(lens)])
        if rng.random() >= frac_random and len(ref) > rl + 24:
            off = int(rng.integers(0, len(ref) - rl - 16))
            r = mutate(ref[off:off + rl + 8], rng, sub, ins, dele, nc)[:rl]
            if len(r) < rl:
                r = np.concatenate([r, rng.integers(0, nc, size=rl - len(r), dtype=np.int8)])
        else:
            r = rng.integers(0, nc, size=rl, dtype=np.int8)
        reads.append(np.ascontiguousarray(r, dtype=np.int8))
    return reads


def free_gap_open_case(rng):
    """One batch of the `gapO = 0` regime (a legal argument of ssw.h:126-134; round-4 fuzz): opening a gap is free, so an alignment of a 33-base
    read can span hundreds of target bases -- far beyond what the gap-extension argument bounds -- and banded_sw (src/ssw.c:590-783) mostly
    gives up on it after the full-band retry (ssw.c:945-957): `cigar NULL, cigarLen 0, flag 1` is then the reference's answer.  Draws what
    makes the spans long: sparse matrices (match 1-2, every mismatch negative) over alphabets up to 24 letters, reads unrelated to the target,
    gapE 0-3, and every flag that asks for a CIGAR.  -> (reads, ref, mat, n, gapO, gapE, flag, filterd, maskLen)"""
    n = int(rng.integers(4, 25)); nc = n - 1 if n > 4 else n
    kind = rng.random()
    if kind < 0.35:
        mat = np.full((n, n), -int(rng.integers(1, 6)), dtype=np.int8)
        for i in range(n):
            mat[i, i] = int(rng.choice([1, 1, 2]))
        mat = np.ascontiguousarray(mat.reshape(-1))
    elif kind < 0.6:
        hi = int(rng.choice([1, 1, 2, 8]))
        mat = np.minimum(rng.integers(-8, 9, size=(n, n)), hi).astype(np.int8)
        for i in range(n):
            mat[i, i] = rng.integers(1, hi + 1)
        mat = np.ascontiguousarray(mat.reshape(-1))
    else:
        from sswutil import dna_matrix
        n, nc, mat = 5, 4, dna_matrix(int(rng.choice([1, 1, 2, 3])), int(rng.integers(1, 6)))
    ref = rng.integers(0, nc, size=int(rng.integers(120, 701)), dtype=np.int8)
    nq = int(rng.integers(1, 6))
    reads = make_reads(rng, ref, nq, rng.integers(15, 71, size=nq), nc)
    if rng.random() < 0.5:
        reads = [rng.integers(0, nc, size=len(r), dtype=np.int8) for r in reads]
    return (reads, ref, mat, n, 0, int(rng.choice([0, 1, 1, 2, 3])), int(rng.choice([1, 2, 9, 12, 15])),
            int(rng.choice([0, 1000])), int(rng.choice([-1, 15, 40])))


def narrow_band_batches(rng, count):
    """batches for the 16-lane traceback teams (k_trace_diag): several alignments per wavefront, teams partly filled, reads and targets short
    enough that a band spans the whole target (the reference's forced index then lands on a real cell: rows 1 .. w + 1, last column), indels
    that make a band double once or twice inside the team's range, some that outgrow it (hand-over to the row kernel), unrelated reads"""
    for _ in range(count):
        kind = rng.random()
        if kind < 0.5:      # tiny: the band covers the target
            ref = rng.integers(0, 4, size=int(rng.integers(6, 70)), dtype=np.int8)
            nq = int(rng.integers(1, 14))
            reads = make_reads(rng, ref, nq, rng.integers(3, 60, size=nq), 4, sub=0.08, ins=0.05, dele=0.05, frac_random=0.3)
        else:
            ref = _random_ref(int(rng.integers(200, 1500)), int(rng.integers(1 << 30)), 4)
            nq = int(rng.integers(3, 11))
            reads = make_reads(rng, ref, nq, rng.integers(30, 420, size=nq), 4, sub=0.04, ins=0.02, dele=0.02, frac_random=0.15)
            if rng.random() < 0.4:      # one read with a long deletion: its band starts beyond the team or outgrows it
                o = int(rng.integers(0, len(ref) - 150))
                reads[0] = np.ascontiguousarray(np.concatenate([ref[o:o + 50], ref[o + 50 + int(rng.integers(10, 40)):o + 140]]))
        gapE = int(rng.integers(1, 3)); gapO = gapE + int(rng.integers(1, 5))
        yield reads, ref, _dna_matrix(int(rng.integers(1, 4)), int(rng.integers(1, 5))), gapO, gapE, int(rng.choice([1, 2, 2, 9, 15]))
