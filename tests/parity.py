"""Shared comparison helpers: batch results of the library (product or emulated) vs the checkers."""
import numpy as np

from sswutil import RES_FIELDS, cigar_str, oracle_align, ref_align, ref_lib
from sswutil import blosum50 as _blosum50, dna_matrix as _dna_matrix, random_ref as _random_ref


def expected(read, mat, n, ref, gapO, gapE, flag, filters, filterd, maskLen, score_size, use_ref=True):
    """The reference's answer: the compiled reference when oracle/_ref is present, else the pinned oracle.
    An EMPTY read is outside the reference's domain (src/ssw.c:264 / :469 index pvHStore[segLen - 1] with segLen == 0: undefined behaviour),
    but include/ssw.h and ssw_gpu.h declare it legal here: its answer is the record ssw_align starts from (src/ssw.c:870-875) -- score 0,
    begins -1 -- which is also what the reference returns for `bests[0].score <= 0` (ssw.c:900-903).  The checkers are not called with it."""
    if len(read) == 0:
        return dict(zip(RES_FIELDS, (0, 0, -1, 0, -1, 0, 0, 0, 0))), []
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


def empties_two_call_repro(run):
    """round-5 verdict, weak #1: k_literal's launches were counted over ALL queries while the device query list holds only the non-empty ones;
    the jobs past the list read whatever the previous call left in the header buffer as a query index and wrote a record through it.  Call A
    leaves large indices and a 28-letter matrix behind, call B (gapO <= gapE, empty queries in the batch) then walked off the list."""
    rng = np.random.default_rng(5)
    n = 28
    matA = np.ascontiguousarray(rng.integers(-128, 128, size=(n, n)).astype(np.int8).reshape(-1))
    matA[40:96] = 127      # (where call B's header buffer keeps its query list: stale words there read as huge positive indices)
    refA = rng.integers(0, n - 1, size=300, dtype=np.int8)
    run([rng.integers(0, n - 1, size=60, dtype=np.int8) for _ in range(6)], [refA], matA, n, 9, 3, 0)
    ref = _random_ref(200, 31, 4)
    empty = np.zeros(0, dtype=np.int8)
    for flag in (0, 1, 9):
        reads = [ref[20:70].copy(), empty, ref[100:140].copy(), empty]
        res = run(reads, [ref], _dna_matrix(2, 2), 5, 1, 1, flag)
        assert (res["score1"][[1, 3], 0] == 0).all() and (res["ref_begin1"][[1, 3], 0] == -1).all() and (res["score1"][[0, 2], 0] > 0).all()


def empties_case(rng):
    """one batch with empty queries and / or empty targets among the slots, any gap regime, any flag, alphabet and batch size changing from
    call to call (meant for ONE long-lived context: what a call leaves in the pooled buffers is part of the next call's environment)"""
    kind = rng.random()
    if kind < 0.4:
        n, nc, mat = 5, 4, _dna_matrix(int(rng.integers(1, 4)), int(rng.integers(1, 6)))
    elif kind < 0.6:
        n, nc, mat = 24, 20, _blosum50()
    else:
        n = int(rng.integers(4, 33)); nc = n - 1
        mat = np.ascontiguousarray(rng.integers(-20, 21, size=(n, n)).astype(np.int8).reshape(-1))
    g = rng.random()
    if g < 0.45:
        gapE = int(rng.integers(1, 4)); gapO = gapE + int(rng.integers(1, 9))
    elif g < 0.85:
        gapO = int(rng.integers(0, 6)); gapE = gapO + int(rng.integers(0, 4))
    else:
        gapO = 0; gapE = int(rng.integers(0, 3))
    nt = 1 if rng.random() < 0.6 else int(rng.integers(2, 8))
    refs = [rng.integers(0, nc, size=0 if rng.random() < 0.12 else int(rng.integers(1, 400)), dtype=np.int8) for _ in range(nt)]
    base = max(refs, key=len)
    nq = int(rng.integers(1, 12))
    lens = [0 if rng.random() < 0.2 else int(rng.integers(1, 500 if rng.random() < 0.15 else 120)) for _ in range(nq)]
    reads = [np.zeros(0, dtype=np.int8) if L == 0 else make_reads(rng, base, 1, [L], nc, frac_random=0.3)[0] for L in lens]
    return (reads, refs, mat, n, gapO, gapE, int(rng.choice([0, 0, 1, 2, 8, 9, 15, 6])), int(rng.choice([0, 0, 30])), int(rng.choice([0, 40, 1000])),
            int(rng.choice([-1, -1, 0, 15, 40])), int(rng.choice([2, 2, 2, 0, 1])))


def early_team_batch(rng, nreads=24, rlen=300):
    """a batch for the traceback's EARLY teams (round 6): a third of the reads are wide from the start (one long deletion: |refLen' - readLen'| + 1 is far above
    round 0's band of 48 -> their teams are launched before round 0), a sixth start narrow and outgrow round 0's scratch (three insertions and three deletions of 20 bases
    that cancel in length: band0 is 1, the path needs a band above 60 -> pending after round 0, second phase), the others are plain.  -> (reads, ref)"""
    ref = _random_ref(10 * rlen, int(rng.integers(1 << 30)), 4)
    reads = make_reads(rng, ref, nreads, [rlen] * nreads, 4, sub=0.02, ins=0.004, dele=0.004, frac_random=0.0)
    for i in range(nreads):
        o = int(rng.integers(0, len(ref) - 3 * rlen))
        if i % 3 == 0:
            gap = int(rng.integers(60, 200))
            reads[i] = np.ascontiguousarray(np.concatenate([ref[o:o + rlen // 2], ref[o + rlen // 2 + gap:o + rlen + gap]])[:rlen])
        elif i % 6 == 1:      # three insertions of 20, then three deletions of 20: the ends line up (band0 = 1), the path strays 60 columns in between
            pc = rlen // 8
            parts, at = [], o
            for k in range(7):
                parts.append(ref[at:at + pc]); at += pc
                if k < 3: parts.append(rng.integers(0, 4, size=20, dtype=np.int8))
                elif k < 6: at += 20
            reads[i] = np.ascontiguousarray(np.concatenate(parts)[:rlen])
    return reads, ref
