"""Shared comparison helpers: batch results of the library (product or emulated) vs the checkers."""
import numpy as np

from sswutil import RES_FIELDS, cigar_str, oracle_align, ref_align, ref_lib


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
