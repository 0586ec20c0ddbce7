"""Seeded synthetic workloads of BASELINE.json's configs 2-5, shared by bench.py, scripts/make_expected.py
(reference results computed in the build container) and scripts/gpu_parity_full.py (the same inputs on the MI355X).

Every read set is generated in BLOCKS with their own seeds, so that "ONE read set sharded by read index" is literal:
block b is the same array whoever generates it (rank b of an N-GPU run, or a parity script looking at block 0).
Test infrastructure / benchmark input only -- nothing here is imported by the product library.
"""
import numpy as np

from sswutil import blosum50, dna_matrix, mutate, random_ref


def make_reads_fast(ref, nreads, length, seed, sub=0.03, ins=0.005, dele=0.005, frac_random=0.05):
    """vectorised read sampler: [nreads, length] int8 -- uniform offsets, per-base substitution / insertion / deletion,
    `frac_random` fully random reads"""
    rng = np.random.default_rng(seed)
    span = length + 32
    off = rng.integers(0, len(ref) - span, size=nreads)
    is_ins = rng.random((nreads, length)) < ins
    is_del = (rng.random((nreads, length)) < dele) & ~is_ins
    consumed = np.cumsum(~is_ins, axis=1) - 1
    deleted = np.cumsum(is_del, axis=1)
    src = off[:, None] + np.clip(consumed + deleted, 0, span - 1)
    reads = ref[src]
    rnd = rng.integers(0, 4, size=(nreads, length), dtype=np.int8)
    reads = np.where(is_ins, rnd, reads)
    do_sub = rng.random((nreads, length)) < sub
    reads = np.where(do_sub, (reads + 1 + rng.integers(0, 3, size=(nreads, length))) % 4, reads)
    whole = rng.random(nreads) < frac_random
    reads[whole] = rnd[whole]
    return np.ascontiguousarray(reads, dtype=np.int8)


# name -> parameters of the DNA configs.  `seed_ref` follows SURVEY 8d (config 2: seed 1, 3: seed 2, 4: seed 3).
DNA_CONFIGS = {
    2: dict(ref_len=1_000_000, seed_ref=1, reads=100_000, read_len=150, seed_reads=1000, sub=0.03, indel=0.005, flag=0, mask_len=-1,
            name="BASELINE config 2: 100k x 150 bp reads vs 1 Mb target"),
    3: dict(ref_len=5_000_000, seed_ref=2, reads=20_000, read_len=150, seed_reads=3000, sub=0.03, indel=0.005, flag=0, mask_len=-1,
            name="BASELINE config 3: 1M x 150 bp reads vs 5 Mb target, sharded by read block; timed subsample = 20k-read blocks"),
    4: dict(ref_len=100_000, seed_ref=3, reads=10_000, read_len=10_000, seed_reads=4000, sub=0.01, indel=0.0025, flag=2, mask_len=5000,
            name="BASELINE config 4: 10k x 10 kb reads vs 100 kb target, CIGAR on"),
}


def dna_config(cfg, block=0, reads=None):
    """-> (ref int8[ref_len], reads int8[n, read_len], params) of DNA config `cfg`, read block `block`"""
    p = dict(DNA_CONFIGS[cfg])
    if reads is not None:
        p["reads"] = int(reads)
    ref = random_ref(p["ref_len"], p["seed_ref"], 4)
    rd = make_reads_fast(ref, p["reads"], p["read_len"], seed=p["seed_reads"] + block, sub=p["sub"], ins=p["indel"], dele=p["indel"])
    return ref, rd, p


# The one workload the reference publishes a number for (README.md:62-74): 1000 Ion Torrent reads of 25-540 bp,
# "most reads are ~200 bp", against the 4 938 920-nt genome of E. coli 536 -- default penalties and -m1 -x3 -o5 -e2.  The data set itself
# is not in the tree: the same SHAPE from seeds -- a random genome of that length, read lengths 85 % N(200, 45) and 15 % uniform over
# [25, 540], Ion-Torrent-like errors (indels dominate: 1 % insertions, 1 % deletions, 0.5 % substitutions), 5 % unrelated reads.
MIXED_CONFIG = dict(ref_len=4_938_920, seed_ref=6, reads=1000, seed_reads=6000, sub=0.005, indel=0.01, min_len=25, max_len=540, flag=0, mask_len=-1,
                    name="config 6 (README.md:62-74 shape): 1000 reads of 25-540 bp (most ~200) vs a 4 938 920-nt genome")


def mixed_config(block=0, reads=None, ref_len=None):
    """-> (ref int8[ref_len], list of int8 reads of mixed lengths, params)"""
    p = dict(MIXED_CONFIG)
    if reads is not None:
        p["reads"] = int(reads)
    if ref_len is not None:
        p["ref_len"] = int(ref_len)
    ref = random_ref(p["ref_len"], p["seed_ref"], 4)
    rng = np.random.default_rng(p["seed_reads"] + 7919 * (block + 1))
    n = p["reads"]
    lens = np.where(rng.random(n) < 0.85, np.rint(rng.normal(200, 45, size=n)), rng.integers(p["min_len"], p["max_len"] + 1, size=n))
    lens = np.clip(lens, p["min_len"], min(p["max_len"], p["ref_len"] // 4)).astype(np.int64)
    full = make_reads_fast(ref, n, int(lens.max()), seed=p["seed_reads"] + block, sub=p["sub"], ins=p["indel"], dele=p["indel"])
    return ref, [np.ascontiguousarray(full[i, :lens[i]]) for i in range(n)], p


_AA_FREQ = np.array([8.3, 5.5, 4.1, 5.5, 1.4, 3.9, 6.8, 7.1, 2.3, 5.9, 9.7, 5.8, 2.4, 3.9, 4.7, 6.6, 5.3, 1.1, 2.9, 6.9])
_AA_FREQ = _AA_FREQ / _AA_FREQ.sum()


def _protein_set(rng, count):
    lens = np.clip(rng.normal(300, 60, size=count), 50, 1000).astype(np.int64)
    flat = rng.choice(20, size=int(lens.sum()), p=_AA_FREQ).astype(np.int8)
    off = np.zeros(count + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    return [flat[off[i]:off[i + 1]] for i in range(count)]


def protein_config(block=0, queries=50_000, db_entries=10_000):
    """BASELINE config 5: `queries` protein queries (len ~ N(300, 60) clipped to [50, 1000], background residue
    frequencies, 1 % planted homologs of DB entries) of query block `block` against a DB of `db_entries` (seed 4, the
    same for every block), BLOSUM50, gaps 3/1, maskLen = len/2, score only.  -> (db list, query list, mat)"""
    db = _protein_set(np.random.default_rng(4), db_entries)
    rng = np.random.default_rng(5000 + block)
    qs = _protein_set(rng, queries)
    for i in range(0, queries, 100):
        src = db[int(rng.integers(0, len(db)))]
        m = mutate(src, rng, 0.15, 0.02, 0.02, 20)
        if len(m) >= 50:
            qs[i] = np.ascontiguousarray(m[:1000])
    return db, qs, blosum50()


def pack(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return np.ascontiguousarray(np.concatenate(seqs).astype(np.int8)), off


def fnv1a_words(words):
    """FNV-1a over the little-endian bytes of uint32 words (what oracle/ref_wrap.c refwrap_bench_hash computes per CIGAR)"""
    h = 2166136261
    for b in np.asarray(words, dtype="<u4").tobytes():
        h = ((h ^ b) * 16777619) & 0xffffffff
    return h


def row_checksums(a):
    """[nq, k] int32 -> [nq] uint64: order-sensitive polynomial checksum of every row (config 5: one per query over its
    10k x 5 result fields) -- exact integer arithmetic mod 2^64"""
    a = np.ascontiguousarray(a).astype(np.uint64) & np.uint64(0xffffffff)
    h = np.zeros(a.shape[0], dtype=np.uint64)
    mul = np.uint64(1099511628211)
    with np.errstate(over="ignore"):
        for k in range(a.shape[1]):
            h = (h ^ a[:, k]) * mul
    return h


# ---- order-independent checksums over compact database-search records (ssw_gpu_hit: u16 score1, u16 score2, i32 ref_end1,
#      i32 read_end1, i32 ref_end2 = two little-endian 64-bit words): XOR of the words and a wrap-around sum of their
#      products with two odd constants.  Any partition of the records gives the same totals, so per-chunk values computed
#      while results stream by can be compared with values computed elsewhere in another order.
_K0 = np.uint64(0x9E3779B97F4A7C15)
_K1 = np.uint64(0xC2B2AE3D27D4EB4F)


def hit_words(score1, score2, ref_end1, read_end1, ref_end2):
    """field arrays -> (w0, w1) uint64 arrays, the two words of each 16-byte record"""
    u = lambda a: np.asarray(a).astype(np.int64).astype(np.uint64) & np.uint64(0xffffffff)
    w0 = (np.asarray(score1).astype(np.uint64) & np.uint64(0xffff)) | ((np.asarray(score2).astype(np.uint64) & np.uint64(0xffff)) << np.uint64(16)) | (u(ref_end1) << np.uint64(32))
    w1 = u(read_end1) | (u(ref_end2) << np.uint64(32))
    return w0, w1


def words_checksum(w0, w1):
    """-> (xor0, xor1, sum) as Python ints"""
    with np.errstate(over="ignore"):
        s = (w0 * _K0 + w1 * _K1).sum(dtype=np.uint64)
    return int(np.bitwise_xor.reduce(w0.ravel())), int(np.bitwise_xor.reduce(w1.ravel())), int(s)


def combine_checksums(a, b):
    return a[0] ^ b[0], a[1] ^ b[1], (a[2] + b[2]) & 0xffffffffffffffff
