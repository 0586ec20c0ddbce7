"""The two facts the fill kernel's target tiling rests on, checked on the CPU models:
(1) restarting the recurrence from the all-zero state `halo` columns early reproduces every later column maximum
    exactly (halo = P + ceil(P*max(mat)/gapE) + 1), also on low-complexity inputs full of ties;
(2) a too-short halo does NOT (so the test is sensitive)."""
import numpy as np

from sswutil import _ptr, blosum50, dna_matrix, i8p, i32p, oracle_lib


def _colmax(O, ref, read, P, mat, n, gapO, gapE):
    out = np.zeros(len(ref), dtype=np.int32)
    O.orc_plain_fill(_ptr(ref, i8p), 0, len(ref), 0, _ptr(read, i8p), len(read), P, _ptr(mat, i8p), n, gapO, gapE, -1,
                     _ptr(out, i32p), None)
    return out


def test_exact_halo():
    O = oracle_lib()
    rng = np.random.default_rng(42)
    short_fail = 0
    for it in range(120):
        if it % 3 == 0:
            n, mat = 24, blosum50()
            ref = rng.integers(0, 20, size=int(rng.integers(300, 2500)), dtype=np.int8)
            read = ref[50:50 + int(rng.integers(8, 40))].copy()
        else:
            n, mat = 5, dna_matrix(int(rng.integers(1, 4)), int(rng.integers(1, 5)))
            unit = rng.integers(0, 4, size=int(rng.integers(1, 6)), dtype=np.int8)
            ref = np.tile(unit, 3000)[:int(rng.integers(500, 3000))].astype(np.int8)   # repeats: worst case for long paths
            ref[rng.random(len(ref)) < 0.02] = rng.integers(0, 4)
            read = np.tile(unit, 100)[:int(rng.integers(5, 60))].astype(np.int8)
        gapE = int(rng.integers(1, 3)); gapO = gapE + int(rng.integers(1, 5))
        P = (len(read) + 15) // 16 * 16
        full = _colmax(O, ref, read, P, mat, n, gapO, gapE)
        halo = O.orc_plain_halo(P, _ptr(mat, i8p), n, gapE)
        tile = int(rng.integers(16, 400))
        tiled = np.zeros(len(ref), dtype=np.int32)
        O.orc_plain_colmax_tiled(_ptr(ref, i8p), len(ref), _ptr(read, i8p), len(read), P, _ptr(mat, i8p), n, gapO, gapE, tile, halo,
                                 _ptr(tiled, i32p))
        assert (tiled == full).all()
        O.orc_plain_colmax_tiled(_ptr(ref, i8p), len(ref), _ptr(read, i8p), len(read), P, _ptr(mat, i8p), n, gapO, gapE, tile, P // 2,
                                 _ptr(tiled, i32p))
        short_fail += int((tiled != full).any())
    assert short_fail > 10
