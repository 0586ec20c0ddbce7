"""The arithmetic facts the f16 form of the fill kernel rests on (csrc/lanes.h pkf_*), checked with numpy's
IEEE binary16 -- no GPU, no emulator: every integer k with |k| <= 2048 is exact as k/2048; sums and differences of such
values that stay in [-2048, 2048] are exact; the bit patterns of the non-negative ones are ordered like the integers (the
best-cell tracking compares packed halves as integers); and the 8-operation cell computes the same H matrix as the
9-operation int16 cell on random inputs (gapO > gapE)."""
import numpy as np


def h(k):
    return (np.asarray(k, dtype=np.float64) / 2048.0).astype(np.float16)


def test_scaled_integers_are_exact_and_closed_under_add():
    k = np.arange(-2048, 2049)
    assert (h(k).astype(np.float64) * 2048.0 == k).all()
    rng = np.random.default_rng(1)
    a = rng.integers(-2048, 2049, size=200000); b = rng.integers(-2048, 2049, size=200000)
    ok = np.abs(a + b) <= 2048
    s = (h(a[ok]) + h(b[ok])).astype(np.float64) * 2048.0            # one rounding in binary16: none happens
    assert (s == (a + b)[ok]).all()


def test_bit_patterns_of_non_negative_values_are_ordered():
    bits = h(np.arange(0, 2049)).view(np.uint16).astype(np.int64)
    assert (np.diff(bits) > 0).all() and bits[0] == 0


def test_f16_cell_equals_int16_cell_on_a_random_matrix():
    rng = np.random.default_rng(2)
    for _ in range(20):
        n, m = int(rng.integers(5, 40)), int(rng.integers(5, 60))
        s = rng.integers(-4, 6, size=(n, m))
        gE = int(rng.integers(1, 4)); gO = gE + int(rng.integers(1, 6))
        # int form: E and F extended from the F-free value h0 (csrc chain_rows)
        H0 = np.zeros((n + 1, m + 1), dtype=np.int64); E = np.zeros((n + 1, m + 2), dtype=np.int64)
        for j in range(1, m + 1):
            f = 0
            for i in range(1, n + 1):
                h0 = max(H0[i - 1, j - 1] + s[i - 1, j - 1], E[i, j], 0)
                H0[i, j] = max(h0, f)
                t0 = max(h0 - gO, 0)
                E[i, j + 1] = max(max(E[i, j] - gE, 0), t0)
                f = max(max(f - gE, 0), t0)
        # f16 form: gaps open from H, everything in binary16 with clamp to [0, 1]
        clamp = lambda x: np.float16(min(max(float(x), 0.0), 1.0))
        H1 = np.zeros((n + 1, m + 1), dtype=np.float16); Eh = np.zeros((n + 1, m + 2), dtype=np.float16)
        nO, nE = h(-gO), h(-gE)
        for j in range(1, m + 1):
            f = np.float16(0)
            for i in range(1, n + 1):
                a = clamp(H1[i - 1, j - 1] + h(s[i - 1, j - 1]))
                hh = max(a, Eh[i, j], f)
                H1[i, j] = hh
                t = clamp(hh + nO)
                Eh[i, j + 1] = max(clamp(Eh[i, j] + nE), t)
                f = max(clamp(f + nE), t)
        assert (H1.astype(np.float64) * 2048.0 == H0).all()
