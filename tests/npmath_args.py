"""The arguments of tests/golden/npmath.npz, from integer arithmetic alone (splitmix64 on a counter) so that the generator
(oracle/make_goldens.py --npmath, which stores NumPy's results for them) and the tests build exactly the same float64 values on any
machine and any NumPy.  Families per function: the whole domain, the neighbourhoods where the kernels switch branches or tables
(|x| = 1/2 and 1 for arcsin / arccos, the quarters and 7.875 for arctan, the multiples of pi / 16 and the poles for tan), tiny
values, both signs, and the special values."""

import numpy as np

FUNCTIONS = ("arcsin", "arccos", "arctan", "tan")
N_PER_FUNCTION = 40_000


def _mix(counter: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (counter.astype(np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z


def _unit(n: int, seed: int) -> np.ndarray:
    """n float64 in [0, 1), 53 random bits each."""
    return (_mix(np.arange(n, dtype=np.uint64), seed) >> np.uint64(11)).astype(np.float64) * 2.0**-53


def _signs(n: int, seed: int) -> np.ndarray:
    return np.where(_mix(np.arange(n, dtype=np.uint64), seed) & np.uint64(1), -1.0, 1.0)


def _pow2(n: int, seed: int, lo: int, hi: int) -> np.ndarray:
    """2^k, k uniform in [lo, hi)."""
    k = (_mix(np.arange(n, dtype=np.uint64), seed) % np.uint64(hi - lo)).astype(np.int64) + lo
    return np.ldexp(1.0, k)


def arguments(fn: str) -> np.ndarray:
    n = N_PER_FUNCTION
    q = n // 8
    s = FUNCTIONS.index(fn) * 1000
    if fn in ("arcsin", "arccos"):
        special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.nextafter(0.5, 0), np.nextafter(0.5, 1), np.nextafter(1.0, 0), -np.nextafter(1.0, 0),
                            5e-324, -5e-324, 1e-310, 2.0 ** -27, 0.25, 0.75, 1.0 + 2.0 ** -52, -1.0 - 2.0 ** -52, 2.0, np.nan])
        parts = [
            2.0 * _unit(4 * q, s + 1) - 1.0,                                            # the whole domain
            (1.0 - _unit(q, s + 2) * _pow2(q, s + 3, -52, 0)) * _signs(q, s + 4),       # towards +-1: the rotated latitudes' hard end
            (0.5 + (2.0 * _unit(q, s + 5) - 1.0) * _pow2(q, s + 6, -50, -2)) * _signs(q, s + 7),  # around the branch at 1/2
            (2.0 * _unit(q, s + 8) - 1.0) * _pow2(q, s + 9, -300, 0),                   # small and tiny
            1.0 - 2.0 * _unit(q, s + 10) ** 2,                                          # denser towards +1, like cos(lat) near a pole
        ]
    elif fn == "arctan":
        special = np.array([0.0, -0.0, 0.125, 0.375, 7.875, -7.875, np.nextafter(7.875, 0), 0.25, 0.5, 1.0, -1.0, 31.0 / 4, 1e300, -1e300,
                            np.inf, -np.inf, 5e-324, 2.0 ** 128, 2.0 ** 129, np.nan])
        parts = [
            (2.0 * _unit(4 * q, s + 1) - 1.0) * 8.0,                                    # r and r / 2 of the lens inverses
            (2.0 * _unit(q, s + 2) - 1.0) * 64.0,
            (np.floor(_unit(q, s + 3) * 64.0) / 4.0 - 8.0) + (2.0 * _unit(q, s + 4) - 1.0) * _pow2(q, s + 5, -50, -3),  # around the quarters (+- 1/8 is the table switch)
            (2.0 * _unit(q, s + 6) - 1.0) * _pow2(q, s + 7, -300, 0),
            (1.0 + _unit(q, s + 8)) * _pow2(q, s + 9, 0, 300) * _signs(q, s + 10),      # large
        ]
    else:
        special = np.array([0.0, -0.0, np.pi / 2, -np.pi / 2, np.pi, np.pi / 4, np.pi / 32, 3 * np.pi / 32, 5e-324, 1e-310, 65536.0, -65536.0,
                            np.nextafter(np.pi / 2, 0), np.nextafter(np.pi / 2, 4), 1.0, -1.0, 0.5, 100.0, 3.0, np.nan])
        parts = [
            (2.0 * _unit(4 * q, s + 1) - 1.0) * np.pi,                                  # theta and theta / 2 of the lens forwards
            (2.0 * _unit(q, s + 2) - 1.0) * 65536.0,                                    # the whole main path
            (np.floor(_unit(q, s + 3) * 128.0) - 64.0) * (np.pi / 16) + (2.0 * _unit(q, s + 4) - 1.0) * _pow2(q, s + 5, -50, -4),  # around the table switches and poles
            (2.0 * _unit(q, s + 6) - 1.0) * _pow2(q, s + 7, -300, 0),
            _unit(q, s + 8) * (np.pi / 2),
        ]
    x = np.concatenate(parts)
    x = np.concatenate([x[: n - special.size], special])
    assert x.size == n
    return np.ascontiguousarray(x, dtype=np.float64)
