"""The arguments of tests/golden/npmath.npz, from integer arithmetic alone (splitmix64 on a counter) so that the generator
(oracle/make_goldens.py --npmath, which stores NumPy's results for them) and the tests build exactly the same float64 values on any
machine and any NumPy.  Families per function: the whole domain, the neighbourhoods where the kernels switch branches or tables
(|x| = 1/2 and 1 for arcsin / arccos, the quarters and 7.875 for arctan, the multiples of pi / 16 and the poles for tan), tiny
values, both signs, and the special values."""

import numpy as np

FUNCTIONS = ("arcsin", "arccos", "arctan", "tan", "sin", "cos", "expi", "arg")
# "expi": np.exp(x * 1j) -> (imag, real) interleaved; "arg": np.log(x + 1j * y).imag of consecutive (y, x) pairs
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
    elif fn in ("sin", "cos", "expi"):
        # latitudes, longitudes and lens arguments; the branch points of glibc's sin / cos (2^-26, 0.126, 0.855469, 2.426265) and the
        # multiples of pi / 2 its reduction lands next to; up to the end of its Cody-Waite range (105 414 350)
        special = np.array([0.0, -0.0, 0.126, -0.126, 0.855469, 2.426265, np.pi, np.pi / 2, -np.pi / 2, 2 * np.pi, 2.0 ** -26, 2.0 ** -27, np.nextafter(2.0 ** -26, 0),
                            5e-324, 1e-310, 105414349.0, -105414349.0, 1.0, -1.0, 100.0])
        parts = [
            (2.0 * _unit(4 * q, s + 1) - 1.0) * np.pi,
            _unit(q, s + 2) * np.pi * 0.713,
            (np.floor(_unit(q, s + 3) * 16.0) - 8.0) * (np.pi / 2) + (2.0 * _unit(q, s + 4) - 1.0) * _pow2(q, s + 5, -50, -2),
            (2.0 * _unit(q, s + 6) - 1.0) * _pow2(q, s + 7, -300, 0),
            (2.0 * _unit(q, s + 8) - 1.0) * _pow2(q, s + 9, 0, 26),
        ]
    elif fn == "arg":
        # (y, x) pairs: the pixel-centre half-integers of a destination map, unit vectors after a rotation (any quadrant), extreme ratios
        # (|y / x| beyond 2^+-57), the octant lines and axes, scaled-tiny and scaled-huge pairs, the special values
        m = n // 2
        q = m // 8
        sp = [(0.0, 1.0), (-0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (0.0, 0.0), (-0.0, 0.0), (0.0, -0.0), (-0.0, -0.0), (1.0, 0.0), (-1.0, 0.0), (1.0, -0.0),
              (np.inf, np.inf), (-np.inf, np.inf), (np.inf, -np.inf), (-np.inf, -np.inf), (1.0, np.inf), (-1.0, -np.inf), (np.inf, 1.0), (1.0, 1.0), (1.0, -1.0),
              (-1.0, -1.0), (-1.0, 1.0), (0.0625, 1.0), (1.0, 0.0625), (1e-300, 1e-300), (1e300, -1e300), (1e-200, 1e200), (1e200, 1e-200), (-1e200, -1e-200), (np.nan, 1.0)]
        half = lambda k, sd: np.floor(_unit(k, sd) * 8192.0) - 4096.0 + 0.5
        ang_y, ang_x = 2.0 * _unit(2 * q, s + 3) - 1.0, 2.0 * _unit(2 * q, s + 4) - 1.0
        ys = [half(3 * q, s + 1), ang_y * _unit(2 * q, s + 5), (2.0 * _unit(q, s + 6) - 1.0) * _pow2(q, s + 7, -70, 70), (2.0 * _unit(q, s + 9) - 1.0) * _pow2(q, s + 10, -1000, 1000),
              np.floor(_unit(q, s + 13) * 64.0) - 32.0]
        xs = [half(3 * q, s + 2), ang_x * _unit(2 * q, s + 5), 2.0 * _unit(q, s + 8) - 1.0, (2.0 * _unit(q, s + 11) - 1.0) * _pow2(q, s + 12, -1000, 1000),
              np.floor(_unit(q, s + 14) * 64.0) - 32.0]
        y, x = np.concatenate(ys), np.concatenate(xs)
        sp = np.array(sp)
        y = np.concatenate([y[: m - len(sp)], sp[:, 0]])
        x = np.concatenate([x[: m - len(sp)], sp[:, 1]])
        assert y.size == x.size == m
        return np.ascontiguousarray(np.stack([y, x], axis=1).ravel(), dtype=np.float64)
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


def reference(fn: str, x: np.ndarray) -> np.ndarray:
    """The result BITS (uint64) of the NumPy expression `fn` stands for, as the reference writes it."""
    if fn == "expi":
        e = np.exp(x * 1j)  # rotation.py:130, projection.py:252
        return np.ascontiguousarray(np.stack([e.imag, e.real], axis=1).ravel()).view(np.uint64)
    if fn == "arg":
        yx = x.reshape(-1, 2)
        z = np.empty(yx.shape[0], dtype=np.complex128)  # make_complex (utils/__init__.py): real and imaginary parts set apart
        z.real, z.imag = yx[:, 1], yx[:, 0]
        return np.ascontiguousarray(np.log(z).imag).view(np.uint64)
    return np.ascontiguousarray(getattr(np, fn)(x)).view(np.uint64)
