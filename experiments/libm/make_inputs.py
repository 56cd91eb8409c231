"""Inputs and NumPy's results (this host: numpy 2.2.6, glibc 2.35, AVX512) for experiments/libm/exp_libm.hip: per-function bit
mismatch rates of the device libm (OCML) and of pb_math.hpp against what the reference reaches."""
import numpy as np
rng = np.random.default_rng(7)
n = 60000
x = rng.uniform(-1, 1, n)
ang = rng.uniform(-np.pi, np.pi, n)
y2 = rng.integers(-4096, 4096, n) + 0.5
x2 = rng.integers(-4096, 4096, n) + 0.5
out = {
    'x': x, 'ang': ang, 'y2': y2, 'x2': x2,
    'sin': np.sin(ang), 'cos': np.cos(ang), 'atan2': np.log(x2 + 1j * y2).imag, 'atan': np.arctan(4 * x),
    'acos': np.arccos(x), 'asin': np.arcsin(x), 'tan': np.tan(1.5 * x),
}
with open('experiments/libm/fixture.bin', 'wb') as f:
    for k in ('x', 'ang', 'y2', 'x2', 'sin', 'cos', 'atan2', 'atan', 'acos', 'asin', 'tan'):
        f.write(np.ascontiguousarray(out[k], dtype=np.float64).tobytes())
print('written', n)
