"""Round 4: which machine code does each transcendental NumPy call of the reference's path run HERE?  Calls the candidates directly and
compares result bits with NumPy.  (Scratch: needs an AVX-512 x86-64 host; gcc -O2 -mavx512f -fPIC -shared svml_direct.c -o
/tmp/libdirect.so -ldl first.)  Findings (this container: Intel Sapphire Rapids, glibc 2.35, NumPy 2.2.6), 2^20 arguments each:
    np.arcsin / arccos / arctan / tan  == __svml_{asin,acos,atan,tan}8_ha (0 mismatches); != the `_la`-less twins (25-40 % differ)
    np.sin / np.cos                    == libm sin / cos (ifunc -> _fma build); != __svml_{sin,cos}8_ha (0.2 %)
    np.exp(x * 1j)                     != (np.sin, np.cos) on 0.07 % of x; == glibc's s_sincos.c evaluated WITHOUT contraction
                                       (cexp calls the internal __sincos, which has no ifunc: plain SSE2 build);
                                       ctypes' public `sincos` is the _fma build and differs from it likewise
    np.log(x + 1j y).imag              == libm atan2 (ifunc -> _fma build)
The operation order and fused operations of each were then read off `objdump -d` of the NumPy extension module / libm.so.6 and
restated in photonbend_amd/csrc/pb_math_np.hpp and pb_math_glibc.hpp; tests/golden/npmath.npz pins the result."""
import ctypes as C

import numpy as np
import numpy._core._multiarray_umath as m

lib = C.CDLL("/tmp/libdirect.so")
assert lib.init(m.__file__.encode()) == 0
libm = C.CDLL("libm.so.6")
rng = np.random.default_rng(1)


def svml(sym, x):
    o = np.empty_like(x)
    assert lib.call(sym, x.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), C.c_long(x.size)) == 0
    return o


def differ(a, b):
    return int((a.view(np.uint64) != b.view(np.uint64)).sum())


for fn, lo, hi in (("asin", -1, 1), ("acos", -1, 1), ("atan", -8, 8), ("tan", -1.6, 1.6), ("sin", -4, 4), ("cos", -4, 4)):
    x = rng.uniform(lo, hi, 1 << 20)
    ref = getattr(np, {"asin": "arcsin", "acos": "arccos", "atan": "arctan"}.get(fn, fn))(x)
    for suffix in ("8_ha", "8"):
        print(f"np.{fn}: vs __svml_{fn}{suffix}: {differ(svml(f'__svml_{fn}{suffix}'.encode(), x), ref)} of {x.size} differ")
x = rng.uniform(-np.pi, np.pi, 1 << 20)
e = np.exp(x * 1j)
print("np.exp(1j x).imag vs np.sin:", differ(np.ascontiguousarray(e.imag), np.sin(x)), " .real vs np.cos:", differ(np.ascontiguousarray(e.real), np.cos(x)))
