// pb_math.hpp - the transcendental functions of the FAITHFUL float64 chain: each one the very function the reference's NumPy runs on
// the platform that produced the goldens (x86-64 with FMA and AVX512_SKX, glibc 2.35, NumPy 2.2.6), restated operation for operation.
//
// Why not the device libm, and why not correctly rounded functions: SURVEY 2's primitive table sends np.sin / np.cos / np.exp(1j x) /
// np.log(z).imag to glibc and np.arcsin / np.arccos / np.arctan / np.tan to NumPy's own AVX-512 kernels (Intel SVML).  None of those
// is correctly rounded - 0.1 % (glibc) to 17 % (SVML) of their results are off the correctly rounded value by one ulp - and the device
// libm (OCML) differs from them on 3-27 % of the arguments.  Almost always that is invisible (a 1-ulp change flips a truncation with
// probability ~1e-12), but on the degenerate geometries where EVERY pre-truncation coordinate sits on an integer (identity and
// near-identity remaps) the last bit decides the texel, and materialised maps are compared in ulps.  History: round 2 (OCML) 41 687 of
// 589 824 pixels one texel off on an identity remap and up to 392 ulp on rotated maps; round 3 (correctly rounded sin / cos / atan2 /
// atan, Ziv two-step evaluation) 185 / 4 271 flips and 24 / 159 ulp; round 4 (this): 0 flips, every float64 map bit-identical.
//   pb_math_glibc.hpp   pb_sin_np, pb_cos_np (np.sin, np.cos), pb_expi_np (np.exp(x * 1j)), pb_arg_np (np.log(z).imag)
//   pb_math_np.hpp      pb_asin_np, pb_acos_np, pb_atan_np, pb_tan_np
// Both are plain C++ (compiled for the host by oracle/check_math.cpp, for gfx950 by hipcc); every fused operation is an explicit
// fma(), the build disables contraction (-ffp-contract=off).  Checked bit for bit against tests/golden/npmath.npz - NumPy's own result
// bits - by the host build (tests/test_oracle_golden.py) and by the gfx950 build (tests/test_hip_math.py).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define PB_MATH_FN __device__ static inline
#define PB_MATH_CONST __device__ static const
#else
#define PB_MATH_FN static inline
#define PB_MATH_CONST static const
#endif

#include "pb_math_np.hpp"
#include "pb_math_glibc.hpp"

// ---- the second math flavour (round 5) ------------------------------------------------------------------------------------------
// On an x86-64 host WITHOUT AVX512_SKX NumPy has no SIMD kernel for arcsin / arccos / arctan / tan and calls libm: the reference's bits
// there are glibc 2.35's asin / acos / atan / tan (the `_fma` builds), which differ from the SVML kernels on 8-17 % of the arguments.
// A build with -DPB_MATH_LIBM (libphotonbend_hip_libm.so; photonbend_amd/build.py) runs THOSE functions in the float64 chain -
// pb_math_libm.hpp, generated instruction by instruction from the machine code (gen_libm_flavour.py), pinned by
// tests/golden/npmath_libm.npz - and is otherwise the same library.  np.sin / np.cos / np.exp(1j x) / np.log(z).imag are glibc on both
// kinds of host: pb_math_glibc.hpp serves both flavours.  pb_asin_svml ... keep the first flavour's functions reachable for the bit checks.
#include "pb_math_libm.hpp"
PB_MATH_FN double pb_asin_svml(double x) { return pb_asin_np(x); }
PB_MATH_FN double pb_acos_svml(double x) { return pb_acos_np(x); }
PB_MATH_FN double pb_atan_svml(double x) { return pb_atan_np(x); }
PB_MATH_FN double pb_tan_svml(double x) { return pb_tan_np(x); }
#ifdef PB_MATH_LIBM
#define pb_asin_np pb_asin_libm
#define pb_acos_np pb_acos_libm
#define pb_atan_np pb_atan_libm
#define pb_tan_np pb_tan_libm
#define PB_MATH_FLAVOUR 1
#else
#define PB_MATH_FLAVOUR 0
#endif
