// pb_math_np.hpp - NumPy's float64 arcsin, arccos, arctan and tan, bit for bit, for the FAITHFUL float64 chain.
// SPDX-License-Identifier: BSD-3-Clause  (restates Intel SVML kernels as vendored by NumPy; see NOTICE)
//
// Why: the reference calls np.arccos (rotation.py:158), np.arcsin (lens.py:210, :261, :307), np.arctan (lens.py:71, :122) and np.tan
// (lens.py:95, :101, :143).  On an AVX512_SKX machine NumPy 2.2.6 does not reach libm for these: its loops
// (numpy/_core/src/umath/loops_umath_fp.dispatch.c.src) call Intel SVML's `__svml_asin8_ha`, `__svml_acos8_ha`, `__svml_atan8_ha`,
// `__svml_tan8_ha`, which NumPy vendors as assembly (third-party, the numpy/SVML submodule:
// linux/avx512/svml_z0_{asin,acos,atan,tan}_d_ha.s; not in /root/reference and not present here as source).  Those kernels are
// accurate to about 0.55 ulp but NOT correctly rounded - 8-17 % of their results differ from the correctly rounded value - so neither
// the device libm nor a correctly rounded function reproduces the reference's bits: round 3 measured 24 / 159 ulp on rotated
// coordinate maps (np.arccos near +-1) and thousands of one-texel flips on identity remaps.  This header restates the four kernels'
// main paths operation for operation: every multiply, add and fused multiply-add in the kernels' order (they are straight-line
// code), the same polynomial coefficients and lookup tables, and the two approximation INSTRUCTIONS the kernels start from,
// VRSQRT14PD and VRCP14PD, as tables sampled from the hardware (pb_np_tables.hpp, gen_np_tables.py).  Checked bit for bit against
// NumPy itself: tests/golden/npmath.npz (captured by oracle/make_goldens.py --npmath in the container whose NumPy made every
// other golden) through the host build (tests/test_oracle_golden.py) and the gfx950 build (tests/test_hip_math.py).
//
// Argument ranges: asin / acos are complete for |x| <= 1 (NaN outside, as NumPy); atan is complete; tan follows the kernel's main
// path, |x| <= 65536 (lens arguments are below pi) and defers to the platform's tan beyond it.
#pragma once
#include "pb_np_tables.hpp"

#if defined(__HIPCC__)
PB_MATH_FN unsigned long long pb_bits(double d) { return (unsigned long long)__double_as_longlong(d); }
PB_MATH_FN double pb_from_bits(unsigned long long u) { return __longlong_as_double((long long)u); }
PB_MATH_FN int pb_popc(unsigned v) { return __popc(v); }
#else
#include <cstring>
PB_MATH_FN unsigned long long pb_bits(double d) { unsigned long long u; std::memcpy(&u, &d, 8); return u; }
PB_MATH_FN double pb_from_bits(unsigned long long u) { double d; std::memcpy(&d, &u, 8); return d; }
PB_MATH_FN int pb_popc(unsigned v) { return __builtin_popcount(v); }
#endif

// bucket i of a delta-coded instruction table: the block's base minus the sum of the first (i & 15) two-bit steps
PB_MATH_FN unsigned pb_np_table(const unsigned short* base, const unsigned* step, unsigned i) {
    const unsigned w = step[i >> 4] & ((1u << (2 * (i & 15))) - 1u);
    return (unsigned)base[i >> 4] - (unsigned)pb_popc(w & 0x55555555u) - 2u * (unsigned)pb_popc(w & 0xAAAAAAAAu);
}
// VRSQRT14PD for a positive normal operand
PB_MATH_FN double pb_rsqrt14(double y) {
    const unsigned long long b = pb_bits(y), m = b & 0xFFFFFFFFFFFFFull;
    const int e = (int)(b >> 52) - 0x3ff, p = e & 1, h = (e - p) / 2;  // y = 4^h * (2^p * 1.m)
    if (m == 0 && p == 0) return pb_from_bits((unsigned long long)(0x3ff - h) << 52);
    return pb_from_bits(((unsigned long long)(0x3fe - h) << 52) | ((unsigned long long)pb_np_table(PB_RSQRT14_BASE, PB_RSQRT14_STEP, (unsigned)(p << 15) | (unsigned)(m >> 37)) << 36));
}
// VRCP14PD for a normal operand whose reciprocal is normal
PB_MATH_FN double pb_rcp14(double y) {
    const unsigned long long b = pb_bits(y), s = b & 0x8000000000000000ull, m = b & 0xFFFFFFFFFFFFFull;
    const int h = (int)((b >> 52) & 0x7ff) - 0x3ff;
    if (m == 0) return pb_from_bits(s | ((unsigned long long)(0x3ff - h) << 52));
    return pb_from_bits(s | ((unsigned long long)(0x3fe - h) << 52) | ((unsigned long long)pb_np_table(PB_RCP14_BASE, PB_RCP14_STEP, (unsigned)(m >> 36)) << 36));
}

// ---- arcsin / arccos ------------------------------------------------------------------------------------------------------------
// Shared: for |x| < 1/2 the series in R = x^2; otherwise in R = Y = (1 - |x|) / 2 around 2 sqrt(Y), which the kernels get from
// VRSQRT14PD and one correction step in double-double (head Sh, correction D: 2 sqrt(Y) = Sh - D).
PB_MATH_FN double pb_np_asin_poly(double R) {  // R * P(R), P ~ (asin(sqrt R) / sqrt R - 1) / R
    const double R2 = R * R, R4 = R2 * R2;
    const double a = fma(0x1.43f44bfbc3baep-6, R, 0x1.a583395d45ed5p-8), c = fma(0x1.f1c72e13ad8bep-6, R, 0x1.6db6db3b445f8p-5);
    const double e = fma(0x1.1c6dcf538ad2ep-6, R, 0x1.6e89cebdefaddp-6);
    double b = fma(0x1.07520c70eb909p-5, R, -0x1.0fb17f7dbb0edp-6), d = fma(0x1.8f8dc2afccad6p-7, R, 0x1.c6dbbcb88bd57p-7);
    b = fma(R2, b, a);
    d = fma(R2, d, e);
    b = fma(R4, b, d);
    b = fma(R2, b, c);
    b = fma(R, b, 0x1.333333337e0dep-4);
    b = fma(R, b, 0x1.555555555529cp-3);
    return R * b;
}
PB_MATH_FN void pb_np_sqrt2(double Y, double& Sh, double& D) {
    const double rs = (Y < 0x1p-255) ? 0.0 : pb_rsqrt14(Y);
    const double Y2 = Y + Y;
    Sh = Y2 * rs;
    const double E = fma(rs * rs, Y2, -2.0), Sl = fma(rs, Y2, -Sh), SE = Sh * E;
    double q = fma(-0x1.18000993b24c3p-6, E, 0x1.400006f70d42dp-5);
    q = fma(E, q, -0x1.7fffffffffe97p-4);
    q = fma(E, q, 0x1.fffffffffff9dp-3);
    D = fma(SE, q, -Sl);
}
PB_MATH_FN double pb_acos_np(double x) {
    const double ax = fabs(x);
    if (!(ax <= 1.0)) return NAN;
    const unsigned long long sgn = pb_bits(x) & 0x8000000000000000ull;
    const double nx = -ax, Y = fma(0.5, nx, 0.5), X2 = nx * nx;
    const double R = (X2 < Y) ? X2 : Y;
    const bool big = !(R < Y), low = !(R < x);  // |x| >= 1/2; x on the negative side
    double Sh, D;
    pb_np_sqrt2(Y, Sh, D);
    const double Dp = big ? D : 0.0, RP = pb_np_asin_poly(R);
    const double H = big ? (low ? 0x1.921fb54442d18p+1 : 0.0) : 0x1.921fb54442d18p+0;    // pi, 0, pi / 2: heads ...
    const double L = big ? (low ? 0x1.1a62633145c07p-53 : 0.0) : 0x1.1a62633145c07p-54;  // ... and tails
    const double Zh = big ? Sh : nx;
    double t = fma(Zh - Dp, RP, pb_from_bits(pb_bits(L) ^ sgn) - Dp);
    t = t + Zh;
    return pb_from_bits(pb_bits(t) ^ sgn) + H;
}
PB_MATH_FN double pb_asin_np(double x) {
    const double ax = fabs(x);
    if (!(ax <= 1.0)) return NAN;
    const unsigned long long sgn = pb_bits(x) & 0x8000000000000000ull;
    const double Y = fma(-0.5, ax, 0.5), X2 = ax * ax;
    const double R = (X2 < Y) ? X2 : Y;
    const bool big = !(ax < 0.5);
    double Sh, D;
    pb_np_sqrt2(Y, Sh, D);
    const double RP = pb_np_asin_poly(R);
    const double H = 0x1.921fb54442d18p+0, A = H - Sh, err = Sh - (H - A);
    const double Z = big ? (D - Sh) : ax, Lo = big ? ((0x1.1a62633145c07p-54 + D) - err) : 0.0;
    const double t = (big ? A : ax) + fma(Z, RP, Lo);
    return pb_from_bits(pb_bits(t) ^ sgn);
}

// ---- arctan -----------------------------------------------------------------------------------------------------------------------
// atan|x| = atan(c) + atan((|x| - c) / (1 + c |x|)), c = |x| rounded to a quarter (|x| < 7.875), or pi / 2 - atan(1 / |x|) beyond;
// the quotient from VRCP14PD refined in double-double, then an odd polynomial.
PB_MATH_CONST double PB_NP_ATAN_HI[32] = {
    0x0.0p+0, 0x1.f5b75f92c80ddp-3, 0x1.dac670561bb4fp-2, 0x1.4978fa3269ee1p-1, 0x1.921fb54442d18p-1, 0x1.cac7c57846f9ep-1, 0x1.f730bd281f69bp-1, 0x1.0d38f2c5ba09fp+0,
    0x1.1b6e192ebbe44p+0, 0x1.270ef55a53a25p+0, 0x1.30b6d796a4da8p+0, 0x1.38d6a6ce13353p+0, 0x1.3fc176b7a8560p+0, 0x1.45b54837351a0p+0, 0x1.4ae10fc6589a5p+0, 0x1.4f68dea672617p+0,
    0x1.5368c951e9cfdp+0, 0x1.56f6f33a3e6a7p+0, 0x1.5a25052114e60p+0, 0x1.5d013c41adabdp+0, 0x1.5f97315254857p+0, 0x1.61f06c6a92b89p+0, 0x1.6414d44094c7cp+0, 0x1.660b02c736a06p+0,
    0x1.67d8863bc99bdp+0, 0x1.698213a9d5053p+0, 0x1.6b0bae830c070p+0, 0x1.6c78c7edeb195p+0, 0x1.6dcc57bb565fdp+0, 0x1.6f08f07435fecp+0, 0x1.7030cf9403197p+0, 0x1.7145eac2088a4p+0,
};
PB_MATH_CONST double PB_NP_ATAN_LO[32] = {
    0x0.0p+0, 0x1.8ab6e3cf7afbdp-57, 0x1.a2b7f222f65e2p-56, 0x1.2419a87f2a458p-56, 0x1.1a62633145c07p-55, 0x1.0dae13ad18a6bp-55, 0x1.007887af0cbbdp-56, -0x1.bd0dc231bfd70p-54,
    0x1.b1b466a88828ep-54, -0x1.a66b1af5f84fbp-54, 0x1.6254cb03bb199p-54, -0x1.12c77e8a80f5cp-55, -0x1.441a3bd3f1084p-59, 0x1.9e4a72eedacc4p-56, -0x1.3b03e8a27f555p-54, 0x1.934f9f2b0020ep-54,
    -0x1.96f47948a99f1p-54, -0x1.df6edd6f1ec3bp-56, 0x1.8c2d0c89de218p-56, 0x1.f82bba194dd5dp-54, -0x1.31151a43b51cap-55, -0x1.487d50bceb1a5p-55, -0x1.c5f60a65c7397p-54, -0x1.acb6afb332a0fp-56,
    -0x1.9b7bd2e1e8c9cp-54, -0x1.b9839085189e3p-54, -0x1.7d1ab82ffb70bp-54, 0x1.9239ad620ffe2p-54, -0x1.29c86447928e7p-54, -0x1.957a7170df016p-55, -0x1.cbe1896221608p-56, -0x1.fda5797b32a0bp-54,
};
PB_MATH_FN double pb_atan_np(double x) {
    if (x != x) return x;
    const double SH = 0x1.8p+50;  // adding it leaves round(4 |x|) in the low mantissa bits
    const double ax = fabs(x);
    const unsigned long long sgn = pb_bits(x) & 0x8000000000000000ull;
    const bool near = ax < 7.875;
    const double S = ax + SH, c = S - SH;
    const unsigned idx = (unsigned)(pb_bits(S) & 31u);
    const double num = near ? (ax - c) : -1.0;
    const double den = near ? fma(c, ax, 1.0) : ((ax < 0x1p+128) ? ax : 0x1p+128);
    const double denl = fma(c, ax, -(den - 1.0));
    const double r0 = pb_rcp14(den), e = fma(-r0, den, 1.0), r1 = fma(e, r0, r0), r2 = fma(e * e, r1, r1);
    const double q = r2 * num, dl = denl * r2, ee = fma(-r2, den, 1.0), ql = fma(r2, num, -q);
    double qc = fma(q, ee, ql);
    const double q2 = q * q;
    if (near) qc = fma(-dl, q, qc);
    const double Ah = near ? PB_NP_ATAN_HI[idx] : 0x1.921fb54442d18p+0, Al = near ? PB_NP_ATAN_LO[idx] : 0x1.1a62633145c07p-54;
    const double q4 = q2 * q2, q3 = q2 * q;
    double p = fma(0x1.2e9b9f5c4fe97p-4, q2, -0x1.74257c46790ccp-4);
    const double p2 = fma(0x1.c71bfeff916a0p-4, q2, -0x1.249248eef04dap-3), p3 = fma(0x1.999999998741ep-3, q2, -0x1.555555555554dp-2);
    const double lo = qc + Al, s = Ah + q;
    p = fma(q4, p, p2);
    const double qe = q - (s - Ah);
    p = fma(q4, p, p3);
    p = fma(q3, p, lo + qe);
    return pb_from_bits(pb_bits(p + s) ^ sgn);
}

// ---- tan --------------------------------------------------------------------------------------------------------------------------
// x = N pi / 16 + r (three-piece Cody-Waite reduction, r as head + tail), tan r from an odd polynomial in double-double,
// tan x = (tan r + T) / (1 - T tan r) with T = tan((N mod 16) pi / 16) from a table (head, tail; the pole entry is "minus huge"),
// the quotient from VRCP14PD refined in double-double.
PB_MATH_CONST double PB_NP_TAN_HI[16] = {
    -0x0.0p+0, 0x1.975f5e0553158p-3, 0x1.a827999fcef32p-2, 0x1.561b82ab7f990p-1, 0x1.0p+0, 0x1.7f218e25a7461p+0, 0x1.3504f333f9de6p+1, 0x1.41bfee2424771p+2,
    -0x1.fffffffffffffp+1023, -0x1.41bfee2424771p+2, -0x1.3504f333f9de6p+1, -0x1.7f218e25a7461p+0, -0x1.0p+0, -0x1.561b82ab7f990p-1, -0x1.a827999fcef32p-2, -0x1.975f5e0553158p-3,
};
PB_MATH_CONST double PB_NP_TAN_LO[16] = {
    -0x0.0p+0, 0x1.ef5d367441946p-61, 0x1.08b2fb1366ea9p-56, 0x1.7a8c52172b675p-55, 0x0.0p+0, 0x1.419fa6954928fp-54, 0x1.21165f626cdd5p-53, 0x1.10706fed37f0ep-55,
    -0x1.0p+971, -0x1.10706fed37f0ep-55, -0x1.21165f626cdd5p-53, -0x1.419fa6954928fp-54, 0x0.0p+0, -0x1.7a8c52172b675p-55, -0x1.08b2fb1366ea9p-56, -0x1.ef5d367441946p-61,
};
PB_MATH_FN double pb_tan_np(double x) {
    if (!(fabs(x) <= 0x1.000000e4db24cp+16)) return tan(x);  // the kernel's large-argument path (Payne-Hanek) is not restated
    const double SH = 0x1.8p+52, P1 = 0x1.921fb54442d18p-3, P2 = 0x1.1a62633000000p-57, P3 = 0x1.45c06e0e68948p-89;
    const double S = fma(x, 0x1.45f306dc9c883p+2, SH), N = S - SH;
    const unsigned idx = (unsigned)(pb_bits(S) & 15u);
    const double r1 = fma(-N, P1, x), r2 = fma(-N, P2, r1), r3 = fma(-N, P3, r2);
    const double rl = fma(-P2, N, r1 - r2) - fma(P3, N, r3 - r2);
    const double rr = r3 * r3, Th = PB_NP_TAN_HI[idx], Tl = PB_NP_TAN_LO[idx];
    double p = fma(0x1.25cccc7c9fa5dp-7, rr, 0x1.664ab664efba9p-6);
    p = fma(rr, p, 0x1.ba1ba489d25cap-5);
    p = fma(rr, p, 0x1.11111110b0802p-3);
    p = fma(rr, p, 0x1.55555555555dcp-2);
    const double c = fma(-rr, p * r3, -rl);       // -(r^3 P(r^2) + tail of r)
    const double th = r3 - c, tl = (r3 - th) - c;  // tan r, head and tail
    const double nh = th + Th, nl = ((th - (nh - Th)) + Tl) + tl;  // numerator
    const double dh = fma(-th, Th, 1.0);                            // denominator: dh - ndl
    const double ndl = fma(th, Tl, fma(tl, Th, fma(th, Th, dh - 1.0)));
    const double r0 = pb_rcp14(dh);
    const double e = fma(ndl, r0, fma(-dh, r0, 1.0)), rc = fma(e, r0, r0);
    const double q = nh * rc;
    const double d = fma(-q, ndl, fma(q, dh, -nh)) - nl;
    return fma(-rc, d, q);
}
