// pb_math.hpp - correctly rounded sine, cosine and two-argument arctangent for the FAITHFUL float64 chain.
//
// Why: the reference reaches glibc's sin / cos (np.sin, np.cos, np.exp(1j * lon)) and glibc's atan2 (np.log(complex).imag) -
// SURVEY 2, primitive table.  glibc's results are correctly rounded except on a few inputs in a million; the device libm
// (OCML) is accurate to about an ulp, i.e. differs from glibc in the last bit on 10-20 % of the inputs.  Almost always that is
// invisible (a 1-ulp change flips a truncation with probability ~1e-12), but on the degenerate geometries where EVERY
// pre-truncation coordinate sits on an integer (identity and near-identity remaps) the last bit decides the texel: round 2
// measured 41 687 of 589 824 pixels one texel off on an identity remap, and tens to hundreds of ulp on rotated coordinate
// maps.  glibc's own algorithm (IBM Accurate Mathematical Library: table lookups and double-double corrections; third-party,
// sysdeps/ieee754/dbl-64/s_sin.c, e_atan2.c, not in /root/reference) is not restated here; instead these functions deliver
// the CORRECTLY ROUNDED value - double-double evaluation to ~2^-100, far beyond the 2^-53 of the result - which is what
// glibc returns wherever glibc is itself correctly rounded.  Measured against this container's glibc 2.35 on 2 x 10^7 random
// arguments each (oracle/check_math.cpp, tests/test_oracle_golden.py::test_device_math_agrees_with_glibc): see DESIGN.md 2.
//
// The code is plain C++ (compiled for the host by the check program, for gfx950 by hipcc); every fused operation is an
// explicit fma(), the build disables contraction.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define PB_MATH_FN __device__ static inline
#define PB_MATH_CONST __device__ static const
#else
#define PB_MATH_FN static inline
#define PB_MATH_CONST static const
#endif

#include "pb_math_tables.hpp"

struct pb_dd {
    double h, l;
};

PB_MATH_FN pb_dd pb_two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
PB_MATH_FN pb_dd pb_fast_two_sum(double a, double b) {  // |a| >= |b| (or a == 0)
    const double s = a + b;
    return {s, b - (s - a)};
}
PB_MATH_FN pb_dd pb_two_prod(double a, double b) {
    const double p = a * b;
    return {p, fma(a, b, -p)};
}
PB_MATH_FN pb_dd pb_dd_add(pb_dd a, pb_dd b) {
    pb_dd s = pb_two_sum(a.h, b.h);
    const pb_dd t = pb_two_sum(a.l, b.l);
    s.l += t.h;
    s = pb_fast_two_sum(s.h, s.l);
    s.l += t.l;
    return pb_fast_two_sum(s.h, s.l);
}
PB_MATH_FN pb_dd pb_dd_add_d(pb_dd a, double b) {
    pb_dd s = pb_two_sum(a.h, b);
    s.l += a.l;
    return pb_fast_two_sum(s.h, s.l);
}
PB_MATH_FN pb_dd pb_dd_neg(pb_dd a) { return {-a.h, -a.l}; }
PB_MATH_FN pb_dd pb_dd_mul(pb_dd a, pb_dd b) {
    pb_dd p = pb_two_prod(a.h, b.h);
    p.l += fma(a.h, b.l, a.l * b.h);
    return pb_fast_two_sum(p.h, p.l);
}
PB_MATH_FN pb_dd pb_dd_mul_d(pb_dd a, double b) {
    pb_dd p = pb_two_prod(a.h, b);
    p.l = fma(a.l, b, p.l);
    return pb_fast_two_sum(p.h, p.l);
}
// a / b to ~2^-102: ONE division (the reciprocal of b's head), the rest products and exact remainders
PB_MATH_FN pb_dd pb_dd_div(pb_dd a, pb_dd b) {
    const double inv = 1.0 / b.h;
    const double q1 = a.h * inv;
    // r = a - q1 * b, exactly in its head: fma(-q1, b.h, a.h) is exact (q1 is within 2 ulp of a.h / b.h)
    double r = fma(-q1, b.h, a.h);
    r += fma(-q1, b.l, a.l);
    const double q2 = r * inv;
    double r2 = fma(-q2, b.h, r);  // second remainder: q2's own rounding matters at the 2^-104 level only; one more term is cheap
    const double q3 = r2 * inv;
    const pb_dd q = pb_fast_two_sum(q1, q2);
    return pb_dd_add_d(q, q3);
}

// sin and cos of x as double-doubles (relative error < 2^-95); x finite, |x| < 2^19 (far beyond the chain's angles)
PB_MATH_FN void pb_sincos_dd(double x, pb_dd& s, pb_dd& c) {
    const double kd = rint(x * PB_TWO_OVER_PI_DD[0]);
    // r = x - kd * pi/2: the first two pieces of pi/2 have 33 bits, so their products with kd are exact
    pb_dd r = pb_two_sum(x, -kd * PB_PIO2_1);
    r = pb_dd_add_d(r, -kd * PB_PIO2_2);
    pb_dd t = pb_two_prod(kd, PB_PIO2_3H);
    t.l = fma(kd, PB_PIO2_3L, t.l);
    r = pb_dd_add(r, pb_dd_neg(t));
    const pb_dd z = pb_dd_mul(r, r);
    // sin r = r * S(z), S = sum (-1)^j z^j / (2j+1)!;  cos r = C(z), C = sum (-1)^j z^j / (2j)!   (|r| <= pi/4: z <= 0.617)
    // the terms from z^7 on contribute < 2^-40 of the sum: plain float64 there (their error: < 2^-93 of the sum)
    double sd = -PB_INV_FACT[27][0], cd = PB_INV_FACT[28][0];  // j = 13 of S, j = 14 of C
    cd = fma(cd, z.h, -PB_INV_FACT[26][0]);                    // j = 13 of C
    for (int j = 12; j >= 7; --j) {
        sd = fma(sd, z.h, (j & 1) ? -PB_INV_FACT[2 * j + 1][0] : PB_INV_FACT[2 * j + 1][0]);
        cd = fma(cd, z.h, (j & 1) ? -PB_INV_FACT[2 * j][0] : PB_INV_FACT[2 * j][0]);
    }
    pb_dd S = {sd, 0.0}, C = {cd, 0.0};
    for (int j = 6; j >= 0; --j) {
        const double sg = (j & 1) ? -1.0 : 1.0;
        S = pb_dd_add(pb_dd_mul(S, z), pb_dd{sg * PB_INV_FACT[2 * j + 1][0], sg * PB_INV_FACT[2 * j + 1][1]});
        C = pb_dd_add(pb_dd_mul(C, z), pb_dd{sg * PB_INV_FACT[2 * j][0], sg * PB_INV_FACT[2 * j][1]});
    }
    S = pb_dd_mul(S, r);
    const long long k = (long long)kd;
    switch ((int)(k & 3)) {
        case 0: s = S; c = C; break;
        case 1: s = C; c = pb_dd_neg(S); break;
        case 2: s = pb_dd_neg(S); c = pb_dd_neg(C); break;
        default: s = pb_dd_neg(C); c = S; break;
    }
}

// Correctly rounded (to nearest) sin / cos.  Non-finite and huge arguments keep the platform libm's behaviour.
PB_MATH_FN void pb_sincos_cr(double x, double* sn, double* cs) {
    if (!(fabs(x) < 524288.0)) {
        *sn = sin(x);
        *cs = cos(x);
        return;
    }
    if (fabs(x) < 0x1p-27) {  // sin x = x, cos x = 1 to well below half an ulp (and -0.0 stays -0.0)
        *sn = x;
        *cs = 1.0;
        return;
    }
    pb_dd s, c;
    pb_sincos_dd(x, s, c);
    *sn = s.h + s.l;
    *cs = c.h + c.l;
}
// Correctly rounded atan(x) (np.arctan is NumPy's SIMD path, itself correctly rounded on all but ~7 arguments in 10 000)
PB_MATH_FN double pb_atan2_cr(double y, double x);
PB_MATH_FN double pb_atan_cr(double x) { return (x == x && x != 0.0) ? pb_atan2_cr(x, 1.0) : x; }
PB_MATH_FN double pb_sin_cr(double x) {
    double s, c;
    pb_sincos_cr(x, &s, &c);
    return s;
}
PB_MATH_FN double pb_cos_cr(double x) {
    double s, c;
    pb_sincos_cr(x, &s, &c);
    return c;
}

// atan of a double-double t in (0, 1] as a double-double (relative error < 2^-95)
PB_MATH_FN pb_dd pb_atan_dd01(pb_dd t) {
    // t = c + (t - c), c = i / 64:  atan t = atan c + atan u,  u = (t - c) / (1 + t c),  |u| <= 2^-7
    const int i = (int)rint(t.h * 64.0);
    const double cc = (double)i * 0.015625;
    pb_dd u = t;
    if (i != 0) {
        const pb_dd num = pb_dd_add_d(t, -cc);
        const pb_dd den = pb_dd_add_d(pb_dd_mul_d(t, cc), 1.0);
        u = pb_dd_div(num, den);
    }
    const pb_dd w = pb_dd_mul(u, u);
    // atan u = u * A(w), A = sum (-1)^k w^k / (2k+1);  w <= 2^-14: terms from w^4 on in float64
    double ad = -PB_INV_ODD[13][0];
    for (int k = 12; k >= 4; --k) ad = fma(ad, w.h, (k & 1) ? -PB_INV_ODD[k][0] : PB_INV_ODD[k][0]);
    pb_dd A = {ad, 0.0};
    for (int k = 3; k >= 0; --k) {
        const double sg = (k & 1) ? -1.0 : 1.0;
        A = pb_dd_add(pb_dd_mul(A, w), pb_dd{sg * PB_INV_ODD[k][0], sg * PB_INV_ODD[k][1]});
    }
    pb_dd r = pb_dd_mul(A, u);
    if (i != 0) r = pb_dd_add(pb_dd{PB_ATAN_TAB[i][0], PB_ATAN_TAB[i][1]}, r);
    return r;
}

// Correctly rounded atan2(y, x) for finite non-zero arguments; zeros, infinities and NaNs take the platform libm (whose
// results there are exact constants or signed zeros).
PB_MATH_FN double pb_atan2_cr(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    if (!(ax < HUGE_VAL) || !(ay < HUGE_VAL) || ax == 0.0 || ay == 0.0) return atan2(y, x);
    // keep the quotient away from overflow / underflow (never near them in the remap chain, but be total)
    if (ax > 0x1p1000 || ay > 0x1p1000 || ax < 0x1p-900 || ay < 0x1p-900) return atan2(y, x);
    const bool swap = ay > ax;
    const pb_dd t = swap ? pb_dd_div(pb_dd{ax, 0.0}, pb_dd{ay, 0.0}) : pb_dd_div(pb_dd{ay, 0.0}, pb_dd{ax, 0.0});
    if (!swap && x > 0.0 && t.h < 0x1p-60) return y < 0.0 ? -(t.h + t.l) : (t.h + t.l);  // atan t = t to far below half an ulp
    pb_dd r = pb_atan_dd01(t);
    const pb_dd pio2 = {PB_PIO2_DD[0], PB_PIO2_DD[1]}, pi = {PB_PI_DD[0], PB_PI_DD[1]};
    if (swap) r = pb_dd_add_d(pb_dd_add(pio2, pb_dd_neg(r)), PB_PIO2_DD[2]);
    if (x < 0.0) r = pb_dd_add_d(pb_dd_add(pi, pb_dd_neg(r)), PB_PI_DD[2]);
    const double v = r.h + r.l;
    return y < 0.0 ? -v : v;
}
