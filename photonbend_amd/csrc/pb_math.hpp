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
// the CORRECTLY ROUNDED value, which is what glibc returns wherever glibc is itself correctly rounded.  Two steps (Ziv's
// strategy): a table-driven evaluation good to 2^-69 decides the rounding of all but ~1 result in 10^4 (pb_sincos_fast_reduced,
// pb_atan_fast); only where the value sits within 2^-67 of a rounding boundary the double-double series (~2^-100) runs, as a
// real call, so that it costs the callers neither registers nor code size.  The bits are the correctly rounded ones either way.  Measured against this container's glibc 2.35 on 2 x 10^7 random
// arguments each (oracle/check_math.cpp, tests/test_oracle_golden.py::test_device_math_agrees_with_glibc): see DESIGN.md 2.
//
// The code is plain C++ (compiled for the host by the check program, for gfx950 by hipcc); every fused operation is an
// explicit fma(), the build disables contraction.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define PB_MATH_FN __device__ static inline
#ifdef PB_MATH_INLINE_SLOW  // A/B builds only (experiments/session_r3_u.sh)
#define PB_MATH_SLOW __device__ static inline
#else
#define PB_MATH_SLOW __device__ static __attribute__((noinline))  // the rarely taken double-double paths: real calls, out of the callers' register budgets
#endif
#define PB_MATH_CONST __device__ static const
#else
#define PB_MATH_FN static inline
#define PB_MATH_SLOW static
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

// 1 / x for the fast paths, whose quotients are corrected with an exact remainder (so 2^-47 is plenty): on the device the hardware
// estimate and one Newton step - a full IEEE float64 division is ~25 instructions there; on the host the division itself.  The
// RESULTS of the functions below are the correctly rounded ones either way (tests/test_hip_math.py compares the two builds bit for
// bit).
#if defined(__HIP_DEVICE_COMPILE__)
PB_MATH_FN double pb_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
PB_MATH_FN float pb_quotf(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#else
PB_MATH_FN double pb_rcp(double x) { return 1.0 / x; }
PB_MATH_FN float pb_quotf(float a, float b) { return a / b; }
#endif

// x = kd * pi/2 + r, |r| <= pi/4 (+ an ulp), r a double-double good to ~2^-105 of ITSELF; x finite, |x| < 2^19.
// Round 4 (the reduction was a third of the fast path): kd == 0 - a quarter of the chain's angles - has nothing to remove; otherwise
// x - kd P1 is exact (P1 has 33 bits, so kd P1 is exact, and it lies within a factor of two of x: Sterbenz), kd P2 is exact, and the
// 106-bit tail kd (P3H + P3L) comes off with one exact head difference and a float64 sum of the tails.
PB_MATH_FN pb_dd pb_reduce_pio2(double x, double& kd) {
    kd = rint(x * PB_TWO_OVER_PI_DD[0]);
    if (kd == 0.0) return {x, 0.0};
    const double a = fma(-kd, PB_PIO2_1, x);
    pb_dd r = pb_two_sum(a, -kd * PB_PIO2_2);
    pb_dd t = pb_two_prod(kd, PB_PIO2_3H);
    t.l = fma(kd, PB_PIO2_3L, t.l);
    pb_dd s = pb_two_sum(r.h, -t.h);
    s.l += r.l - t.l;
    return pb_fast_two_sum(s.h, s.l);
}

// sin and cos of the reduced argument r as double-doubles (relative error < 2^-95): the SLOW path
struct pb_dd_pair {
    pb_dd s, c;
};
PB_MATH_SLOW pb_dd_pair pb_sincos_dd_reduced(pb_dd r) {
    pb_dd S, C;
    const pb_dd z = pb_dd_mul(r, r);
    // sin r = r * S(z), S = sum (-1)^j z^j / (2j+1)!;  cos r = C(z), C = sum (-1)^j z^j / (2j)!   (|r| <= pi/4: z <= 0.617)
    // the terms from z^7 on contribute < 2^-40 of the sum: plain float64 there (their error: < 2^-93 of the sum)
    double sd = -PB_INV_FACT[27][0], cd = PB_INV_FACT[28][0];  // j = 13 of S, j = 14 of C
    cd = fma(cd, z.h, -PB_INV_FACT[26][0]);                    // j = 13 of C
    for (int j = 12; j >= 7; --j) {
        sd = fma(sd, z.h, (j & 1) ? -PB_INV_FACT[2 * j + 1][0] : PB_INV_FACT[2 * j + 1][0]);
        cd = fma(cd, z.h, (j & 1) ? -PB_INV_FACT[2 * j][0] : PB_INV_FACT[2 * j][0]);
    }
    S = {sd, 0.0};
    C = {cd, 0.0};
    for (int j = 6; j >= 0; --j) {
        const double sg = (j & 1) ? -1.0 : 1.0;
        S = pb_dd_add(pb_dd_mul(S, z), pb_dd{sg * PB_INV_FACT[2 * j + 1][0], sg * PB_INV_FACT[2 * j + 1][1]});
        C = pb_dd_add(pb_dd_mul(C, z), pb_dd{sg * PB_INV_FACT[2 * j][0], sg * PB_INV_FACT[2 * j][1]});
    }
    S = pb_dd_mul(S, r);
    return {S, C};
}

// The FAST path: sin r and cos r as NORMALISED head + tail pairs with a relative error below 2^-69 (PB_FAST_REL leaves a
// factor of four).  |r| = j/256 + t, |t| <= 2^-9 (+ an ulp): sin(j/256), cos(j/256) from a double-double table, C * t and S * t
// as double-double products, the rest - S (cos t - 1) + C (sin t - t), below 2^-19 of the result - in float64.
#define PB_FAST_REL 0x1p-67
PB_MATH_FN void pb_sincos_fast_reduced(pb_dd r, pb_dd& S, pb_dd& C) {
    const double a = fabs(r.h), sg = r.h < 0.0 ? -1.0 : 1.0;
    const int j = (int)rint(a * 256.0);  // 0 .. 202
    // t = |r| - j/256: the subtraction of the heads is exact (a non-zero result is a multiple of ulp(r.h) > |r.l|)
    const pb_dd t = pb_fast_two_sum(a - (double)j * 0.00390625, sg * r.l);
    const double Sh = PB_SINCOS_TAB[j][0], Sl = PB_SINCOS_TAB[j][1], Ch = PB_SINCOS_TAB[j][2], Cl = PB_SINCOS_TAB[j][3];
    const double z = t.h * t.h;
    const double sc = t.h * (z * fma(z, fma(z, -0x1.a01a01a01a01ap-13, 0x1.1111111111111p-7), -0x1.5555555555555p-3));  // sin t - t
    const double cc = z * fma(z, fma(z, -0x1.6c16c16c16c17p-10, 0x1.5555555555555p-5), -0.5);                            // cos t - 1
    pb_dd p = pb_two_prod(Ch, t.h);  // C * t
    p.l += fma(Ch, t.l, Cl * t.h);
    pb_dd q = pb_two_prod(Sh, t.h);  // S * t
    q.l += fma(Sh, t.l, Sl * t.h);
    // sin |r| = S + C t + (S cc + C sc);  |S| >= |C t| for j >= 1, S = 0 for j = 0
    pb_dd s = pb_fast_two_sum(Sh, p.h);
    s.l += (Sl + p.l) + fma(Sh, cc, Ch * sc);
    s = pb_fast_two_sum(s.h, s.l);
    // cos |r| = C - S t + (C cc - S sc)
    pb_dd c = pb_fast_two_sum(Ch, -q.h);
    c.l += (Cl - q.l) + fma(Ch, cc, -(Sh * sc));
    C = pb_fast_two_sum(c.h, c.l);
    S = {sg * s.h, sg * s.l};
}

// is RN(h + l) the same for every value within rel * |h| of h + l?  (h, l normalised: h == RN(h + l))
PB_MATH_FN bool pb_rounding_decided(pb_dd v, double* out) {
    const double e = fabs(v.h) * PB_FAST_REL;
    const double lo = v.h + (v.l - e), hi = v.h + (v.l + e);
    *out = lo;
    return lo == hi;
}

PB_MATH_FN void pb_quadrant(long long k, pb_dd S, pb_dd C, pb_dd& s, pb_dd& c) {
    switch ((int)(k & 3)) {
        case 0: s = S; c = C; break;
        case 1: s = C; c = pb_dd_neg(S); break;
        case 2: s = pb_dd_neg(S); c = pb_dd_neg(C); break;
        default: s = pb_dd_neg(C); c = S; break;
    }
}

// sin and cos of x as double-doubles (relative error < 2^-95); x finite, |x| < 2^19 (far beyond the chain's angles)
PB_MATH_FN void pb_sincos_dd(double x, pb_dd& s, pb_dd& c) {
    double kd;
    const pb_dd r = pb_reduce_pio2(x, kd);
    const pb_dd_pair v = pb_sincos_dd_reduced(r);
    pb_quadrant((long long)kd, v.s, v.c, s, c);
}

#ifdef PB_MATH_COUNT  // oracle/check_math.cpp: how often the fast paths leave the rounding undecided
static long pb_math_slow_sincos = 0, pb_math_slow_atan2 = 0;
#define PB_MATH_COUNT_SLOW(c) (++(c))
#else
#define PB_MATH_COUNT_SLOW(c) ((void)0)
#endif

// Correctly rounded (to nearest) sin / cos.  Non-finite and huge arguments keep the platform libm's behaviour.
// Two steps (Ziv): the fast evaluation decides the rounding of all but a few results in 10^4; the double-double series runs only
// for those - the bits are the correctly rounded ones either way.
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
    double kd;
    const pb_dd r = pb_reduce_pio2(x, kd);
    pb_dd S, C, s, c;
    pb_sincos_fast_reduced(r, S, C);
    pb_quadrant((long long)kd, S, C, s, c);
    const bool ok_s = pb_rounding_decided(s, sn), ok_c = pb_rounding_decided(c, cs);
    if (ok_s && ok_c) return;
    PB_MATH_COUNT_SLOW(pb_math_slow_sincos);
    const pb_dd_pair v = pb_sincos_dd_reduced(r);
    pb_quadrant((long long)kd, v.s, v.c, s, c);
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

// The FAST path of atan(num / den), 0 < num <= den: a NORMALISED head + tail pair, relative error below 2^-69.
// atan(num / den) = atan(c) + atan(u), c = i / 256 the table point next to the quotient, u = (num - c den) / (den + c num) formed
// from the ARGUMENTS (one reciprocal in all), |u| <= 2^-9 (+): atan u = u + u^3 (-1/3 + u^2 / 5 - u^4 / 7) with the bracket in float64.
PB_MATH_FN pb_dd pb_atan_fast(double num, double den) {
    int i = (int)rintf(256.0f * pb_quotf((float)num, (float)den));  // 0 .. 256; a neighbour of the best index serves as well
    i = i < 0 ? 0 : (i > 256 ? 256 : i);
    const double c = (double)i * 0.00390625;
    const pb_dd p = pb_two_prod(c, den), q = pb_two_prod(c, num);
    pb_dd N = pb_two_sum(num, -p.h);
    N = pb_fast_two_sum(N.h, N.l - p.l);
    pb_dd D = pb_fast_two_sum(den, q.h);  // den >= c num
    D = pb_fast_two_sum(D.h, D.l + q.l);
    const double inv = pb_rcp(D.h);
    const double u1 = N.h * inv;
    double rem = fma(-u1, D.h, N.h);  // the remainder of u1 (exact to 2^-100 of N: u1 is within 2^-47 of the quotient)
    rem = fma(-u1, D.l, rem) + N.l;
    const double u2 = rem * inv;
    const double w = u1 * u1;
    const double corr = u1 * (w * fma(w, fma(w, -0x1.2492492492492p-3, 0x1.999999999999ap-3), -0x1.5555555555555p-2));
    pb_dd r = pb_fast_two_sum(PB_ATAN_TAB256[i][0], u1);  // atan(c) >= |u| for i >= 1, 0 for i = 0
    r.l += (PB_ATAN_TAB256[i][1] + u2) + corr;
    return pb_fast_two_sum(r.h, r.l);
}

// the double-double evaluation of atan2 (num, den = the smaller and the larger of |y|, |x|)
PB_MATH_SLOW double pb_atan2_slow(double y, double x, double num, double den, bool swap) {
    const pb_dd pio2 = {PB_PIO2_DD[0], PB_PIO2_DD[1]}, pi = {PB_PI_DD[0], PB_PI_DD[1]};
    const pb_dd t = pb_dd_div(pb_dd{num, 0.0}, pb_dd{den, 0.0});
    if (!swap && x > 0.0 && t.h < 0x1p-60) return y < 0.0 ? -(t.h + t.l) : (t.h + t.l);  // atan t = t to far below half an ulp
    pb_dd r = pb_atan_dd01(t);
    if (swap) r = pb_dd_add_d(pb_dd_add(pio2, pb_dd_neg(r)), PB_PIO2_DD[2]);
    if (x < 0.0) r = pb_dd_add_d(pb_dd_add(pi, pb_dd_neg(r)), PB_PI_DD[2]);
    const double v = r.h + r.l;
    return y < 0.0 ? -v : v;
}

// Correctly rounded atan2(y, x) for finite non-zero arguments; zeros, infinities and NaNs take the platform libm (whose
// results there are exact constants or signed zeros).  Two steps like pb_sincos_cr.
PB_MATH_FN double pb_atan2_cr(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    if (!(ax < HUGE_VAL) || !(ay < HUGE_VAL) || ax == 0.0 || ay == 0.0) return atan2(y, x);
    // keep the quotient away from overflow / underflow (never near them in the remap chain, but be total)
    if (ax > 0x1p1000 || ay > 0x1p1000 || ax < 0x1p-900 || ay < 0x1p-900) return atan2(y, x);
    const bool swap = ay > ax;
    const double num = swap ? ax : ay, den = swap ? ay : ax;
    const pb_dd pio2 = {PB_PIO2_DD[0], PB_PIO2_DD[1]}, pi = {PB_PI_DD[0], PB_PI_DD[1]};
    if (den < 0x1p100 && den > 0x1p-100 && num > den * 0x1p-40) {
        pb_dd r = pb_atan_fast(num, den);
        if (swap) {  // pi/2 - r
            pb_dd s = pb_two_sum(pio2.h, -r.h);
            r = pb_fast_two_sum(s.h, s.l + (pio2.l - r.l));
        }
        if (x < 0.0) {  // pi - r
            pb_dd s = pb_two_sum(pi.h, -r.h);
            r = pb_fast_two_sum(s.h, s.l + (pi.l - r.l));
        }
        double v;
        if (pb_rounding_decided(r, &v)) return y < 0.0 ? -v : v;
        PB_MATH_COUNT_SLOW(pb_math_slow_atan2);
    }
    return pb_atan2_slow(y, x, num, den, swap);
}

#include "pb_math_np.hpp"  // NumPy's own (SVML) arcsin / arccos / arctan / tan, bit for bit
#include "pb_math_glibc.hpp"  // glibc 2.35's sin / cos / sincos / atan2 as NumPy reaches them, bit for bit
