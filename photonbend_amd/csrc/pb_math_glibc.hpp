// pb_math_glibc.hpp - the sin, cos, sincos and atan2 the reference's NumPy reaches in glibc 2.35, bit for bit, for the FAITHFUL chain.
// SPDX-License-Identifier: LGPL-2.1-or-later  (restates GNU C Library code: IBM Accurate Mathematical Library; see NOTICE)
//
// Which call reaches what (checked against NumPy 2.2.6 itself on 10^7 arguments each, oracle/make_goldens.py --npmath):
//   np.sin(x), np.cos(x)        libm's sin / cos through their ifunc: on a machine with FMA the `_fma` build of
//                               sysdeps/ieee754/dbl-64/s_sin.c, i.e. the C source with the compiler's contractions
//                               -> pb_sin_np, pb_cos_np
//   np.exp(1j * x)              libm's cexp -> the INTERNAL __sincos, which has no ifunc: the plain SSE2 build of s_sincos.c,
//                               every operation rounded on its own.  Its sine / cosine differ from np.sin / np.cos in the last bit
//                               on 0.07 % of the arguments                                          -> pb_expi_np
//   np.log(x + 1j * y).imag     libm's clog -> __ieee754_atan2 through its ifunc: the `_fma` build of e_atan2.c  -> pb_arg_np
// The algorithm is glibc's (IBM Accurate Mathematical Library; third-party, LGPL source not in /root/reference and not present here): a
// table of sin / cos at k / 128 plus short polynomials (results within 0.55 ulp, NOT correctly rounded: about 1 result in 1000 differs
// from the correctly rounded one, which is why round 3's correctly rounded functions left 185-296 one-texel flips on identity remaps
// and a handful of ulp on rotated maps), a three-piece Cody-Waite reduction, and for atan2 a division carried in double-double and a
// 241-row table of local expansions.  Restated here operation for operation from the published source, with the fused operations
// where the shipped x86-64 build has them (read off the machine code: every fma() below is one vfmadd/vfnmadd/vfmsub there, every
// plain product or sum a vmulsd / vaddsd), constants and tables as published (pb_glibc_tables.hpp).  Argument range: |x| < 105414350
// for sin / cos / sincos (beyond it glibc switches to a Payne-Hanek reduction, not restated: lens and map arguments are below 2 pi).
#pragma once
#include "pb_glibc_tables.hpp"

PB_MATH_CONST double PB_GL_S1 = -0x1.5555555555555p-3, PB_GL_S2 = 0x1.1111111110ecep-7, PB_GL_S3 = -0x1.a01a019db08b8p-13, PB_GL_S4 = 0x1.71de27b9a7ed9p-19,
                     PB_GL_S5 = -0x1.addffc2fcdf59p-26;
PB_MATH_CONST double PB_GL_SN3 = -0x1.5555555555515p-3, PB_GL_SN5 = 0x1.11110e829872fp-7, PB_GL_CS2 = 0.5, PB_GL_CS4 = -0x1.5555555555535p-5,
                     PB_GL_CS6 = 0x1.6c16bedd9e239p-10;
PB_MATH_CONST double PB_GL_BIG = 0x1.8p+45, PB_GL_HP0 = 0x1.921fb54442d18p+0, PB_GL_HP1 = 0x1.1a62633145c07p-54, PB_GL_MP1 = 0x1.921fb58000000p+0,
                     PB_GL_MP2 = -0x1.dde973c000000p-27, PB_GL_PP3 = -0x1.cb3b398000000p-55, PB_GL_PP4 = -0x1.d747f23e32ed7p-83,
                     PB_GL_HPINV = 0x1.45f306dc9c883p-1, PB_GL_TOINT = 0x1.8p+52;

// FMA = true: the contractions of the `_fma` build (sin, cos); false: the plain build (sincos).  The expressions are s_sin.c's.
template <bool FMA>
PB_MATH_FN double pb_gl_taylor_sin(double x, double dx) {
    const double xx = x * x;
    if (FMA) {
        double p = fma(xx, PB_GL_S5, PB_GL_S4);
        p = fma(xx, p, PB_GL_S3);
        p = fma(xx, p, PB_GL_S2);
        p = fma(xx, p, PB_GL_S1);
        return x + fma(xx, fma(p, x, -(0.5 * dx)), dx);
    }
    const double p = ((((PB_GL_S5 * xx + PB_GL_S4) * xx + PB_GL_S3) * xx + PB_GL_S2) * xx) + PB_GL_S1;
    return x + ((p * x - 0.5 * dx) * xx + dx);
}
template <bool FMA>
PB_MATH_FN double pb_gl_do_sin(double x, double dx) {
    const double xold = x;
    if (fabs(x) < 0.126) return pb_gl_taylor_sin<FMA>(x, dx);
    if (x <= 0) dx = -dx;
    const double u = PB_GL_BIG + fabs(x);
    x = fabs(x) - (u - PB_GL_BIG);
    const double* t = &PB_GL_SINCOSTAB[4 * (int)(unsigned)pb_bits(u)];
    const double sn = t[0], ssn = t[1], cs = t[2], ccs = t[3], xx = x * x;
    double cor;
    if (FMA) {
        const double s = x + fma(x * xx, fma(xx, PB_GL_SN5, PB_GL_SN3), dx);
        const double c = fma(x, dx, xx * fma(xx, fma(xx, PB_GL_CS6, PB_GL_CS4), PB_GL_CS2));
        cor = fma(s, cs, fma(-c, sn, fma(s, ccs, ssn)));
    } else {
        const double s = x + (dx + x * xx * (PB_GL_SN3 + xx * PB_GL_SN5));
        const double c = x * dx + xx * (PB_GL_CS2 + xx * (PB_GL_CS4 + xx * PB_GL_CS6));
        cor = (ssn + s * ccs - sn * c) + cs * s;
    }
    return copysign(sn + cor, xold);
}
template <bool FMA>
PB_MATH_FN double pb_gl_do_cos(double x, double dx) {
    if (x < 0) dx = -dx;
    const double u = PB_GL_BIG + fabs(x);
    x = fabs(x) - (u - PB_GL_BIG) + dx;
    const double* t = &PB_GL_SINCOSTAB[4 * (int)(unsigned)pb_bits(u)];
    const double sn = t[0], ssn = t[1], cs = t[2], ccs = t[3], xx = x * x;
    double cor;
    if (FMA) {
        const double s = fma(x * xx, fma(xx, PB_GL_SN5, PB_GL_SN3), x);
        const double c = xx * fma(xx, fma(xx, PB_GL_CS6, PB_GL_CS4), PB_GL_CS2);
        cor = fma(-s, sn, fma(-c, cs, fma(-s, ssn, ccs)));
    } else {
        const double s = x + x * xx * (PB_GL_SN3 + xx * PB_GL_SN5);
        const double c = xx * (PB_GL_CS2 + xx * (PB_GL_CS4 + xx * PB_GL_CS6));
        cor = (ccs - s * ssn - cs * c) - sn * s;
    }
    return cs + cor;
}
template <bool FMA>
PB_MATH_FN int pb_gl_reduce(double x, double& a, double& da) {  // x = n pi / 2 + (a + da)
    double t, xn, y, t2, db, b;
    if (FMA) {
        t = fma(x, PB_GL_HPINV, PB_GL_TOINT);
        xn = t - PB_GL_TOINT;
        y = fma(-xn, PB_GL_MP2, fma(-xn, PB_GL_MP1, x));
        t2 = fma(-xn, PB_GL_PP3, y);
        db = fma(-PB_GL_PP3, xn, y - t2);
        b = fma(-xn, PB_GL_PP4, t2);
        db = db + fma(-xn, PB_GL_PP4, t2 - b);
    } else {
        t = x * PB_GL_HPINV + PB_GL_TOINT;
        xn = t - PB_GL_TOINT;
        y = (x - xn * PB_GL_MP1) - xn * PB_GL_MP2;
        double t1 = xn * PB_GL_PP3;
        t2 = y - t1;
        db = (y - t2) - t1;
        t1 = xn * PB_GL_PP4;
        b = t2 - t1;
        db += (t2 - b) - t1;
    }
    a = b;
    da = db;
    return (int)(pb_bits(t) & 3u);
}
template <bool FMA>
PB_MATH_FN double pb_gl_do_sincos(double a, double da, int n) {
    const double r = (n & 1) ? pb_gl_do_cos<FMA>(a, da) : pb_gl_do_sin<FMA>(a, da);
    return (n & 2) ? -r : r;
}

// np.sin
PB_MATH_FN double pb_sin_np(double x) {
    const unsigned k = (unsigned)(pb_bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e500000u) return x;
    if (k < 0x3feb6000u) return pb_gl_do_sin<true>(x, 0.0);                                                   // |x| < 0.855469
    if (k < 0x400368fdu) return copysign(pb_gl_do_cos<true>(PB_GL_HP0 - fabs(x), PB_GL_HP1), x);             // |x| < 2.426265
    if (k < 0x419921fbu) {
        double a, da;
        const int n = pb_gl_reduce<true>(x, a, da);
        return pb_gl_do_sincos<true>(a, da, n);
    }
    return sin(x);
}
// np.cos
PB_MATH_FN double pb_cos_np(double x) {
    const unsigned k = (unsigned)(pb_bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e400000u) return 1.0;
    if (k < 0x3feb6000u) return pb_gl_do_cos<true>(x, 0.0);
    if (k < 0x400368fdu) {
        const double y = PB_GL_HP0 - fabs(x), a = y + PB_GL_HP1, da = (y - a) + PB_GL_HP1;
        return pb_gl_do_sin<true>(a, da);
    }
    if (k < 0x419921fbu) {
        double a, da;
        const int n = pb_gl_reduce<true>(x, a, da);
        return pb_gl_do_sincos<true>(a, da, n + 1);
    }
    return cos(x);
}
// np.exp(x * 1j): *sn = imaginary part, *cs = real part.  (x * 1j has the imaginary part x * 1 + 0 * 0: a longitude of -0.0 enters as +0.0.)
PB_MATH_FN void pb_expi_np(double x, double* sn, double* cs) {
    x = x + 0.0;
    const unsigned k = (unsigned)(pb_bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x400368fdu) {
        if (k < 0x3e400000u) {
            *sn = x;
            *cs = 1.0;
        } else if (k < 0x3feb6000u) {
            *sn = pb_gl_do_sin<false>(x, 0.0);
            *cs = pb_gl_do_cos<false>(x, 0.0);
        } else {
            const double y = PB_GL_HP0 - fabs(x), a = y + PB_GL_HP1, da = (y - a) + PB_GL_HP1;
            *sn = copysign(pb_gl_do_cos<false>(a, da), x);
            *cs = pb_gl_do_sin<false>(a, da);
        }
        return;
    }
    if (k < 0x419921fbu) {
        double a, da;
        const int n = pb_gl_reduce<false>(x, a, da);
        *sn = pb_gl_do_sincos<false>(a, da, n);
        *cs = pb_gl_do_sincos<false>(a, da, n + 1);
        return;
    }
    *sn = sin(x);
    *cs = cos(x);
}

// ---- atan2 (e_atan2.c, `_fma` build) ---------------------------------------------------------------------------------------------
PB_MATH_CONST double PB_GL_HPI = 0x1.921fb54442d18p+0, PB_GL_HPI1 = 0x1.1a62633145c07p-54, PB_GL_OPI = 0x1.921fb54442d18p+1, PB_GL_OPI1 = 0x1.1a62633145c07p-53,
                     PB_GL_QPI = 0x1.921fb54442d18p-1, PB_GL_TQPI = 0x1.2d97c7f3321d2p+1;
PB_MATH_FN double pb_gl_atan_poly(double v) {  // d3 + v (d5 + v (d7 + v (d9 + v (d11 + v d13))))
    return fma(v, fma(v, fma(v, fma(v, fma(v, 0x1.375f08b31cbcep-4, -0x1.7458022b13c25p-4), 0x1.c71c6e5129a3bp-4), -0x1.24924923f7603p-3), 0x1.99999999997fdp-3),
               -0x1.5555555555555p-2);
}
PB_MATH_FN const double* pb_gl_cij_row(double u) { return PB_GL_CIJ[(int)(fma(u, 256.0, 0x1p+52) - 0x1p+52) - 16]; }
PB_MATH_FN double pb_gl_cij_poly(const double* c, double v) { return fma(v, fma(v, fma(v, fma(v, c[6], c[5]), c[4]), c[3]), c[2]); }
// np.log(x + 1j * y).imag
PB_MATH_FN double pb_arg_np(double y, double x) {
    if (x != x || y != y) return x + y;
    if (y == 0.0) return (x < 0.0 || (x == 0.0 && (pb_bits(x) >> 63))) ? copysign(PB_GL_OPI, y) : y;
    if (x == 0.0) return (y > 0.0) ? PB_GL_HPI : -PB_GL_HPI;
    const bool xinf = fabs(x) == __builtin_inf(), yinf = fabs(y) == __builtin_inf();
    if (xinf) return yinf ? copysign(x > 0.0 ? PB_GL_QPI : PB_GL_TQPI, y) : copysign(x > 0.0 ? 0.0 : PB_GL_OPI, y);
    if (yinf) return copysign(PB_GL_HPI, y);
    double ax = (x < 0.0) ? -x : x, ay = (y < 0.0) ? -y : y;
    const int de = (int)((unsigned)(pb_bits(y) >> 32) & 0x7ff00000u) - (int)((unsigned)(pb_bits(x) >> 32) & 0x7ff00000u);
    if (de >= 59768832) return (y > 0.0) ? PB_GL_HPI : -PB_GL_HPI;  // |y / x| > 2^57
    if (de <= -59768832) return (x > 0.0) ? copysign(ay / ax, y) : ((y > 0.0) ? PB_GL_OPI : -PB_GL_OPI);
    if (ax < 0x1p-500 || ay < 0x1p-500) {
        ax *= 0x1p+500;
        ay *= 0x1p+500;
    }
    if (ax > 0x1p+500 || ay > 0x1p+500) {
        ax *= 0x1p-500;
        ay *= 0x1p-500;
    }
    // u + du = min / max, the quotient carried in double-double
    const bool flat = ax > ay;
    const double big = flat ? ax : ay, small = flat ? ay : ax;
    const double u = small / big, v = big * u, vv = fma(big, u, -v), du = ((small - v) - vv) / big;
    double z;
    if (x > 0.0) {
        if (flat) {  // (i) atan(ay / ax)
            if (u < 0.0625) {
                const double v2 = u * u;
                z = u + fma(u * v2, pb_gl_atan_poly(v2), du);
            } else {
                const double* c = pb_gl_cij_row(u);
                const double t3 = u - c[0], w = du + t3, dv = (fabs(t3) > fabs(du)) ? ((t3 - w) + du) : ((du - w) + t3);
                double zz = (w * w) * fma(w, fma(w, fma(w, c[6], c[5]), c[4]), c[3]);
                zz = fma(dv, c[2], zz);
                zz = fma(w, c[2], zz);
                z = zz + c[1];
            }
        } else if (u < 0.0625) {  // (ii) pi / 2 - atan(ax / ay)
            const double v2 = u * u, zz = (u * v2) * pb_gl_atan_poly(v2), t2 = PB_GL_HPI - u;
            const double cor = (PB_GL_HPI > fabs(u)) ? ((PB_GL_HPI - t2) - u) : (PB_GL_HPI - (u + t2));
            z = (((cor + PB_GL_HPI1) - du) - zz) + t2;
        } else {
            const double* c = pb_gl_cij_row(u);
            const double w = (u - c[0]) + du;
            z = (PB_GL_HPI - c[1]) + fma(-w, pb_gl_cij_poly(c, w), PB_GL_HPI1);
        }
    } else if (ay > ax) {  // (iii) pi / 2 + atan(ax / ay)
        if (u < 0.0625) {
            const double v2 = u * u, zz = (v2 * u) * pb_gl_atan_poly(v2), t2 = u + PB_GL_HPI;
            const double cor = (PB_GL_HPI > fabs(u)) ? ((PB_GL_HPI - t2) + u) : ((u - t2) + PB_GL_HPI);
            z = (((cor + PB_GL_HPI1) + du) + zz) + t2;
        } else {
            const double* c = pb_gl_cij_row(u);
            const double w = (u - c[0]) + du;
            z = (PB_GL_HPI + c[1]) + fma(w, pb_gl_cij_poly(c, w), PB_GL_HPI1);
        }
    } else if (u < 0.0625) {  // (iv) pi - atan(ay / ax)
        const double v2 = u * u, zz = (v2 * u) * pb_gl_atan_poly(v2), t2 = PB_GL_OPI - u;
        const double cor = (PB_GL_OPI > fabs(u)) ? ((PB_GL_OPI - t2) - u) : (PB_GL_OPI - (t2 + u));
        z = (((cor + PB_GL_OPI1) - du) - zz) + t2;
    } else {
        const double* c = pb_gl_cij_row(u);
        const double w = (u - c[0]) + du;
        z = (PB_GL_OPI - c[1]) + fma(-w, pb_gl_cij_poly(c, w), PB_GL_OPI1);
    }
    return copysign(z, y);
}
