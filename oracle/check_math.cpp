// check_math.cpp - TEST INFRASTRUCTURE (oracle/): the device math of photonbend_amd/csrc/pb_math.hpp compiled for the HOST and
// compared, bit for bit, with this machine's glibc (what the reference reaches through NumPy: SURVEY 2) and with the
// correctly rounded value (libquadmath, 113-bit).  Prints one line per function: arguments, mismatches against glibc,
// mismatches against correct rounding, glibc's own mismatches against correct rounding - and, for the two-step (Ziv) evaluation,
// how often the fast path left the rounding undecided and the largest relative error the fast path showed (its decision
// threshold PB_FAST_REL must stay well above that).
//   g++ -O2 -ffp-contract=off -mfma -o oracle/_ref/check_math oracle/check_math.cpp -lquadmath && oracle/_ref/check_math [n]
#include <quadmath.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define PB_MATH_COUNT
#include "../photonbend_amd/csrc/pb_math.hpp"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static double uni() { return (double)(rnd() >> 11) * 0x1p-53; }
static bool same(double a, double b) { return memcmp(&a, &b, 8) == 0 || (a != a && b != b); }

// --eval <fn> <in> <out>: evaluates the header's functions on the doubles of a binary file (fn 0: sin, cos of each value, interleaved;
// 1: atan2 of consecutive (y, x) pairs; 2: atan; 3..6: pb_math_np.hpp's asin, acos, atan, tan - NumPy's own;
// 7, 8: pb_math_glibc.hpp's np.sin, np.cos; 9: np.exp(1j x) as (imag, real) interleaved; 10: np.log(x + 1j y).imag of (y, x) pairs) - the host half of tests/test_hip_math.py, which compares the gfx950 build of the
// same header with this one bit for bit.
static int eval_file(int fn, const char* in, const char* out) {
    FILE* f = fopen(in, "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f) / 8;
    fseek(f, 0, SEEK_SET);
    double* x = (double*)malloc(8 * (n > 0 ? n : 1));
    if (fread(x, 8, n, f) != (size_t)n) return 2;
    fclose(f);
    FILE* g = fopen(out, "wb");
    if (!g) return 2;
    for (long i = 0; i < n; ++i) {
        double r[2];
        int k = 1;
        if (fn == 0) { pb_sincos_cr(x[i], &r[0], &r[1]); k = 2; }
        else if (fn == 1) { if (i & 1) continue; r[0] = (i + 1 < n) ? pb_atan2_cr(x[i], x[i + 1]) : 0.0; }
        else if (fn == 2) r[0] = pb_atan_cr(x[i]);
        else if (fn == 3) r[0] = pb_asin_np(x[i]);
        else if (fn == 4) r[0] = pb_acos_np(x[i]);
        else if (fn == 5) r[0] = pb_atan_np(x[i]);
        else if (fn == 6) r[0] = pb_tan_np(x[i]);
        else if (fn == 7) r[0] = pb_sin_np(x[i]);
        else if (fn == 8) r[0] = pb_cos_np(x[i]);
        else if (fn == 9) { pb_expi_np(x[i], &r[0], &r[1]); k = 2; }
        else { if (i & 1) continue; r[0] = (i + 1 < n) ? pb_arg_np(x[i], x[i + 1]) : 0.0; }
        fwrite(r, 8, k, g);
    }
    fclose(g);
    free(x);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 5 && !strcmp(argv[1], "--eval")) return eval_file(atoi(argv[2]), argv[3], argv[4]);
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    const double pi = 3.141592653589793;
    long bad_g = 0, bad_q = 0, g_q = 0;
    double max_rel = 0.0;
    // ---- sine / cosine: longitudes in [-pi, pi], latitudes in [0, pi], lens arguments (halves, 0.713 x), small values
    for (long i = 0; i < n; ++i) {
        double x;
        switch (i & 3) {
            case 0: x = (2.0 * uni() - 1.0) * pi; break;
            case 1: x = uni() * pi; break;
            case 2: x = uni() * pi * 0.713; break;
            default: x = ldexp(2.0 * uni() - 1.0, -(int)(rnd() % 40)); break;
        }
        double s, c;
        pb_sincos_cr(x, &s, &c);
        if (fabs(x) >= 0x1p-27) {  // the fast path's own error
            double kd;
            const pb_dd r = pb_reduce_pio2(x, kd);
            pb_dd S, C, fs, fc;
            pb_sincos_fast_reduced(r, S, C);
            pb_quadrant((long long)kd, S, C, fs, fc);
            const __float128 ts = sinq((__float128)x), tc = cosq((__float128)x);
            const double es = (double)fabsq((((__float128)fs.h + fs.l) - ts) / ts), ec = (double)fabsq((((__float128)fc.h + fc.l) - tc) / tc);
            if (es > max_rel) max_rel = es;
            if (ec > max_rel) max_rel = ec;
        }
        const double gs = sin(x), gc = cos(x);
        const double qs = (double)sinq((__float128)x), qc = (double)cosq((__float128)x);
        bad_g += !same(s, gs) + !same(c, gc);
        bad_q += !same(s, qs) + !same(c, qc);
        g_q += !same(gs, qs) + !same(gc, qc);
    }
    printf("sincos  n=%ld values=%ld  vs_glibc=%ld  vs_correctly_rounded=%ld  glibc_vs_correctly_rounded=%ld\n", n, 2 * n, bad_g, bad_q, g_q);
    printf("  fast path: undecided on %ld of %ld calls, largest relative error 2^%.1f (threshold 2^%.0f)\n", pb_math_slow_sincos, n, log2(max_rel), log2(PB_FAST_REL));
    max_rel = 0.0;
    // ---- atan2: pixel-centre offsets (half-integers) as a destination's mesh gives them, and unit-vector components as a
    // rotation gives them
    bad_g = bad_q = g_q = 0;
    for (long i = 0; i < n; ++i) {
        double y, x;
        if (i & 1) {
            y = (double)((long)(rnd() % 8192) - 4096) + 0.5;
            x = (double)((long)(rnd() % 8192) - 4096) + 0.5;
        } else {
            const double lat = uni() * pi, lon = (2.0 * uni() - 1.0) * pi;
            x = cos(lon) * sin(lat);
            y = sin(lon) * sin(lat);
            if (rnd() & 1) x *= uni();
        }
        {
            const double ax = fabs(x), ay = fabs(y), num = ay > ax ? ax : ay, den = ay > ax ? ay : ax;
            if (den < 0x1p100 && den > 0x1p-100 && num > den * 0x1p-40) {
                const pb_dd f = pb_atan_fast(num, den);
                const __float128 t = atanq((__float128)num / (__float128)den);
                const double e = (double)fabsq((((__float128)f.h + f.l) - t) / t);
                if (e > max_rel) max_rel = e;
            }
        }
        const double r = pb_atan2_cr(y, x), g = atan2(y, x), q = (double)atan2q((__float128)y, (__float128)x);
        bad_g += !same(r, g);
        bad_q += !same(r, q);
        g_q += !same(g, q);
    }
    printf("atan2   n=%ld values=%ld  vs_glibc=%ld  vs_correctly_rounded=%ld  glibc_vs_correctly_rounded=%ld\n", n, n, bad_g, bad_q, g_q);
    printf("  fast path: undecided on %ld of %ld calls, largest relative error 2^%.1f (threshold 2^%.0f)\n", pb_math_slow_atan2, n, log2(max_rel), log2(PB_FAST_REL));
    const long atan2_slow = pb_math_slow_atan2;
    // ---- atan (lens inverses): radii in focal-length units
    bad_g = bad_q = g_q = 0;
    for (long i = 0; i < n; ++i) {
        const double x = (i & 1) ? uni() * 4.0 : ldexp(uni(), -(int)(rnd() % 30));
        const double r = pb_atan_cr(x), g = atan(x), q = (double)atanq((__float128)x);
        bad_g += !same(r, g);
        bad_q += !same(r, q);
        g_q += !same(g, q);
    }
    printf("atan    n=%ld values=%ld  vs_glibc=%ld  vs_correctly_rounded=%ld  glibc_vs_correctly_rounded=%ld\n", n, n, bad_g, bad_q, g_q);
    printf("  fast path: undecided on %ld of %ld calls\n", pb_math_slow_atan2 - atan2_slow, n);
    // ---- special values keep the platform's results
    const double sp[] = {0.0, -0.0, 1.0, -1.0, 0.5, INFINITY, -INFINITY, NAN, 1e300, -1e-300, 4e-320};
    long bad_s = 0;
    for (double y : sp)
        for (double x : sp)
            if (!same(pb_atan2_cr(y, x), atan2(y, x))) {
                ++bad_s;
                printf("  atan2(%g, %g): %a vs glibc %a\n", y, x, pb_atan2_cr(y, x), atan2(y, x));
            }
    for (double x : sp) bad_s += !same(pb_atan_cr(x), atan(x));
    for (double x : sp) {
        double s, c;
        pb_sincos_cr(x, &s, &c);
        bad_s += !same(s, sin(x)) + !same(c, cos(x));
    }
    printf("special values: %ld mismatches against glibc\n", bad_s);
    return 0;
}
