// check_math.cpp - TEST INFRASTRUCTURE (oracle/): the device math of photonbend_amd/csrc/pb_math.hpp compiled for the HOST, as an
// evaluator: `check_math <fn> <in> <out>` reads float64 arguments from a binary file and writes the header's results, so that the tests
// can compare them bit for bit with NumPy's own result bits (tests/golden/npmath.npz, tests/test_oracle_golden.py) and with what the
// gfx950 build of the same header returns on the device (tests/test_hip_math.py).
//   fn (the order of tests/npmath_args.py FUNCTIONS): 0 arcsin, 1 arccos, 2 arctan, 3 tan, 4 sin, 5 cos,
//   6 np.exp(x * 1j) -> (imag, real) interleaved, 7 np.log(x + 1j y).imag of consecutive (y, x) pairs;
//   8 arcsin, 9 arccos, 10 arctan, 11 tan of the SECOND math flavour (glibc's, what NumPy runs without AVX512_SKX: pb_math_libm.hpp,
//   tests/golden/npmath_libm.npz); 0-3 are always the first flavour's (NumPy's AVX-512 kernels)
//   g++ -O2 -ffp-contract=off -mfma -o oracle/_ref/check_math oracle/check_math.cpp
#include <cstdio>
#include <cstdlib>

#include "../photonbend_amd/csrc/pb_math.hpp"

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s <fn 0..11> <in> <out>\n", argv[0]);
        return 2;
    }
    const int fn = atoi(argv[1]);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fn < 0 || fn > 11) return 2;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f) / 8;
    fseek(f, 0, SEEK_SET);
    double* x = (double*)malloc(8 * (n > 0 ? n : 1));
    if (fread(x, 8, n, f) != (size_t)n) return 2;
    fclose(f);
    FILE* g = fopen(argv[3], "wb");
    if (!g) return 2;
    for (long i = 0; i < n; ++i) {
        double r[2];
        int k = 1;
        switch (fn) {
            case 0: r[0] = pb_asin_svml(x[i]); break;
            case 1: r[0] = pb_acos_svml(x[i]); break;
            case 2: r[0] = pb_atan_svml(x[i]); break;
            case 3: r[0] = pb_tan_svml(x[i]); break;
            case 8: r[0] = pb_asin_libm(x[i]); break;
            case 9: r[0] = pb_acos_libm(x[i]); break;
            case 10: r[0] = pb_atan_libm(x[i]); break;
            case 11: r[0] = pb_tan_libm(x[i]); break;
            case 4: r[0] = pb_sin_np(x[i]); break;
            case 5: r[0] = pb_cos_np(x[i]); break;
            case 6: pb_expi_np(x[i], &r[0], &r[1]); k = 2; break;
            default:
                if ((i & 1) || i + 1 >= n) continue;
                r[0] = pb_arg_np(x[i], x[i + 1]);
        }
        fwrite(r, 8, k, g);
    }
    fclose(g);
    free(x);
    return 0;
}
