// exp_libm.hip - per-function bit mismatches against NumPy's results (fixture.bin from make_inputs.py): the device libm (OCML)
// and photonbend_amd/csrc/pb_math.hpp.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o exp_libm exp_libm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../photonbend_amd/csrc/pb_math.hpp"
#define N 60000
__global__ void k(const double* in, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double x = in[i], a = in[N + i], y2 = in[2 * N + i], x2 = in[3 * N + i];
    double s, c;
    pb_sincos_cr(a, &s, &c);
    // rows: OCML sin cos atan2 atan acos asin tan | pb_math sin cos atan2 atan
    out[0 * N + i] = sin(a); out[1 * N + i] = cos(a); out[2 * N + i] = atan2(y2, x2); out[3 * N + i] = atan(4 * x);
    out[4 * N + i] = acos(x); out[5 * N + i] = asin(x); out[6 * N + i] = tan(1.5 * x);
    out[7 * N + i] = s; out[8 * N + i] = c; out[9 * N + i] = pb_atan2_cr(y2, x2); out[10 * N + i] = pb_atan_cr(4 * x);
}
int main() {
    std::vector<double> h(11 * N), o(11 * N);
    FILE* f = fopen("experiments/libm/fixture.bin", "rb");
    if (!f || fread(h.data(), 8, 11 * N, f) != 11 * N) { printf("no fixture\n"); return 1; }
    double *din, *dout;
    hipMalloc(&din, 4 * N * 8); hipMalloc(&dout, 11 * N * 8);
    hipMemcpy(din, h.data(), 4 * N * 8, hipMemcpyHostToDevice);
    k<<<(N + 255) / 256, 256>>>(din, dout);
    hipMemcpy(o.data(), dout, 11 * N * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"sin", "cos", "atan2", "atan", "acos", "asin", "tan"};
    for (int fn = 0; fn < 7; ++fn) {
        long bad = 0, bad2 = 0;
        for (int i = 0; i < N; ++i) {
            bad += memcmp(&o[fn * N + i], &h[(4 + fn) * N + i], 8) != 0;
            if (fn < 4) bad2 += memcmp(&o[(7 + fn) * N + i], &h[(4 + fn) * N + i], 8) != 0;
        }
        if (fn < 4) printf("%-6s device libm differs from NumPy on %6ld of %d (%.2f %%)   pb_math.hpp: %ld (%.3f %%)\n", names[fn], bad, N, 100.0 * bad / N, bad2, 100.0 * bad2 / N);
        else printf("%-6s device libm differs from NumPy on %6ld of %d (%.2f %%)\n", names[fn], bad, N, 100.0 * bad / N);
    }
    return 0;
}
