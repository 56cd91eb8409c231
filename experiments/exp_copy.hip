// exp_copy.hip - the practical HBM ceiling of this box: plain device copies of 512 MiB in a few shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const u4* __restrict__ s, u4* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) d[i] = s[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_chunk(const u4* __restrict__ s, u4* __restrict__ d, size_t n) {
    // each workgroup owns a contiguous chunk of U * 256 u4
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (; base + (U - 1) * 256 < n; base += stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + base + u * 256) : s[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + base + u * 256); else d[base + u * 256] = v[u]; }
    }
}
int main() {
    const size_t bytes = 512ull << 20, n = bytes / 16;
    u4 *a, *b; CK(hipMalloc((void**)&a, bytes)); CK(hipMalloc((void**)&b, bytes)); CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; r++) { hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-44s %7.1f us  %.2f TB/s (read + write)\n", name, best * 1e3, 2.0 * bytes / best / 1e9);
        return 0;
    };
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        char nm[128];
        snprintf(nm, 128, "grid-stride U=1 plain, %d blocks", blocks); run(nm, [&] { k_copy<1, false><<<blocks, 256>>>(a, b, n); });
        snprintf(nm, 128, "grid-stride U=4 plain, %d blocks", blocks); run(nm, [&] { k_copy<4, false><<<blocks, 256>>>(a, b, n); });
        snprintf(nm, 128, "grid-stride U=4 nt, %d blocks", blocks); run(nm, [&] { k_copy<4, true><<<blocks, 256>>>(a, b, n); });
        snprintf(nm, 128, "chunk U=4 plain, %d blocks", blocks); run(nm, [&] { k_copy_chunk<4, false><<<blocks, 256>>>(a, b, n); });
        snprintf(nm, 128, "chunk U=8 nt, %d blocks", blocks); run(nm, [&] { k_copy_chunk<8, true><<<blocks, 256>>>(a, b, n); });
    }
    run("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
