// exp_calib.hip - calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of the remap kernels.
// Three kernels, each reads a 100.7 MB source exactly once per launch (6 sources in rotation: beyond the Infinity
// Cache) and writes nothing / a known amount:
//   calib_stream_x4    contiguous 16 B per lane (the guide's calibrated case: FETCH_SIZE reports 1/2)
//   calib_dma_rows96   LDS-DMA, 64-row x 96-B windows, rows 24 576 B apart, 16 B off line alignment (LEAN tiles)
//   calib_gather_dword unaligned 4-byte gathers, one per lane, 6 B apart along the rows of 32-row x 192-B windows (DIRECT tiles)
//   calib_store_12     12 B per lane stores of a 50.3 MB output in 32x32-pixel tiles (the kernels' stores)
// run under: rocprofv3 --kernel-trace --pmc FETCH_SIZE   and   --pmc WRITE_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));
extern __shared__ __attribute__((aligned(16))) unsigned lds[];
__global__ __launch_bounds__(256) void calib_stream_x4(const u4* __restrict__ s, unsigned* sink, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * 256) { const u4 v = s[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_dma_rows96(const uint8_t* __restrict__ src, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned w = blockIdx.x * 4 + wave;  // 16384 windows: 64 window rows x 256 windows per row
    unsigned* win = lds + wave * (64 * 96 / 4 + 16);
    const unsigned wr = w / 256u, wc = w % 256u;
    const unsigned gbase = wr * 64u * 24576u + wc * 96u + 16u;
    const unsigned lrow = (unsigned)lane / 6u, chunk = (unsigned)lane - lrow * 6u;
    for (unsigned rowb = 0; rowb < 64u; rowb += 10u) {
        const unsigned row = rowb + lrow;
        if (lrow < 10u && row < 64u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + gbase + row * 24576u + 16u * chunk),
                                             (__attribute__((address_space(3))) void*)(win + ((rowb * 96u) >> 2)), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (win[(lane * 37 + w) % (64 * 96 / 4)] == 0x12345678u) sink[0] = 1;
}
__global__ __launch_bounds__(256) void calib_gather_dword(const uint8_t* __restrict__ src, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned w = blockIdx.x * 4 + wave;  // 16384 windows of 32 rows x 192 B (128 x 128 of them = the whole source)
    const unsigned wr = w / 128u, wc = w % 128u;  // wr < 128: rows wr * 32 + 31 < 4096
    const unsigned gbase = wr * 32u * 24576u + wc * 192u;
    unsigned acc = 0;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const unsigned x = lane & 31, y = (lane >> 5) + 2u * n;  // y < 32; 32 samples per row, 6 B apart: all three sectors
        unsigned t;
        __builtin_memcpy(&t, src + gbase + y * 24576u + 6u * x, 4);
        acc ^= t;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_store_12(uint8_t* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned w = blockIdx.x * 4 + wave;
    const int tx = w & 127, ty = w >> 7;
    for (int jr = 0; jr < 4; ++jr) {
        const size_t off = 3ull * ((size_t)(ty * 32 + (lane >> 3) + 8 * jr) * 4096 + tx * 32 + 4 * (lane & 7));
        __builtin_nontemporal_store(u3{w, (unsigned)lane, (unsigned)jr}, reinterpret_cast<u3*>(dst + off));
    }
}
// sparse reads: one dword at byte `off` of every `stride`-byte block (stride 128, off 0: the first 64-B sector of every
// 128-B line; two launches with off 0 and off 64 in one kernel = both sectors).  Does the L2 fetch sectors or lines?
template <int STRIDE, int SECOND>
__global__ __launch_bounds__(256) void calib_sparse(const uint8_t* __restrict__ src, unsigned* sink, size_t n_blocks) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i < n_blocks; i += (size_t)gridDim.x * 256) {
        acc ^= *reinterpret_cast<const unsigned*>(src + i * STRIDE);
        if (SECOND) acc ^= *reinterpret_cast<const unsigned*>(src + i * STRIDE + SECOND);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const size_t bytes = 24576ull * 4096, dbytes = 3ull * 4096 * 4096;
    const int POOL = 6;
    std::vector<uint8_t*> srcs(POOL), dsts(POOL);
    for (int p = 0; p < POOL; p++) { CK(hipMalloc((void**)&srcs[p], bytes + (4 << 20))); CK(hipMemset(srcs[p], p + 1, bytes + (4 << 20))); CK(hipMalloc((void**)&dsts[p], dbytes)); }
    unsigned* sink; CK(hipMalloc((void**)&sink, 64));
    for (int i = 0; i < 12; i++) calib_stream_x4<<<4096, 256>>>((const u4*)srcs[i % POOL], sink, bytes / 16);
    for (int i = 0; i < 12; i++) calib_dma_rows96<<<4096, 256, 4 * (64 * 96 + 64)>>>(srcs[i % POOL], sink);
    for (int i = 0; i < 12; i++) calib_gather_dword<<<4096, 256>>>(srcs[i % POOL], sink);
    for (int i = 0; i < 12; i++) calib_store_12<<<4096, 256>>>(dsts[i % POOL]);
    // n_blocks * STRIDE <= bytes: the last block starts at bytes - STRIDE, its second dword at + 64 + 4 <= bytes
    for (int i = 0; i < 12; i++) calib_sparse<128, 0><<<2048, 256>>>(srcs[i % POOL], sink, bytes / 128);
    for (int i = 0; i < 12; i++) calib_sparse<128, 64><<<2048, 256>>>(srcs[i % POOL], sink, bytes / 128);
    for (int i = 0; i < 12; i++) calib_sparse<256, 0><<<2048, 256>>>(srcs[i % POOL], sink, bytes / 256);
    for (int i = 0; i < 12; i++) calib_sparse<64, 0><<<2048, 256>>>(srcs[i % POOL], sink, bytes / 64);
    CK(hipDeviceSynchronize());
    printf("known per launch: reads %zu B (all three read kernels), calib_store_12 writes %zu B\n", bytes, dbytes);
    return 0;
}
