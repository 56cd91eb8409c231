// Practical HBM ceiling for the c2 byte mix: stream-read R bytes and stream-write W bytes per launch
// (c2: R = 100 MB source, W = 50 MB destination), plain 16-byte loads vs LDS-DMA loads.
// build: hipcc --offload-arch=gfx950 -O3 -o exp_stream exp_stream.hip ; run: ./exp_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// each block: reads rd_per_blk 16-byte vectors, writes wr_per_blk 16-byte vectors
__global__ __launch_bounds__(256) void k_plain(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n_rd, size_t n_wr) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = t; i < n_rd; i += stride) acc ^= __builtin_nontemporal_load(src + i);
    for (size_t i = t; i < n_wr; i += stride) {
        u32x4 v = acc; v.x += (uint32_t)i;
        __builtin_nontemporal_store(v, dst + i);
    }
}

// tile-shaped: each block owns one contiguous chunk of reads (as 2 reads per 1 write) and its chunk of writes
__global__ __launch_bounds__(256) void k_chunk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int rd_per_thr, int wr_per_thr) {
    const u32x4* s = src + (size_t)blockIdx.x * 256 * rd_per_thr;
    u32x4* d = dst + (size_t)blockIdx.x * 256 * wr_per_thr;
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < rd_per_thr; i++) acc ^= s[i * 256 + threadIdx.x];
    for (int i = 0; i < wr_per_thr; i++) { u32x4 v = acc; v.x += i; __builtin_nontemporal_store(v, d + i * 256 + threadIdx.x); }
}

__global__ __launch_bounds__(256) void k_dma(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int rd_per_thr, int wr_per_thr) {
    __shared__ u32x4 lds[256 * 8];
    const u32x4* s = src + (size_t)blockIdx.x * 256 * rd_per_thr;
    u32x4* d = dst + (size_t)blockIdx.x * 256 * wr_per_thr;
    u32x4 acc = {0, 0, 0, 0};
    for (int i0 = 0; i0 < rd_per_thr; i0 += 8) {
        for (int i = 0; i < 8 && i0 + i < rd_per_thr; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + (i0 + i) * 256 + threadIdx.x),
                                             (__attribute__((address_space(3))) void*)(lds + i * 256 + (threadIdx.x & ~63)), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        for (int i = 0; i < 8 && i0 + i < rd_per_thr; i++) acc ^= lds[i * 256 + threadIdx.x];
        __syncthreads();
    }
    for (int i = 0; i < wr_per_thr; i++) { u32x4 v = acc; v.x += i; __builtin_nontemporal_store(v, d + i * 256 + threadIdx.x); }
}

int main() {
    const size_t R = 100663296, W = 50331648;  // c2 bytes
    const int POOL = 6;
    std::vector<void*> srcs(POOL), dsts(POOL);
    for (int p = 0; p < POOL; p++) { CK(hipMalloc(&srcs[p], R)); CK(hipMalloc(&dsts[p], W)); CK(hipMemset(srcs[p], p + 1, R)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; i++) launch(i % POOL);
        hipDeviceSynchronize();
        const int N = 60;
        hipEventRecord(e0, 0);
        for (int i = 0; i < N; i++) launch(i % POOL);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double us = ms * 1e3 / N;
        printf("%-34s %8.2f us  %7.2f TB/s (R+W = %.0f MB)\n", name, us, (R + W) / us * 1e-6, (R + W) * 1e-6);
    };
    for (int blocks : {1024, 2048, 4096, 8192}) {
        char nm[64]; snprintf(nm, 64, "plain grid-stride, %d blocks", blocks);
        run(nm, [&](int p) { k_plain<<<blocks, 256>>>((const u32x4*)srcs[p], (u32x4*)dsts[p], R / 16, W / 16); });
    }
    // chunked: 4096 blocks (= c2's block count: 16384 tiles / 4), 6 reads + 3 writes of 16 B per thread
    run("chunk 4096 blocks (6 rd + 3 wr)", [&](int p) { k_chunk<<<4096, 256>>>((const u32x4*)srcs[p], (u32x4*)dsts[p], 6, 3); });
    run("chunk 2048 blocks (12 rd + 6 wr)", [&](int p) { k_chunk<<<2048, 256>>>((const u32x4*)srcs[p], (u32x4*)dsts[p], 12, 6); });
    run("chunk 8192 blocks (3 rd + 1 wr)", [&](int p) { k_chunk<<<8192, 256>>>((const u32x4*)srcs[p], (u32x4*)dsts[p], 3, 1); });
    run("lds-dma 4096 blocks (6 rd + 3 wr)", [&](int p) { k_dma<<<4096, 256>>>((const u32x4*)srcs[p], (u32x4*)dsts[p], 6, 3); });
    run("lds-dma 2048 blocks (12 rd + 6 wr)", [&](int p) { k_dma<<<2048, 256>>>((const u32x4*)srcs[p], (u32x4*)dsts[p], 12, 6); });
    return 0;
}
