// exp_req.hip - what bounds a CU's window fetches: bytes, or L1->L2 REQUESTS (64 B vs 128 B)?
// Every wave pulls a window of R rows x S bytes (rows `rowbytes` apart, like a panorama) into LDS by LDS-DMA
// (16 B per lane), windows partition a 100 MB source exactly once; optional 3 KiB nt store per tile.
// Sweep S and the alignment of the row segments against 64 / 128-byte lines at equal bytes per tile.
// build: hipcc --offload-arch=gfx950 -O3 -o exp_req exp_req.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
struct Args { const uint8_t* src; uint8_t* dst; unsigned* sink; int R, n16, wins_per_row, n_tiles, store, mis, regs; unsigned rowbytes; };
extern __shared__ __attribute__((aligned(16))) unsigned lds[];

__global__ __launch_bounds__(256) void k_win(const Args A, int lds_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wave;
    if (w >= A.n_tiles) return;
    unsigned* win = lds + wave * (lds_per_wave >> 2);
    const int n16 = A.n16;
    const unsigned inv = (65536u + n16 - 1) / n16;
    const unsigned lrow = ((unsigned)lane * inv) >> 16, chunk = (unsigned)lane - lrow * n16;
    const unsigned rpp = 64u / n16;
    const bool lane_on = lrow < rpp;
    const unsigned wr = (unsigned)w / A.wins_per_row, wc = (unsigned)w % A.wins_per_row;
    const unsigned gbase = wr * A.R * A.rowbytes + wc * n16 * 16u + A.mis;
    const unsigned pitch = 16u * n16;
    unsigned acc = 0;
    if (A.regs) {
        for (unsigned rowb = 0; rowb < (unsigned)A.R; rowb += rpp) {
            const unsigned row = rowb + lrow;
            if (lane_on && row < (unsigned)A.R) {
                const uint4 v = *reinterpret_cast<const uint4*>(A.src + gbase + row * A.rowbytes + 16u * chunk);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    } else {
        for (unsigned rowb = 0; rowb < (unsigned)A.R; rowb += rpp) {
            const unsigned row = rowb + lrow;
            if (lane_on && row < (unsigned)A.R)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.src + gbase + row * A.rowbytes + 16u * chunk),
                                                 (__attribute__((address_space(3))) void*)(win + ((rowb * pitch) >> 2)), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= win[(lane * 37 + w) % (lds_per_wave / 4)];
    }
    if (A.store) {
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            u32x3 o = {acc + jr, acc ^ lane, acc};
            const int tx = w & 127, ty = w >> 7;
            const size_t off = 3ull * ((size_t)(ty * 32 + (lane >> 3) + 8 * jr) * 4096 + tx * 32 + 4 * (lane & 7));
            __builtin_nontemporal_store(o, reinterpret_cast<u32x3*>(A.dst + off));
        }
    } else if (acc == 0x12345678u) A.sink[0] = acc;
}

int main() {
    const unsigned rowbytes = 24576, H = 4096;
    const size_t bytes = (size_t)rowbytes * H;
    const size_t dbytes = 3ull * 4096 * 4096;
    const int POOL = 6;
    std::vector<uint8_t*> srcs(POOL), dsts(POOL);
    for (int p = 0; p < POOL; p++) {
        CK(hipMalloc((void**)&srcs[p], bytes + 256 * rowbytes)); CK(hipMemset(srcs[p], p + 1, bytes + 256 * rowbytes));
        CK(hipMalloc((void**)&dsts[p], dbytes));
    }
    unsigned* sink; CK(hipMalloc((void**)&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // {R, S bytes}: ~6 KiB per tile in different aspect ratios, then c2-like 64 x 96
    const int shapes[][2] = {{96, 64}, {64, 96}, {48, 128}, {32, 192}, {24, 256}, {12, 512}, {6, 1024}, {64, 128}, {64, 64}, {128, 64}, {32, 128}};
    for (int regs : {0, 1})
    for (int store : {0, 1})
        for (auto& sh : shapes)
            for (int mis : {0, 16, 64}) {
                Args A;
                A.R = sh[0]; A.n16 = sh[1] / 16; A.rowbytes = rowbytes; A.wins_per_row = rowbytes / sh[1];
                A.n_tiles = (H / A.R) * A.wins_per_row; A.store = store; A.mis = mis; A.sink = sink; A.regs = regs;
                if (store && A.n_tiles > 16384) A.n_tiles = 16384;
                const int lds_per_wave = A.R * sh[1] + 64;
                if (regs && (mis == 16 || store)) continue;
                auto launch = [&](int p) {
                    A.src = srcs[p]; A.dst = dsts[p];
                    k_win<<<(A.n_tiles + 3) / 4, 256, regs ? 64 : 4 * lds_per_wave>>>(A, lds_per_wave);
                };
                for (int i = 0; i < 5; i++) launch(i % POOL);
                CK(hipDeviceSynchronize());
                const int N = 30;
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < N; i++) launch(i % POOL);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / N, rd = (double)A.n_tiles * A.R * sh[1];
                printf("%s window %3d rows x %4d B  misalign %2d  store %d: %6d tiles (%5.1f MB read) %7.2f us  read %.2f TB/s  %.2f ns/row-segment/CU\n",
                       regs ? "regs" : "dma ", sh[0], sh[1], mis, store, A.n_tiles, rd / 1e6, us, rd / us / 1e6, us * 1e3 * 256 / ((double)A.n_tiles * A.R));
            }
    return 0;
}
