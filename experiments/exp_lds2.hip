// Experiment 2: lean LDS-staged gather skeleton for c2.  Per-tile source windows are precomputed
// (stand-in for the per-tile model the real kernel will have), per-pixel (row, col) come packed in
// one int (stand-in for the per-pixel polynomial evaluation).  Measures the memory-side floor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../photonbend_amd/csrc/pb_params.hpp"
#include "../photonbend_amd/csrc/pb_stages.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

__global__ void k_index_rc(const PbParams P, int* out) {  // packed: r << 16 | c, or -1
    unsigned total = P.dst.height * P.dst.width;
    unsigned p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    unsigned i = p / P.dst.width, j = p - i * P.dst.width;
    PbCoord c = pb_dst_coord(P, i, j);
    int idx = pb_src_pano_index(P, c);
    out[p] = idx < 0 ? -1 : (((idx / P.src.width) << 16) | (idx % P.src.width));
}

struct Win { int r0, nrows, c0, pad; };

template <int TW, int TH>
__global__ void k_windows(const int* rc, Win* win, int W, int H) {  // one thread per tile (slow, setup only)
    const int tiles_x = W / TW;
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= tiles_x * (H / TH)) return;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    int rmin = 1 << 30, rmax = -1, cmin = 1 << 30, cmax = -1;
    for (int y = 0; y < TH; ++y) for (int x = 0; x < TW; ++x) {
        int v = rc[(ty * TH + y) * W + tx * TW + x];
        if (v >= 0) { int r = v >> 16, c = v & 0xFFFF; rmin = min(rmin, r); rmax = max(rmax, r); cmin = min(cmin, c); cmax = max(cmax, c); }
    }
    Win w; w.r0 = rmin; w.nrows = rmax < 0 ? 0 : rmax - rmin + 1; w.c0 = cmin; w.pad = cmax - cmin + 1;
    win[t] = w;
}

__device__ __forceinline__ unsigned ld4u(const uint8_t* s, unsigned long long byteoff) {
    unsigned v; __builtin_memcpy(&v, s + byteoff, 4); return v & 0xFFFFFF;
}

// PX = pixels per thread along x (4).  Block = TW x TH px, 256 threads.
template <int TW, int TH, int LDSBYTES, int ABL>
__global__ __launch_bounds__(256) void k_dyn(const int* __restrict__ rc, const Win* __restrict__ win, const uint8_t* __restrict__ src,
                                              uint8_t* __restrict__ dst, int W, int H, unsigned rowbytes, unsigned long long srcbytes) {
    __shared__ __attribute__((aligned(16))) unsigned tile32[LDSBYTES / 4 + 4];
    uint8_t* tile = (uint8_t*)tile32;
    const int tiles_x = W / TW;
    const int t = blockIdx.x;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    constexpr int TPR = TW / 4;
    constexpr int ROWS_PER_PASS = 256 / TPR;
    const int lx = (threadIdx.x % TPR) * 4, ly = threadIdx.x / TPR;
    const Win w = win[t];
    // window: rows [r0, r0+nrows), bytes [3*c0 - a0, ...) with a0 = alignment slack; pitch multiple of 16
    const unsigned a0 = (3u * w.c0) & 15u;                 // rowbytes % 16 == 0 in this experiment
    const unsigned pitch = (3u * w.pad + a0 + 1 + 15u) & ~15u;  // +1: 4-byte reads of the last texel
    const unsigned maxrows = LDSBYTES / pitch;
    const unsigned nrows = min((unsigned)w.nrows, maxrows);
    if (ABL != 1 && w.nrows) {
        const unsigned lpr = pitch >> 4;
        const unsigned rpp = 256 / lpr;
        const unsigned rr = threadIdx.x / lpr, sub = threadIdx.x - rr * lpr;
        if (rr < rpp)
            for (unsigned row = rr; row < nrows; row += rpp) {
                const unsigned long long ga = (unsigned long long)(w.r0 + row) * rowbytes + 3u * w.c0 - a0 + 16u * sub;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ga + 16 <= srcbytes) v = *(const uint4*)(src + ga);
                *(uint4*)(tile + row * pitch + 16 * sub) = v;
            }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < TH / ROWS_PER_PASS; ++pass) {
        const int y = ty * TH + ly + pass * ROWS_PER_PASS;
        const unsigned p0 = (unsigned)y * W + tx * TW + lx;
        const int4 id4 = *(const int4*)(rc + p0);
        const int id[4] = {id4.x, id4.y, id4.z, id4.w};
        unsigned a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned v = 0;
            if (id[k] >= 0) {
                const unsigned r = (unsigned)id[k] >> 16, c = id[k] & 0xFFFF;
                const unsigned row = r - w.r0, dc = c - w.c0;
                if (ABL == 2) v = row + dc;
                else if (row < nrows) {
                    const unsigned la = __umul24(row, pitch) + __umul24(dc, 3u) + a0;
                    const unsigned lo = tile32[la >> 2], hi = tile32[(la >> 2) + 1];
                    v = __builtin_amdgcn_alignbyte(hi, lo, la & 3) & 0xFFFFFF;
                } else v = ld4u(src, (unsigned long long)r * rowbytes + 3u * c);
            }
            a[k] = v;
        }
        u32x3 o;
        o.x = a[0] | (a[1] << 24); o.y = (a[1] >> 8) | (a[2] << 16); o.z = (a[2] >> 16) | (a[3] << 8);
        *(u32x3*)(dst + 3ull * p0) = o;
    }
}

template <int TW, int TH, int MAXROWS, int PITCH, int ABL>
__global__ __launch_bounds__(256) void k_lean(const int* __restrict__ rc, const Win* __restrict__ win, const uint8_t* __restrict__ src,
                                              uint8_t* __restrict__ dst, int W, int H, unsigned rowbytes, unsigned long long srcbytes) {
    __shared__ __attribute__((aligned(16))) unsigned tile32[MAXROWS * PITCH / 4 + 4];
    uint8_t* tile = (uint8_t*)tile32;
    const int tiles_x = W / TW;
    const int t = blockIdx.x;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    constexpr int TPR = TW / 4;             // threads per tile row
    constexpr int ROWS_PER_PASS = 256 / TPR;
    const int lx = (threadIdx.x % TPR) * 4, ly = threadIdx.x / TPR;
    const Win w = win[t];
    const int nrows = min(w.nrows, MAXROWS);
    if (ABL != 1) {
        constexpr int LPR = PITCH / 16;
        const int sub = threadIdx.x % LPR, rr = threadIdx.x / LPR;
        for (int row = rr; row < nrows; row += 256 / LPR) {
            const unsigned long long g = (unsigned long long)(w.r0 + row) * rowbytes + 3u * w.c0;
            const unsigned long long ga = (g & ~15ull) + 16u * sub;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ga + 16 <= srcbytes) v = *(const uint4*)(src + ga);
            *(uint4*)(tile + row * PITCH + 16 * sub) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < TH / ROWS_PER_PASS; ++pass) {
        const int y = ty * TH + ly + pass * ROWS_PER_PASS;
        const unsigned p0 = (unsigned)y * W + tx * TW + lx;
        const int4 id4 = *(const int4*)(rc + p0);
        const int id[4] = {id4.x, id4.y, id4.z, id4.w};
        unsigned a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned v = 0;
            if (id[k] >= 0) {
                const int r = id[k] >> 16, c = id[k] & 0xFFFF;
                const int row = r - w.r0;
                const unsigned rb = (unsigned)r * rowbytes;     // < 2^32 for these sizes
                const unsigned gbase = (rb + 3u * w.c0) & ~15u;
                const unsigned off = rb + 3u * c - gbase;
                if (ABL == 2) v = off + row;
                else if (row < MAXROWS && off + 4 <= PITCH) {
                    const unsigned la = row * PITCH + off;
                    const unsigned lo = tile32[la >> 2], hi = tile32[(la >> 2) + 1];
                    v = __builtin_amdgcn_alignbyte(hi, lo, la & 3) & 0xFFFFFF;
                } else v = ld4u(src, (unsigned long long)rb + 3u * c);
            }
            a[k] = v;
        }
        u32x3 o;
        o.x = a[0] | (a[1] << 24); o.y = (a[1] >> 8) | (a[2] << 16); o.z = (a[2] >> 16) | (a[3] << 8);
        *(u32x3*)(dst + 3ull * p0) = o;
    }
}

__global__ void k_ref(const int* __restrict__ rc, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, unsigned total, unsigned rowbytes) {
    unsigned p = blockIdx.x * 256 + threadIdx.x; if (p >= total) return;
    int v = rc[p]; unsigned a = 0;
    if (v >= 0) a = ld4u(src, (unsigned long long)(v >> 16) * rowbytes + 3u * (v & 0xFFFF));
    dst[3ull * p] = a; dst[3ull * p + 1] = a >> 8; dst[3ull * p + 2] = a >> 16;
}
__global__ void k_cmp(const uint8_t* a, const uint8_t* b, size_t n, unsigned* bad) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n && a[i] != b[i]) atomicAdd(bad, 1u);
}

int main() {
    const int DH = 4096, DW_ = 4096, SH = 4096, SW = 8192;
    PbParams P; memset(&P, 0, sizeof(P));
    P.dst = {PB_KIND_CAMERA, PB_LENS_EQUIDISTANT, DH, DW_, 2 * PB_PI, 2047.5 / PB_PI};
    P.src = {PB_KIND_PANO, 0, SH, SW, 0, 0};
    pb_derive(P);
    const unsigned total = DH * DW_;
    int* rc; CK(hipMalloc(&rc, 4ull * total));
    const int POOL = 4;
    uint8_t *src[POOL], *dst[POOL];
    std::vector<uint8_t> h(3ull * SH * SW);
    unsigned x = 12345; for (auto& b : h) { x = x * 1664525u + 1013904223u; b = x >> 24; }
    for (int i = 0; i < POOL; ++i) { CK(hipMalloc(&src[i], 3ull * SH * SW + 16)); CK(hipMalloc(&dst[i], 3ull * total + 16)); CK(hipMemcpy(src[i], h.data(), h.size(), hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_index_rc, dim3(total / 256), dim3(256), 0, 0, P, rc);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch(i % POOL);
        CK(hipDeviceSynchronize());
        const int N = 40;
        CK(hipEventRecord(e0));
        for (int i = 0; i < N; ++i) launch(i % POOL);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8.2f us/frame   alg %.0f GB/s\n", name, ms * 1000 / N, 89842104.0 / (ms / N * 1e-3) / 1e9);
    };
    unsigned* bad; CK(hipMalloc(&bad, 4));
    uint8_t* refout; CK(hipMalloc(&refout, 3ull * total + 16));
    hipLaunchKernelGGL(k_ref, dim3(total / 256), dim3(256), 0, 0, rc, src[0], refout, total, 3u * SW);
    auto run = [&](auto tw_, auto th_, auto mr_, auto pitch_, const char* name) {
        constexpr int TW = decltype(tw_)::value, TH = decltype(th_)::value, MR = decltype(mr_)::value, PT = decltype(pitch_)::value;
        const int ntiles = (DW_ / TW) * (DH / TH);
        Win* win; CK(hipMalloc(&win, sizeof(Win) * ntiles));
        hipLaunchKernelGGL((k_windows<TW, TH>), dim3((ntiles + 63) / 64), dim3(64), 0, 0, rc, win, DW_, DH);
        CK(hipDeviceSynchronize());
        std::vector<Win> hw(ntiles); CK(hipMemcpy(hw.data(), win, sizeof(Win) * ntiles, hipMemcpyDeviceToHost));
        double rows = 0, cols = 0; int over_r = 0, over_c = 0, nz = 0;
        for (auto& w : hw) if (w.nrows) { rows += w.nrows; cols += w.pad; nz++; over_r += w.nrows > MR; over_c += 3 * w.pad + 19 > PT; }
        printf("[%s] tiles %d (nonempty %d) mean rows %.1f mean cols %.1f; rows>MAX %d, cols>pitch %d\n", name, ntiles, nz, rows / nz, cols / nz, over_r, over_c);
        char nm[128];
        snprintf(nm, 128, "%s full", name);
        timeit(nm, [&](int f) { hipLaunchKernelGGL((k_lean<TW, TH, MR, PT, 0>), dim3(ntiles), dim3(256), 0, 0, rc, win, src[f], dst[f], DW_, DH, 3u * SW, 3ull * SH * SW); });
        snprintf(nm, 128, "%s no-stage-loads", name);
        timeit(nm, [&](int f) { hipLaunchKernelGGL((k_lean<TW, TH, MR, PT, 1>), dim3(ntiles), dim3(256), 0, 0, rc, win, src[f], dst[f], DW_, DH, 3u * SW, 3ull * SH * SW); });
        snprintf(nm, 128, "%s no-lds-reads", name);
        timeit(nm, [&](int f) { hipLaunchKernelGGL((k_lean<TW, TH, MR, PT, 2>), dim3(ntiles), dim3(256), 0, 0, rc, win, src[f], dst[f], DW_, DH, 3u * SW, 3ull * SH * SW); });
        CK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL((k_lean<TW, TH, MR, PT, 0>), dim3(ntiles), dim3(256), 0, 0, rc, win, src[1], dst[1], DW_, DH, 3u * SW, 3ull * SH * SW);
        hipLaunchKernelGGL(k_cmp, dim3((3ull * total + 255) / 256), dim3(256), 0, 0, dst[0], dst[1], 3ull * total, bad);
        unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); printf("   mismatching bytes vs reference gather: %u\n", hb);
        CK(hipFree(win));
    };
    auto rund = [&](auto tw_, auto th_, auto lds_, const char* name) {
        constexpr int TW = decltype(tw_)::value, TH = decltype(th_)::value, LB = decltype(lds_)::value;
        const int ntiles = (DW_ / TW) * (DH / TH);
        Win* win; CK(hipMalloc(&win, sizeof(Win) * ntiles));
        hipLaunchKernelGGL((k_windows<TW, TH>), dim3((ntiles + 63) / 64), dim3(64), 0, 0, rc, win, DW_, DH);
        CK(hipDeviceSynchronize());
        std::vector<Win> hw(ntiles); CK(hipMemcpy(hw.data(), win, sizeof(Win) * ntiles, hipMemcpyDeviceToHost));
        double bytes = 0; int clamp = 0;
        for (auto& w : hw) if (w.nrows) { unsigned pitch = (3u * w.pad + 31) & ~15u; unsigned mr = LB / pitch; bytes += (double)pitch * std::min<unsigned>(w.nrows, mr); clamp += (unsigned)w.nrows > mr; }
        printf("[%s] staged bytes/frame %.1f MB, tiles with clamped rows %d\n", name, bytes / 1e6, clamp);
        char nm[128];
        snprintf(nm, 128, "%s full", name);
        timeit(nm, [&](int f) { hipLaunchKernelGGL((k_dyn<TW, TH, LB, 0>), dim3(ntiles), dim3(256), 0, 0, rc, win, src[f], dst[f], DW_, DH, 3u * SW, 3ull * SH * SW); });
        snprintf(nm, 128, "%s no-stage-loads", name);
        timeit(nm, [&](int f) { hipLaunchKernelGGL((k_dyn<TW, TH, LB, 1>), dim3(ntiles), dim3(256), 0, 0, rc, win, src[f], dst[f], DW_, DH, 3u * SW, 3ull * SH * SW); });
        snprintf(nm, 128, "%s no-lds-reads", name);
        timeit(nm, [&](int f) { hipLaunchKernelGGL((k_dyn<TW, TH, LB, 2>), dim3(ntiles), dim3(256), 0, 0, rc, win, src[f], dst[f], DW_, DH, 3u * SW, 3ull * SH * SW); });
        CK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL((k_dyn<TW, TH, LB, 0>), dim3(ntiles), dim3(256), 0, 0, rc, win, src[1], dst[1], DW_, DH, 3u * SW, 3ull * SH * SW);
        hipLaunchKernelGGL(k_cmp, dim3((3ull * total + 255) / 256), dim3(256), 0, 0, refout, dst[1], 3ull * total, bad);
        unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); printf("   mismatching bytes vs reference gather: %u\n", hb);
        CK(hipFree(win));
    };
    using std::integral_constant;
    rund(integral_constant<int, 32>{}, integral_constant<int, 32>{}, integral_constant<int, 32768>{}, "dyn 32x32 32K");
    rund(integral_constant<int, 64>{}, integral_constant<int, 16>{}, integral_constant<int, 32768>{}, "dyn 64x16 32K");
    rund(integral_constant<int, 64>{}, integral_constant<int, 32>{}, integral_constant<int, 49152>{}, "dyn 64x32 48K");
    rund(integral_constant<int, 16>{}, integral_constant<int, 64>{}, integral_constant<int, 32768>{}, "dyn 16x64 32K");
    return 0;
    run(integral_constant<int, 32>{}, integral_constant<int, 32>{}, integral_constant<int, 96>{}, integral_constant<int, 128>{}, "32x32 r96 p128");
    run(integral_constant<int, 32>{}, integral_constant<int, 32>{}, integral_constant<int, 96>{}, integral_constant<int, 192>{}, "32x32 r96 p192");
    run(integral_constant<int, 64>{}, integral_constant<int, 64>{}, integral_constant<int, 192>{}, integral_constant<int, 256>{}, "64x64 r192 p256");
    run(integral_constant<int, 64>{}, integral_constant<int, 32>{}, integral_constant<int, 144>{}, integral_constant<int, 192>{}, "64x32 r144 p192");
    run(integral_constant<int, 64>{}, integral_constant<int, 16>{}, integral_constant<int, 128>{}, integral_constant<int, 192>{}, "64x16 r128 p192");
    return 0;
}
