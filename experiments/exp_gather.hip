// Experiment: memory-side ceiling of the c2 gather, index math removed (precomputed
// int32 index map).  Variants differ in pixel->lane mapping and load width.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "../photonbend_amd/csrc/pb_params.hpp"
#include "../photonbend_amd/csrc/pb_stages.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_index(const PbParams P, int* out) {
    unsigned total = P.dst.height * P.dst.width;
    unsigned p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    unsigned i = p / P.dst.width, j = p - i * P.dst.width;
    PbCoord c = pb_dst_coord(P, i, j);
    out[p] = pb_src_pano_index(P, c);
}

__device__ __forceinline__ unsigned ld3(const uint8_t* s, int idx) {
    if (idx < 0) return 0;
    const uint8_t* p = s + 3ull * (unsigned)idx;
    return p[0] | (p[1] << 8) | (p[2] << 16);
}
__device__ __forceinline__ unsigned ld4u(const uint8_t* s, int idx) {
    if (idx < 0) return 0;
    unsigned v;
    __builtin_memcpy(&v, s + 3ull * (unsigned)idx, 4);
    return v & 0xFFFFFF;
}
__device__ __forceinline__ void st12(uint8_t* o, unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    uint32_t* o32 = (uint32_t*)o;
    o32[0] = a0 | (a1 << 24);
    o32[1] = (a1 >> 8) | (a2 << 16);
    o32[2] = (a2 >> 16) | (a3 << 8);
}

// V0: row-major, thread = 4 consecutive px, byte loads (the round-1 baseline shape)
template <int DW>
__global__ __launch_bounds__(256) void k_rowmajor(const int* __restrict__ idx, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, unsigned total) {
    unsigned g = blockIdx.x * 256 + threadIdx.x;
    unsigned p0 = g * 4;
    if (p0 >= total) return;
    int4 id = *(const int4*)(idx + p0);
    unsigned a0, a1, a2, a3;
    if (DW) { a0 = ld4u(src, id.x); a1 = ld4u(src, id.y); a2 = ld4u(src, id.z); a3 = ld4u(src, id.w); }
    else { a0 = ld3(src, id.x); a1 = ld3(src, id.y); a2 = ld3(src, id.z); a3 = ld3(src, id.w); }
    st12(dst + 3ull * p0, a0, a1, a2, a3);
}

// V1: 2-D tiles.  A block of 256 threads covers TW x TH pixels (TW*TH = 1024), thread = 4 px along x.
// Tiles are ordered so that consecutive blocks on one XCD are spatial neighbours when XCD=1.
template <int TW, int TH, int XCD>
__global__ __launch_bounds__(256) void k_tile(const int* __restrict__ idx, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H) {
    const int tiles_x = W / TW, tiles_y = H / TH;
    unsigned b = blockIdx.x;
    if (XCD) {
        const unsigned nb = tiles_x * tiles_y;
        const unsigned per = nb / 8;  // assumes nb % 8 == 0
        b = (b % 8) * per + b / 8;    // blocks with equal b%8 (one XCD) get a contiguous range of tiles
    }
    const int ty = b / tiles_x, tx = b - ty * tiles_x;
    const int lx = (threadIdx.x % (TW / 4)) * 4, ly = threadIdx.x / (TW / 4);
    const int x = tx * TW + lx, y = ty * TH + ly;
    const unsigned p0 = (unsigned)y * W + x;
    int4 id = *(const int4*)(idx + p0);
    unsigned a0 = ld4u(src, id.x), a1 = ld4u(src, id.y), a2 = ld4u(src, id.z), a3 = ld4u(src, id.w);
    st12(dst + 3ull * p0, a0, a1, a2, a3);
}


// V2: per load instruction the 64 lanes cover a compact 8x8 px block (timing only: the
// store layout is the thread's own 12-byte slot, not the right place).
template <int MODE>
__global__ __launch_bounds__(256) void k_compact(const int* __restrict__ idx, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H) {
    // block = 32x32 px; wave w covers rows 8w..8w+7, 32 px wide = four 8x8 sub-blocks
    const int tiles_x = W / 32;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = ty * 32 + wave * 8 + (lane >> 3);
    const int xb = tx * 32 + (lane & 7);
    unsigned a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int id = (MODE == 2) ? -1 : idx[(unsigned)y * W + xb + 8 * k];
        a[k] = (MODE == 1) ? (unsigned)id : ld4u(src, id);
    }
    const unsigned p0 = ((unsigned)(ty * 32 + wave * 8 + (lane >> 3)) * W + tx * 32 + (lane & 7) * 4);
    st12(dst + 3ull * p0, a[0], a[1], a[2], a[3]);
}


// V3: LDS-staged gather.  Block = TWxTH px (thread = 4 px along x).  The block's source bounding
// box (rows x 16B-aligned row segments) is loaded with coalesced dwordx4 loads into LDS, pixels
// inside the window read LDS, the rest (wrap / pole tiles) read global memory.
// XCD-aware order: blocks with equal b%8 share an XCD (observed round-robin dispatch); give each XCD
// whole super-tiles (ST x ST tiles), super-tiles dealt round-robin so the XCDs stay balanced.
template <int ST>
__device__ __forceinline__ void xcd_tile(unsigned b, int tiles_x, int tiles_y, int& tx, int& ty) {
    const unsigned xcd = b & 7, slot = b >> 3;
    const unsigned s_local = slot / (ST * ST), inner = slot % (ST * ST);
    const unsigned S = s_local * 8 + xcd;           // global super-tile id
    const unsigned sx_n = tiles_x / ST;
    const unsigned sy = S / sx_n, sx = S - sy * sx_n;
    tx = sx * ST + inner % ST;
    ty = sy * ST + inner / ST;
}
template <int TW, int TH, int MAXROWS, int PITCH, int ST = 0, int ABL = 0>
__global__ __launch_bounds__(256) void k_lds(const int* __restrict__ idx, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H, int SW, int SH) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[MAXROWS * PITCH];
    __shared__ int bb[4];
    const int tiles_x = W / TW;
    int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    if (ST) xcd_tile<ST ? ST : 1>(blockIdx.x, tiles_x, H / TH, tx, ty);
    const int lx = (threadIdx.x % (TW / 4)) * 4, ly = threadIdx.x / (TW / 4);
    const unsigned p0 = (unsigned)(ty * TH + ly) * W + tx * TW + lx;
    if (threadIdx.x == 0) { bb[0] = 0x7fffffff; bb[1] = -1; bb[2] = 0x7fffffff; bb[3] = -1; }
    int4 id4 = *(const int4*)(idx + p0);
    int id[4] = {id4.x, id4.y, id4.z, id4.w};
    int r[4], c[4];
    int rmin = 0x7fffffff, rmax = -1, cmin = 0x7fffffff, cmax = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] = id[k] / SW; c[k] = id[k] - r[k] * SW;
        if (id[k] >= 0) { rmin = min(rmin, r[k]); rmax = max(rmax, r[k]); cmin = min(cmin, c[k]); cmax = max(cmax, c[k]); }
    }
    // wave reduce then one LDS atomic per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rmin = min(rmin, __shfl_xor(rmin, o)); rmax = max(rmax, __shfl_xor(rmax, o));
        cmin = min(cmin, __shfl_xor(cmin, o)); cmax = max(cmax, __shfl_xor(cmax, o));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicMin(&bb[0], rmin); atomicMax(&bb[1], rmax); atomicMin(&bb[2], cmin); atomicMax(&bb[3], cmax); }
    __syncthreads();
    const int r0 = bb[0], r1 = bb[1], c0 = bb[2];
    const int nrows = min(r1 - r0 + 1, MAXROWS);
    const unsigned rowbytes = 3u * SW;
    // cooperative load: each row segment = PITCH bytes starting at the 16B-aligned address <= first byte
    if (r1 >= 0 && ABL != 1) {
        const int lanes_per_row = PITCH / 16;
        const int rows_per_pass = 256 / lanes_per_row;
        const int sub = threadIdx.x % lanes_per_row, rr = threadIdx.x / lanes_per_row;
        for (int row = rr; row < nrows; row += rows_per_pass) {
            const unsigned long long g = (unsigned long long)(r0 + row) * rowbytes + 3u * c0;
            const unsigned long long ga = (g & ~15ull) + 16u * sub;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ga + 16 <= (unsigned long long)SH * rowbytes) v = *(const uint4*)(src + ga);
            *(uint4*)(tile + row * PITCH + 16 * sub) = v;
        }
    }
    __syncthreads();
    unsigned a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned v = 0;
        if (id[k] >= 0) {
            const int row = r[k] - r0;
            const unsigned long long g = (unsigned long long)r[k] * rowbytes + 3u * c[k];
            const unsigned long long gbase = ((unsigned long long)r[k] * rowbytes + 3u * c0) & ~15ull;
            const unsigned off = (unsigned)(g - gbase);
            if (ABL == 2) { v = off; } else if (row < MAXROWS && off + 4 <= PITCH) {
                const uint8_t* q = tile + row * PITCH + off;
                v = q[0] | (q[1] << 8) | (q[2] << 16);
            } else {
                v = ld4u(src, id[k]);
            }
        }
        a[k] = v;
    }
    st12(dst + 3ull * p0, a[0], a[1], a[2], a[3]);
}
__global__ void k_cmp(const uint8_t* a, const uint8_t* b, size_t n, unsigned* bad) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(bad, 1u);
}

// streaming copy reference: read idx (4B/px) + write 3B/px, and plain memcpy-like
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

int main(int argc, char** argv) {
    const int DH = 4096, DW_ = 4096, SH = 4096, SW = 8192;
    PbParams P; memset(&P, 0, sizeof(P));
    P.dst = {PB_KIND_CAMERA, PB_LENS_EQUIDISTANT, DH, DW_, 2 * PB_PI, 2047.5 / PB_PI};
    P.src = {PB_KIND_PANO, 0, SH, SW, 0, 0};
    pb_derive(P);
    const unsigned total = DH * DW_;
    int* idx; CK(hipMalloc(&idx, 4ull * total));
    const int POOL = 4;
    uint8_t *src[POOL], *dst[POOL];
    for (int i = 0; i < POOL; ++i) { CK(hipMalloc(&src[i], 3ull * SH * SW + 16)); CK(hipMalloc(&dst[i], 3ull * total)); CK(hipMemset(src[i], 17 * i + 1, 3ull * SH * SW)); }
    hipLaunchKernelGGL(k_index, dim3(total / 256), dim3(256), 0, 0, P, idx);
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
        printf("%-28s %8.2f us/frame   alg %.0f GB/s\n", name, ms * 1000 / N, 89842104.0 / (ms / N * 1e-3) / 1e9);
    };
    timeit("rowmajor bytes", [&](int f) { hipLaunchKernelGGL(k_rowmajor<0>, dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], total); });
    timeit("rowmajor dword", [&](int f) { hipLaunchKernelGGL(k_rowmajor<1>, dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], total); });
    timeit("tile 64x16", [&](int f) { hipLaunchKernelGGL((k_tile<64, 16, 0>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH); });
    timeit("tile 64x16 xcd", [&](int f) { hipLaunchKernelGGL((k_tile<64, 16, 1>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH); });
    timeit("tile 32x32", [&](int f) { hipLaunchKernelGGL((k_tile<32, 32, 0>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH); });
    timeit("tile 32x32 xcd", [&](int f) { hipLaunchKernelGGL((k_tile<32, 32, 1>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH); });
    timeit("tile 128x8", [&](int f) { hipLaunchKernelGGL((k_tile<128, 8, 0>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH); });
    timeit("tile 128x8 xcd", [&](int f) { hipLaunchKernelGGL((k_tile<128, 8, 1>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH); });
    timeit("tile 16x64 xcd", [&](int f) { hipLaunchKernelGGL((k_tile<16, 64, 1>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH); });
    timeit("compact 8x8 gather", [&](int f) { hipLaunchKernelGGL((k_compact<0>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH); });
    timeit("compact idx+store only", [&](int f) { hipLaunchKernelGGL((k_compact<1>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH); });
    timeit("compact store only", [&](int f) { hipLaunchKernelGGL((k_compact<2>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH); });
    // real (non-constant) source so that the comparison means something
    {
        std::vector<uint8_t> h(3ull * SH * SW);
        unsigned x = 12345; for (auto& b : h) { x = x * 1664525u + 1013904223u; b = x >> 24; }
        for (int i = 0; i < POOL; ++i) CK(hipMemcpy(src[i], h.data(), h.size(), hipMemcpyHostToDevice));
    }
    timeit("lds 32x32 r128 p256", [&](int f) { hipLaunchKernelGGL((k_lds<32, 32, 128, 256>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH, SW, SH); });
    timeit("lds 32x32 r128 p256 st8", [&](int f) { hipLaunchKernelGGL((k_lds<32, 32, 128, 256, 8>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH, SW, SH); });
    timeit("lds 32x32 r96 p128 st8", [&](int f) { hipLaunchKernelGGL((k_lds<32, 32, 96, 128, 8>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH, SW, SH); });
    timeit("lds 32x32 r96 p128 st4", [&](int f) { hipLaunchKernelGGL((k_lds<32, 32, 96, 128, 4>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH, SW, SH); });
    timeit("lds 32x32 r96 p128 st16", [&](int f) { hipLaunchKernelGGL((k_lds<32, 32, 96, 128, 16>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH, SW, SH); });
    timeit("lds r96 p128 st8 NO-STAGE-LOADS", [&](int f) { hipLaunchKernelGGL((k_lds<32, 32, 96, 128, 8, 1>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH, SW, SH); });
    timeit("lds r96 p128 st8 NO-LDS-READS", [&](int f) { hipLaunchKernelGGL((k_lds<32, 32, 96, 128, 8, 2>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH, SW, SH); });
    timeit("lds 64x16 r128 p256", [&](int f) { hipLaunchKernelGGL((k_lds<64, 16, 128, 256>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH, SW, SH); });
    timeit("lds 32x32 r96 p128", [&](int f) { hipLaunchKernelGGL((k_lds<32, 32, 96, 128>), dim3(total / 1024), dim3(256), 0, 0, idx, src[f], dst[f], DW_, DH, SW, SH); });
    {
        unsigned* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(k_rowmajor<0>, dim3(total / 1024), dim3(256), 0, 0, idx, src[0], dst[0], total);
        hipLaunchKernelGGL((k_lds<32, 32, 128, 256>), dim3(total / 1024), dim3(256), 0, 0, idx, src[1], dst[1], DW_, DH, SW, SH);
        hipLaunchKernelGGL(k_cmp, dim3((3ull * total + 255) / 256), dim3(256), 0, 0, dst[0], dst[1], 3ull * total, bad);
        unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); printf("lds vs rowmajor mismatching bytes: %u\n", hb);
    }
    size_t n16 = 3ull * SH * SW / 16;
    timeit("copy 100MB->100MB (16B)", [&](int f) { hipLaunchKernelGGL(k_copy16, dim3((n16 + 255) / 256), dim3(256), 0, 0, (const uint4*)src[f], (uint4*)src[(f + 1) % POOL], n16); });
    return 0;
}
