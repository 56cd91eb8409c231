// pb_kernels_tile.hpp - the fast path for pano / camera sources.
//
//   per frame (pb_remap_u8):   pb_hot_kernel  -> pb_fix_kernel          (same stream)
//   per plan  (pb_plan_create): pb_threshold_kernel -> pb_model_kernel -> pb_certify_kernel
//
// pb_hot_kernel: one WAVE per 32x32 output tile, 4 tiles (a 64x64 block) per workgroup,
// no workgroup barrier.  Math phase: lane = (row, half-row), 16 pixels each, float32 model
// evaluation (pb_tile.hpp), indices into a wave-private LDS tile.  Gather phase: lane = 4
// consecutive pixels x 4 rows, so that one load instruction of the wave covers a compact
// 32x8-pixel patch; unaligned dword loads; one 12-byte store per 4 pixels.
// pb_fix_kernel: the faithful float64 chain for the plan's fix list (whole failed tiles
// and single pixels), overwriting what the hot kernel wrote there.
#pragma once
#include "pb_kernels_faithful.hpp"
#include "pb_tile.hpp"

#define PB_TILE_WAVES 4
typedef unsigned pb_u32x3 __attribute__((ext_vector_type(3)));

struct PbWaveLds {
    int idx[PB_TILE * PB_TILE_PITCH];
};

__device__ __forceinline__ unsigned pb_load_px32(const uint8_t* __restrict__ src, int idx) {
    if (idx < 0) return 0u;
    unsigned v;
    __builtin_memcpy(&v, src + 3ull * (unsigned)idx, 4);  // unaligned dword (the last pixel is special-cased by the caller)
    return v & 0xFFFFFFu;
}

__device__ __forceinline__ int pb_tiles_x(const PbParams& P) { return (P.dst.width + PB_TILE - 1) / PB_TILE; }
__device__ __forceinline__ int pb_tiles_y(const PbParams& P) { return (P.dst.height + PB_TILE - 1) / PB_TILE; }

// block -> 2x2 group of tiles; wave -> tile.  Returns false for waves beyond the image.
__device__ __forceinline__ bool pb_tile_of_wave(const PbParams& P, int wave, int& tx, int& ty) {
    const int gx = (pb_tiles_x(P) + 1) / 2;
    const int by = blockIdx.x / gx, bx = blockIdx.x - by * gx;
    tx = 2 * bx + (wave & 1);
    ty = 2 * by + (wave >> 1);
    return tx < pb_tiles_x(P) && ty < pb_tiles_y(P);
}

// OUT 0: gather + store n_frames frames; OUT 1: write the int32 index map
template <int SRC_KIND, int OUT>
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_hot_kernel(const PbParams P, const PbTileEntry* __restrict__ table,
                                                                     const uint8_t* __restrict__ src,
                                                                     uint8_t* __restrict__ dst, int n_frames,
                                                                     unsigned long long src_stride,
                                                                     unsigned long long dst_stride,
                                                                     int32_t* __restrict__ idx_out) {
    __shared__ PbWaveLds lds[PB_TILE_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tx, ty;
    if (!pb_tile_of_wave(P, wave, tx, ty)) return;  // wave-uniform; no workgroup barriers below
    const PbTileEntry* __restrict__ e = table + ((size_t)ty * pb_tiles_x(P) + tx);
    if (e->flags & PB_TILE_FAILED) return;          // the fix kernel owns this tile
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    PbWaveLds& L = lds[wave];
    {
        const int y = lane & 31, xh = (lane >> 5) * 16;
        PbRowModel R;
        pb_model_row(P, e, X0, Y0, y, xh, R);
#pragma unroll
        for (int k = 0; k < 16; ++k) L.idx[y * PB_TILE_PITCH + xh + k] = pb_model_px<SRC_KIND>(P, R, xh, k);
    }
    pb_wave_sync();
    const int xg = lane & 7, yb = lane >> 3;
    const int W = P.dst.width, H = P.dst.height;
    int id[4][4];
#pragma unroll
    for (int jr = 0; jr < 4; ++jr)
#pragma unroll
        for (int k = 0; k < 4; ++k) id[jr][k] = L.idx[(yb + 8 * jr) * PB_TILE_PITCH + 4 * xg + k];
    const int x = X0 + 4 * xg;
    if (OUT == 1) {
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            const int y = Y0 + yb + 8 * jr;
            if (y < H)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x + k < W) idx_out[(size_t)y * W + x + k] = id[jr][k];
        }
        return;
    }
    const unsigned last_px = (unsigned)P.src.height * (unsigned)P.src.width - 1u;
    for (int f = 0; f < n_frames; ++f) {
        const uint8_t* s = src + (unsigned long long)f * src_stride;
        uint8_t* d = dst + (unsigned long long)f * dst_stride;
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            const int y = Y0 + yb + 8 * jr;
            unsigned a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int v = id[jr][k];
                // a 4-byte read of the frame's very last pixel would touch one byte past the buffer
                a[k] = ((unsigned)v == last_px) ? pb_load_px(s, v) : pb_load_px32(s, v);
            }
            if (y < H) {
                const unsigned long long off = 3ull * ((unsigned long long)y * W + x);
                if (x + 3 < W && (((uintptr_t)d + off) & 3u) == 0) {
                    pb_u32x3 o;
                    o.x = a[0] | (a[1] << 24);
                    o.y = (a[1] >> 8) | (a[2] << 16);
                    o.z = (a[2] >> 16) | (a[3] << 8);
                    *reinterpret_cast<pb_u32x3*>(d + off) = o;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (x + k < W) {
                            d[off + 3 * k + 0] = (uint8_t)(a[k] & 0xFF);
                            d[off + 3 * k + 1] = (uint8_t)((a[k] >> 8) & 0xFF);
                            d[off + 3 * k + 2] = (uint8_t)((a[k] >> 16) & 0xFF);
                        }
                }
            }
        }
    }
}

// The plan's fix list: blocks [0, 4 * n_fail_tiles) take the failed tiles (256 px each), the
// remaining blocks take single pixels (linear output positions).
template <int SRC_KIND, int OUT>
__global__ __launch_bounds__(PB_BLOCK) void pb_fix_kernel(const PbParams P, const int32_t* __restrict__ fail_tiles,
                                                          int n_fail_tiles, const int32_t* __restrict__ fix_px,
                                                          int n_fix_px, const uint8_t* __restrict__ src,
                                                          uint8_t* __restrict__ dst, int n_frames,
                                                          unsigned long long src_stride, unsigned long long dst_stride,
                                                          int32_t* __restrict__ idx_out) {
    int i, j;
    if ((int)blockIdx.x < 4 * n_fail_tiles) {
        const int t = fail_tiles[blockIdx.x >> 2];
        const int ty = t / pb_tiles_x(P), tx = t - ty * pb_tiles_x(P);
        const int local = (blockIdx.x & 3) * 256 + threadIdx.x;
        i = ty * PB_TILE + (local >> 5);
        j = tx * PB_TILE + (local & 31);
        if (i >= P.dst.height || j >= P.dst.width) return;
    } else {
        const unsigned item = (blockIdx.x - 4u * n_fail_tiles) * PB_BLOCK + threadIdx.x;
        if (item >= (unsigned)n_fix_px) return;
        const unsigned p = (unsigned)fix_px[item];
        i = p / (unsigned)P.dst.width;
        j = p - (unsigned)i * (unsigned)P.dst.width;
    }
    const int id = pb_exact_index<SRC_KIND>(P, i, j);
    const size_t p = (size_t)i * P.dst.width + j;
    if (OUT == 1) {
        idx_out[p] = id;
        return;
    }
    for (int f = 0; f < n_frames; ++f) {
        const unsigned v = pb_load_px(src + (unsigned long long)f * src_stride, id);
        uint8_t* o = dst + (unsigned long long)f * dst_stride + 3 * p;
        o[0] = (uint8_t)(v & 0xFF);
        o[1] = (uint8_t)((v >> 8) & 0xFF);
        o[2] = (uint8_t)((v >> 16) & 0xFF);
    }
}

// ---- plan creation -------------------------------------------------------------------
// Bisection for the destination-validity thresholds with the exact predicate.
__global__ void pb_threshold_kernel(const PbParams P, long long* __restrict__ out) {
    const int side = threadIdx.x;  // 0: left / single, 1: right eye of a double destination
    if (side > 1) return;
    const long long wc = (P.dst.kind == PB_KIND_DOUBLE) ? P.dst_half_w : P.dst.width;
    const long long nmax = (wc - 1) * (wc - 1) + (long long)(P.dst.height - 1) * (P.dst.height - 1);
    // first n4 where the lens inverse leaves its domain (asin argument > 1); nmax + 1 if never
    long long lo = 0, hi = nmax + 1;
    while (lo < hi) {
        const long long mid = lo + (hi - lo) / 2;
        bool outside;
        pb_dst_inv_pred(P, mid, side != 0, &outside);
        if (outside) hi = mid; else lo = mid + 1;
    }
    const long long n_dom = lo;
    // first n4 in [0, n_dom) where the pixel is invalid (monotone inside the domain)
    lo = 0;
    hi = n_dom;
    while (lo < hi) {
        const long long mid = lo + (hi - lo) / 2;
        if (pb_dst_inv_pred(P, mid, side != 0, nullptr)) hi = mid; else lo = mid + 1;
    }
    out[2 * side + 0] = lo;      // invalid  <=>  lo <= n4 < n_dom
    out[2 * side + 1] = n_dom;
}

// One wave per tile: 25 faithful node evaluations -> monomial coefficients (float64) ->
// integer anchors + float32 coefficients.
template <int SRC_KIND>
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_model_kernel(const PbParams P, PbTileEntry* __restrict__ table) {
    __shared__ double F[PB_TILE_WAVES][2][25];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tx, ty;
    if (!pb_tile_of_wave(P, wave, tx, ty)) return;
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    PbTileEntry* e = table + ((size_t)ty * pb_tiles_x(P) + tx);
    const double half = 0.5 * (PB_TILE - 1);
    const bool node = lane < 25;
    double f0 = 0.0, f1 = 0.0;
    if (node) {
        const double v = PB_NODE[lane / 5], u = PB_NODE[lane % 5];
        pb_chain_real<SRC_KIND>(P, (double)Y0 + half + half * v, (double)X0 + half + half * u, f0, f1);
        F[wave][0][lane] = f0;
        F[wave][1][lane] = f1;
    }
    bool bad = node && !(fabs(f0) < 1.0e9 && fabs(f1) < 1.0e9);  // NaN / inf / absurd: no model
    // a double-destination tile that straddles the two eyes has no single model
    if (P.dst.kind == PB_KIND_DOUBLE && X0 < P.dst_half_w && X0 + PB_TILE > P.dst_half_w) bad = true;
    pb_wave_sync();
    double c0 = 0.0, c1 = 0.0;
    if (node) {  // lane = m*5+n: C_mn = sum_ij A[m][i] A[n][j] F[i][j]   (i: rows / v, j: columns / u)
        const int m = lane / 5, n = lane % 5;
        for (int i = 0; i < 5; ++i) {
            const double am = PB_A[m][i];
            for (int j = 0; j < 5; ++j) {
                const double w = am * PB_A[n][j];
                c0 = fma(w, F[wave][0][i * 5 + j], c0);
                c1 = fma(w, F[wave][1][i * 5 + j], c1);
            }
        }
    }
    const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
    // anchors from the constant terms (lane 0 holds C_00)
    const double a0 = floor(__shfl(c0, 0)), a1 = floor(__shfl(c1, 0));
    if (node) {
        e->c[0][lane] = any_bad ? 0.0f : (float)(lane == 0 ? c0 - a0 : c0);
        e->c[1][lane] = any_bad ? 0.0f : (float)(lane == 0 ? c1 - a1 : c1);
    }
    if (lane == 0) {
        e->anchor_r = any_bad ? 0 : (int)a0;
        e->anchor_c = any_bad ? 0 : (int)a1;
        e->flags = any_bad ? PB_TILE_FAILED : PB_TILE_HAS_MODEL;
        e->pad0 = 0;
    }
}

// One wave per tile: compares the hot path's index with the faithful one for every pixel of the
// tile; differing pixels go to the fix list, tiles with more than PB_TILE_FAIL_LIMIT of them (or
// without a model) are marked failed.  counters: [0] fix pixels, [1] failed tiles, [2] pixels
// differing in total (statistics), [3] tiles with a model.
template <int SRC_KIND>
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_certify_kernel(const PbParams P, PbTileEntry* __restrict__ table,
                                                                         int32_t* __restrict__ fail_tiles,
                                                                         int32_t* __restrict__ fix_px, unsigned fix_capacity,
                                                                         unsigned* __restrict__ counters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tx, ty;
    if (!pb_tile_of_wave(P, wave, tx, ty)) return;
    const int tile = ty * pb_tiles_x(P) + tx;
    PbTileEntry* e = table + tile;
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    const int y = lane & 31, xh = (lane >> 5) * 16;
    const int i = Y0 + y;
    bool failed = (e->flags & PB_TILE_FAILED) != 0;
    unsigned diff = 0;  // bit k: pixel xh + k of this lane's row differs
    if (!failed) {
        PbRowModel R;
        pb_model_row(P, e, X0, Y0, y, xh, R);
        for (int k = 0; k < 16; ++k) {
            const int j = X0 + xh + k;
            if (i < P.dst.height && j < P.dst.width) {
                const int fast = pb_model_px<SRC_KIND>(P, R, xh, k);
                const int exact = pb_exact_index<SRC_KIND>(P, i, j);
                diff |= (unsigned)(fast != exact) << k;
            }
        }
        unsigned total = __popc(diff);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
        if (lane == 0) {
            atomicAdd(&counters[2], total);
            atomicAdd(&counters[3], 1u);
        }
        if (total > PB_TILE_FAIL_LIMIT) failed = true;
        if (!failed && total) {
            // reserve `total` slots; if the list is full the tile is failed instead
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&counters[0], total);
            base = __shfl(base, 0);
            if (base + total > fix_capacity) {
                failed = true;
            } else {
                // exclusive prefix of per-lane counts
                unsigned mine = __popc(diff), pre = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned t = __shfl_up(pre, o);
                    if (lane >= o) pre += t;
                }
                unsigned pos = base + pre - mine;
                for (int k = 0; k < 16; ++k)
                    if (diff & (1u << k)) fix_px[pos++] = i * P.dst.width + (X0 + xh + k);
            }
        }
    }
    if (failed && lane == 0) {
        e->flags = PB_TILE_FAILED;
        fail_tiles[atomicAdd(&counters[1], 1u)] = tile;
    }
}
