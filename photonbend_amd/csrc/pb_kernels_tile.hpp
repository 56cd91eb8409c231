// pb_kernels_tile.hpp - the fast path for pano / camera sources.
//
//   per call (pb_remap_u8):     pb_hot_win_kernel, ONE launch for all frames (16-byte aligned frames: LDS-DMA windows)
//                               else pb_hot_kernel -> pb_fix_kernel (direct gathers; also the int32 index-map output)
//   per plan  (pb_plan_create): pb_threshold_kernel -> pb_model_kernel -> pb_window_kernel -> pb_certify_kernel
//                               -> pb_fix_tables_kernel (exact lookup tables) -> pb_budget_kernel (LDS budget, tuned)
//
// pb_hot_win_kernel (below, "hot kernel with prefetched LDS windows"): one WAVE per 32x32 output tile, 4 tiles (a
// 64x64 block) per workgroup, no workgroup barrier; tile classes LEAN / DIRECT / BLACK / generic / failed.
// pb_hot_kernel: the same models without windows - math phase: lane = (row, half-row), 16 pixels each, float32
// model evaluation (pb_tile.hpp), indices into a wave-private LDS tile; gather phase: lane = 4 consecutive pixels
// x 4 rows, unaligned dword loads, one 12-byte store per 4 pixels.
// pb_fix_kernel: the plan's fix list (whole failed tiles and single pixels) behind pb_hot_kernel, served from the
// exact-index tables like everything else the models miss; no per-frame kernel runs the float64 chain.
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
__device__ __forceinline__ bool pb_tile_of_wave(const PbParams& P, int wave, int& tx, int& ty, unsigned block = blockIdx.x) {
    const int gx = (pb_tiles_x(P) + 1) / 2, gy = (pb_tiles_y(P) + 1) / 2;
    // XCD-aware order (speed only): blocks with equal id % 8 share an XCD and its L2 (observed round-robin
    // dispatch), and neighbouring tiles share source lines - so each XCD gets whole 4x4-block super-tiles
    // (256x256 px), super-tiles dealt round-robin over the XCDs to keep them balanced.
    const bool pow2 = (gx & (gx - 1)) == 0;  // the common frame sizes: shifts instead of two integer divisions
    const int lg = 31 - __builtin_clz((unsigned)gx);
    if ((gx & 3) == 0 && (gy & 3) == 0 && ((gx * gy) & 127) == 0) {
        const unsigned xcd = block & 7u, slot = block >> 3;
        const unsigned S = (slot >> 4) * 8u + xcd, inner = slot & 15u;   // super-tile id, block inside it
        const unsigned sgx = (unsigned)gx >> 2;
        const unsigned sy = pow2 ? S >> (lg - 2) : S / sgx, sx = S - sy * sgx;
        block = (sy * 4u + (inner >> 2)) * (unsigned)gx + sx * 4u + (inner & 3u);
    }
    const int by = pow2 ? (int)(block >> lg) : (int)block / gx, bx = (int)block - by * gx;
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
        unsigned a[4][4];
#pragma unroll
        for (int jr = 0; jr < 4; ++jr)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int v = id[jr][k];
                // a 4-byte read of the frame's very last pixel would touch one byte past the buffer
                a[jr][k] = ((unsigned)v == last_px) ? pb_load_px(s, v) : pb_load_px32(s, v);
            }
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            const int y = Y0 + yb + 8 * jr;
            if (y < H) {
                const unsigned long long off = 3ull * ((unsigned long long)y * W + x);
                if (x + 3 < W && (((uintptr_t)d + off) & 3u) == 0) {
                    pb_u32x3 o;
                    o.x = a[jr][0] | (a[jr][1] << 24);
                    o.y = (a[jr][1] >> 8) | (a[jr][2] << 16);
                    o.z = (a[jr][2] >> 16) | (a[jr][3] << 8);
                    *reinterpret_cast<pb_u32x3*>(d + off) = o;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (x + k < W) {
                            d[off + 3 * k + 0] = (uint8_t)(a[jr][k] & 0xFF);
                            d[off + 3 * k + 1] = (uint8_t)((a[jr][k] >> 8) & 0xFF);
                            d[off + 3 * k + 2] = (uint8_t)((a[jr][k] >> 16) & 0xFF);
                        }
                }
            }
        }
    }
}

__device__ __forceinline__ int pb_wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int pb_wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int pb_wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// 4 RGB pixels held as the low 3 bytes of a0..a3 -> 12 packed bytes (three v_perm_b32)
__device__ __forceinline__ pb_u32x3 pb_pack_px4(unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    pb_u32x3 o;
    o.x = __builtin_amdgcn_perm(a1, a0, 0x04020100u);  // a0.b0 a0.b1 a0.b2 a1.b0
    o.y = __builtin_amdgcn_perm(a2, a1, 0x05040201u);  // a1.b1 a1.b2 a2.b0 a2.b1
    o.z = __builtin_amdgcn_perm(a3, a2, 0x06050402u);  // a2.b2 a3.b0 a3.b1 a3.b2
    return o;
}

// ---- hot kernel with prefetched LDS windows ---------------------------------------------------------
// The plan entry carries the bounding box of the tile's source samples.  The wave evaluates the tile model,
// issues the whole box as LDS-DMA loads (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, one row
// segment = n16 consecutive lanes, no VGPRs), then gathers every pixel from LDS (aligned dword pair +
// v_alignbyte) and stores 12 packed bytes per 4 pixels; the CU's other waves fill the memory pipeline meanwhile.
//   LEAN tiles (the common case, flagged by the plan builder): the model is anchored at the window
//   origin, so (int)f IS the window row / column: ~11 VALU instructions per pixel, no validity, bounds,
//   wrap or fallback code.
//   Other tiles: generic path - validity thresholds, wrap, per-pixel "inside the window" test with an
//   unaligned global load as fallback.
// Requires frame pointers and strides that are multiples of 16 bytes (else pb_hot_kernel is used).
// LDS window per wave: the plan builder classifies with the largest budget (PB_WINLDS_MAX); the budget in use - the library
// default of 7168 bytes, the caller's choice, or PB_PLAN_TUNE's pick by timing - only moves tiles between the window and the
// direct-gather path (pb_budget_kernel); the hot kernels take it from PbParams::win_budget and use dynamic LDS.
#define PB_WINLDS_MAX 12288
#define PB_DIRECT_LDS_BYTES (33 * 32 * 4)  // regrouping buffer of a DIRECT tile: the smallest usable budget
#define PB_WINLDS_BYTES PB_WINLDS_MAX  // classification budget of the plan builder
extern __shared__ __attribute__((aligned(16))) unsigned pb_dyn_lds[];
__device__ __forceinline__ unsigned* pb_wave_window(const PbParams& P, int wave, int pad_dwords = 4) {
    return pb_dyn_lds + (size_t)wave * ((P.win_budget >> 2) + pad_dwords);
}
// The five numbers a plain tile (LEAN / DIRECT / BLACK) needs besides its entry.  The single-source hot kernel receives them as
// kernel ARGUMENTS - they arrive with the table pointer in the wave's first scalar round trip - while everything else of the
// parameter block stays behind the plan-resident pointer and is fetched only by the tiles that need it (masked, generic, failed).
struct PbHot {
    int32_t dst_w, dst_h, src_w, src_h, win_budget;
};
__device__ __forceinline__ PbHot pb_hot_of(const PbParams& P) { return {P.dst.width, P.dst.height, P.src.width, P.src.height, P.win_budget}; }
static inline PbHot pb_hot_of_host(const PbParams& P) { return {P.dst.width, P.dst.height, P.src.width, P.src.height, P.win_budget}; }
// pb_remap_u8v: a batch whose frames are NOT at a uniform stride (a ring of separately allocated buffers).  The frame pointers travel
// BY VALUE in the kernel-argument segment of the VEC instantiations of the hot kernels - a wave reads its frame's pair with one scalar
// load at a uniform offset, in the same round trip as its tile entry; no device-side table to allocate, fill or guard, so the launch
// stays allocation- and synchronisation-free (graph-capture safe) like pb_remap_u8.  The non-VEC instantiations carry 4 bytes.
#define PB_MAX_VFRAMES 64
struct PbFrameTab {
    const uint8_t* src[PB_MAX_VFRAMES];
    uint8_t* dst[PB_MAX_VFRAMES];
};
struct PbNoFrameTab {
    int unused;
};
template <bool VEC> struct PbFrameTabOf { typedef PbNoFrameTab type; };
template <> struct PbFrameTabOf<true> { typedef PbFrameTab type; };
__device__ __forceinline__ const uint8_t* pb_frame_src(const PbFrameTab& t, unsigned f, const uint8_t*) { return t.src[f]; }
__device__ __forceinline__ uint8_t* pb_frame_dst(const PbFrameTab& t, unsigned f, uint8_t*) { return t.dst[f]; }
__device__ __forceinline__ const uint8_t* pb_frame_src(const PbNoFrameTab&, unsigned, const uint8_t* s) { return s; }
__device__ __forceinline__ uint8_t* pb_frame_dst(const PbNoFrameTab&, unsigned, uint8_t* d) { return d; }
static inline size_t pb_window_lds_bytes(const PbParams& P, int pad_dwords = 4) {
    return (size_t)PB_TILE_WAVES * ((size_t)P.win_budget + 4u * pad_dwords);
}
#ifdef PB_TRACE  // diagnostic build (experiments/diag_trace.py): absolute time of every phase of every tile's wave
// pb_trace[tile * 16 + i]: 0 wave started, 1 entry in SGPRs, 2 addresses computed, 3 loads issued, 4 loads landed,
// 5 stores issued, 6 tile done (stores complete), 7 wave done (incl. fix pixels); 15: HW_ID.  s_memrealtime ticks (10 ns).
__device__ unsigned long long pb_trace[65536 * 16];
__device__ unsigned pb_trace_wpf = 0xFFFFFFFFu, pb_trace_frame = 0;  // workgroups per frame / which frame of a batch to record (pb_debug_trace_frame)
#define PB_TR(i) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (lane == 0 && blockIdx.x / pb_trace_wpf == pb_trace_frame) pb_trace[(size_t)(ty * pb_tiles_x(P) + tx) * 16 + (i)] = t_; } while (0)
#else
#define PB_TR(i)
#endif

// The 12-byte store of four output pixels.  NT: non-temporal.  A single fisheye source is re-read by neighbouring tiles
// and keeps its place in L2 when the write-once output goes around it (c1 13.8 us against 17.4 with plain stores, c3
// batches 29.4 against 33.0); a panorama or a double-fisheye source streams through once, and there plain stores - L2
// merges the 96-byte row pieces of neighbouring tiles into whole lines - are the faster ones (c2 41.6 -> 40.2 us,
// c5shard 73.6 -> 71.1; WRITE_SIZE 1.06x the output against 1.23x).
template <bool NT>
__device__ __forceinline__ void pb_store3(const pb_u32x3 v, uint8_t* p) {
    if (NT)
        __builtin_nontemporal_store(v, reinterpret_cast<pb_u32x3*>(p));
    else
        *reinterpret_cast<pb_u32x3*>(p) = v;
}

// part: this wave issues the row groups part, part + parts, ... (a workgroup loading one window together)
__device__ __forceinline__ void pb_issue_window_loads(const uint8_t* __restrict__ s, unsigned* win, int lane, unsigned gbase,
                                                      unsigned rowbytes, int nrows, int n16, unsigned safe_len, unsigned part = 0,
                                                      unsigned parts = 1) {
    const unsigned pitch = 16u * (unsigned)n16;
    const unsigned inv = (65536u + n16 - 1) / n16;  // lane / n16 for lane < 64
    const unsigned lrow = ((unsigned)lane * inv) >> 16, chunk = (unsigned)lane - lrow * n16;
    const unsigned rpp = 64u / n16;
    const bool lane_on = lrow < rpp;
    for (unsigned rowb = part * rpp; rowb < (unsigned)nrows; rowb += parts * rpp) {
        const unsigned row = rowb + lrow;
        // a row segment starts at the 16-byte boundary at or below its first sample
        const unsigned ga = ((gbase + row * rowbytes) & ~15u) + 16u * chunk;
        if (lane_on && row < (unsigned)nrows && ga + 16u <= safe_len)
            __builtin_amdgcn_global_load_lds(s + ga, (__attribute__((address_space(3))) void*)(win + ((rowb * pitch) >> 2)), 16, 0, 0);
    }
}

// One tile of the windowed hot kernel (the four tile classes); returns when the tile's pixels are stored.
#define PB_NT_DEFAULT(kind) ((kind) == PB_KIND_CAMERA)
template <int SRC_KIND, bool NT = PB_NT_DEFAULT(SRC_KIND)>
__device__ __forceinline__ void pb_win_tile(const PbParams& P, const PbHot& Hd, const PbTileEntry* __restrict__ e, const int flags, const int tx,
                                            const int ty, const int lane, unsigned* win, const uint8_t* __restrict__ src,
                                            uint8_t* __restrict__ dst, const int n_frames, const unsigned long long src_stride,
                                            const unsigned long long dst_stride) {
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    const unsigned rowbytes = 3u * (unsigned)Hd.src_w;
    const unsigned frame_bytes = rowbytes * (unsigned)Hd.src_h;       // < 2^31 (host check)
    const unsigned safe_len = frame_bytes & ~15u;                     // every 16-byte chunk below this is loadable
    const int xg = lane & 7, yb = lane >> 3;
    const int W = Hd.dst_w, H = Hd.dst_h;
    const int x = X0 + 4 * xg;

#ifdef PB_ABLATION  // timing experiments only (experiments/): skipped work = wrong pixels; never compiled into the product
    if ((P.exp_flags & 4) && (flags & PB_TILE_DIRECT)) return;
    if ((P.exp_flags & 8) && (flags & PB_TILE_LEAN)) return;
    if ((P.exp_flags & 64) && !(flags & (PB_TILE_LEAN | PB_TILE_DIRECT | PB_TILE_BLACK))) return;
    if ((P.exp_flags & 128) && (flags & PB_TILE_BLACK)) return;
#define PB_ABL_NO_STORE(v) ((P.exp_flags & 16) && (v) != 0x12345678u)
#define PB_ABL_NO_LOAD (P.exp_flags & 32)
#else
#define PB_ABL_NO_STORE(v) false
#define PB_ABL_NO_LOAD false
#endif
    // where the lane's 12-byte groups go: 4 pixels at (st_x, st_y0 + st_dy * jr), jr = 0..3
    int st_x = x, st_y0 = Y0 + yb, st_dy = 8;
#ifdef PB_ABLATION  // PB_EXP bit 1024: the stores of a 64 x 16 tile shape (whole 192-byte row pieces per instruction; pixels land in the wrong places)
    if (P.exp_flags & 1024) {
        st_x = 64 * (tx >> 1) + 4 * (lane & 15);
        st_y0 = Y0 + 16 * (tx & 1) + (lane >> 4);
        st_dy = 4;
    }
#endif
    if (flags & PB_TILE_BLACK) {
        for (int f = 0; f < n_frames; ++f) {
            uint8_t* d = dst + (unsigned long long)f * dst_stride;
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                const int y = Y0 + yb + 8 * jr;
                if (y >= H) continue;
                const unsigned long long off = 3ull * ((unsigned long long)y * W + x);
                if (x + 3 < W && (((uintptr_t)d + off) & 3u) == 0) {
                    const pb_u32x3 z = {0u, 0u, 0u};
                    *reinterpret_cast<pb_u32x3*>(d + off) = z;
                } else {
                    for (int k = 0; k < 4; ++k)
                        if (x + k < W) d[off + 3 * k] = d[off + 3 * k + 1] = d[off + 3 * k + 2] = 0;
                }
            }
        }
        return;
    }
    if (flags & PB_TILE_DIRECT) {
        // The gathers run along the tile direction in which the SOURCE ROW changes least - lane = pixel column
        // (or row), 16 loads down the other direction - so that one load instruction touches few 64-byte lines
        // whatever the tile's orientation in the source; the samples are then regrouped for the 12-byte stores
        // (4 consecutive pixels x 4 rows per lane) through the wave's LDS window, which a DIRECT tile does not
        // otherwise use (PB_DIRECT_LDS_BYTES of it).  Measured -6 % on c2.  The model is evaluated in one of the two
        // certified orders: row-first (collapse along v, Horner in u) or column-first.
        const unsigned gbase = (unsigned)e->anchor_r * rowbytes + 3u * (unsigned)e->anchor_c;
        const bool along_x = fabsf(e->c[1][0]) <= fabsf(e->c[5][0]);  // |d row / du| <= |d row / dv|
        const int p = lane & 31, hh = lane >> 5;
        // shear: the 32 lanes of a half-wave follow the line of constant source row through the tile (the pixel
        // of lane p in pass n is (p, (2n + hh + shift(p)) mod 32), a bijection of the tile): c3 -7 %, c2 unchanged
        const float num = along_x ? e->c[1][0] : e->c[5][0], den = along_x ? e->c[5][0] : e->c[1][0];
        const float slope = (den != 0.0f) ? -num / den : 0.0f;
        const int shift = (int)rintf(slope * ((float)p - 15.5f));
        // MASKED tiles: bit n of `dead` = pixel n of this lane is an invalid destination pixel (exact integer test, as pb_model_row's)
        unsigned dead = 0u;
        if (flags & PB_TILE_MASKED) {
            const int side = (P.dst.kind == PB_KIND_DOUBLE) && (X0 >= P.dst_half_w);
            const int wc = (P.dst.kind == PB_KIND_DOUBLE) ? P.dst_half_w : P.dst.width;
            const long long lo = P.inv_lo[side], hi = P.inv_hi[side];
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int q = (2 * n + hh + shift) & 31;
                const int px = along_x ? p : q, py = along_x ? q : p;
                const long long x2 = 2ll * (X0 + px - (side ? P.dst_half_w : 0)) - (wc - 1), y2 = (long long)(P.dst.height - 1) - 2ll * (Y0 + py);
                const long long n4 = x2 * x2 + y2 * y2;
                dead |= (unsigned)(n4 >= lo && n4 < hi) << n;
            }
        }
        unsigned go[16];
#ifdef PB_ABLATION  // VALU sensitivity of the direct-gather path (wrong pixels): PB_EXP bit 2 = no polynomial, addresses from the lane id
        if (P.exp_flags & 2) {
#pragma unroll
            for (int n = 0; n < 16; ++n) go[n] = gbase + ((unsigned)((2 * n + hh) & 31) % (unsigned)e->win_rows) * rowbytes + 3u * ((unsigned)p % (unsigned)e->win_cols);  // inside the tile's own box
        } else
#endif
        if (along_x) {
            // column-first evaluation (one collapse per lane instead of one per pixel; certified by pb_certify_kernel
            // alongside the row-first order)
            pb_f2 bcol[5];
            pb_collapse_col(e, p, bcol);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const pb_f2 fv = pb_eval_row(bcol, pb_tile_coord((2 * n + hh + shift) & 31));
                go[n] = gbase + (unsigned)(int)fv.x * rowbytes + __umul24((unsigned)(int)fv.y, 3u);
            }
        } else {
            pb_f2 a[5];
            pb_collapse_row(e, p, a);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const pb_f2 fv = pb_eval_row(a, pb_tile_coord((2 * n + hh + shift) & 31));
                go[n] = gbase + (unsigned)(int)fv.x * rowbytes + __umul24((unsigned)(int)fv.y, 3u);
            }
        }
#ifdef PB_ABLATION  // VALU sensitivity, same addresses: PB_EXP bit 1 = the whole model arithmetic a second time (results kept alive, unused)
        if (P.exp_flags & 1) {
            pb_f2 a2[5];
            pb_collapse_row(e, p ^ 1, a2);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const pb_f2 fv = pb_eval_row(a2, pb_tile_coord((2 * n + hh + shift + 1) & 31));
                const unsigned g2 = gbase + (unsigned)(int)fv.x * rowbytes + __umul24((unsigned)(int)fv.y, 3u);
                asm volatile("" ::"v"(g2));
            }
        }
#endif
        PB_TR(2);
        for (int f = 0; f < n_frames; ++f) {
            const uint8_t* s = src + (unsigned long long)f * src_stride;
            uint8_t* d = dst + (unsigned long long)f * dst_stride;
            unsigned t[16];
            if (PB_ABL_NO_LOAD) {
#pragma unroll
                for (int n = 0; n < 16; ++n) t[n] = go[n];
            } else if (flags & PB_TILE_MASKED) {
#pragma unroll
                for (int n = 0; n < 16; ++n) {
                    t[n] = 0u;
                    if (!((dead >> n) & 1u)) __builtin_memcpy(&t[n], s + go[n], 4);
                }
            } else {
#pragma unroll
                for (int n = 0; n < 16; ++n) __builtin_memcpy(&t[n], s + go[n], 4);
            }
#ifdef PB_TRACE
            PB_TR(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PB_TR(4);
#endif
            // park as [y][x] with a 33-dword pitch, read back as 4 consecutive pixels x 4 rows per lane
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int q = (2 * n + hh + shift) & 31;
                win[along_x ? q * 33 + p : p * 33 + q] = t[n];
            }
            pb_wave_sync();
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                const unsigned* r = win + (yb + 8 * jr) * 33 + 4 * xg;
                const unsigned long long off = 3ull * ((unsigned long long)(st_y0 + st_dy * jr) * W + st_x);
                if (PB_ABL_NO_STORE(r[0])) continue;
                if ((((uintptr_t)d + off) & 3u) == 0) {
                    pb_store3<NT>(pb_pack_px4(r[0], r[1], r[2], r[3]), d + off);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        d[off + 3 * k + 0] = (uint8_t)(r[k] & 0xFF);
                        d[off + 3 * k + 1] = (uint8_t)((r[k] >> 8) & 0xFF);
                        d[off + 3 * k + 2] = (uint8_t)((r[k] >> 16) & 0xFF);
                    }
                }
            }
            PB_TR(5);
            pb_wave_sync();
        }
        return;
    }
    if (flags & PB_TILE_LEAN) {
        const int nrows = e->win_rows, n16 = e->win_n16;
        const unsigned a0 = (unsigned)e->win_a0, pitch = 16u * (unsigned)n16;
        const unsigned gbase = (unsigned)e->anchor_r * rowbytes + 3u * (unsigned)e->anchor_c;
        float u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = pb_tile_coord(4 * xg + k);
        for (int f = 0; f < n_frames; ++f) {
            const uint8_t* s = src + (unsigned long long)f * src_stride;
            uint8_t* d = dst + (unsigned long long)f * dst_stride;
            unsigned la[4][4];
#ifdef PB_ABLATION  // VALU sensitivity (wrong pixels): bit 1 = one collapse instead of four, bit 2 = no polynomial at all
            if (P.exp_flags & 3) {
                pb_f2 a[5];
                pb_collapse_row(e, yb, a);
#pragma unroll
                for (int jr = 0; jr < 4; ++jr)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const pb_f2 fv = (P.exp_flags & 2) ? a[(jr + k) % 5] : pb_eval_row(a, u[k]);
                        const unsigned dr = (unsigned)(int)fv.x % (unsigned)nrows, dc = (unsigned)(int)fv.y % (unsigned)e->win_cols;
                        la[jr][k] = __umul24(dr, pitch) + (__umul24(dc, 3u) + a0);
                    }
            } else
#endif
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                pb_f2 a[5];
                pb_collapse_row(e, yb + 8 * jr, a);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const pb_f2 fv = pb_eval_row(a, u[k]);
                    const unsigned dr = (unsigned)(int)fv.x, dc = (unsigned)(int)fv.y;  // >= 0: truncation == floor
                    la[jr][k] = __umul24(dr, pitch) + (__umul24(dc, 3u) + a0);
                }
            }
            // the window loads are issued AFTER the model math: measured 2 % faster on c2 than issuing them first
            // and computing while they fly (the other waves of the CU keep the memory pipeline busy anyway)
            asm volatile("" ::: "memory");
            PB_TR(2);
            if (!PB_ABL_NO_LOAD) pb_issue_window_loads(s, win, lane, gbase, rowbytes, nrows, n16, safe_len);
            PB_TR(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pb_wave_sync();
            PB_TR(4);
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                unsigned a[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned l = la[jr][k];
                    a[k] = __builtin_amdgcn_alignbyte(win[(l >> 2) + 1], win[l >> 2], l);
                }
                // LEAN tiles lie fully inside the image; the 12-byte store is 4-byte aligned when the base is
                const unsigned long long off = 3ull * ((unsigned long long)(st_y0 + st_dy * jr) * W + st_x);
                if (PB_ABL_NO_STORE(a[0])) continue;
                if ((((uintptr_t)d + off) & 3u) == 0) {
                    pb_store3<NT>(pb_pack_px4(a[0], a[1], a[2], a[3]), d + off);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        d[off + 3 * k + 0] = (uint8_t)(a[k] & 0xFF);
                        d[off + 3 * k + 1] = (uint8_t)((a[k] >> 8) & 0xFF);
                        d[off + 3 * k + 2] = (uint8_t)((a[k] >> 16) & 0xFF);
                    }
                }
            }
            PB_TR(5);
            // no wait for the stores: their data left the registers at issue; the next frame's window loads only
            // need this frame's LDS reads done (their results were consumed by the packing above)
            pb_wave_sync();  // the window is overwritten by the next frame's loads
        }
        return;
    }

    // ---- generic tile ---------------------------------------------------------------------------------
    const int r0 = e->win_r0, c0 = e->win_c0;
    int nrows = e->win_rows;
    const unsigned gbase = (unsigned)r0 * rowbytes + 3u * (unsigned)c0;   // byte offset of the box's first sample
    const unsigned a0 = gbase & 15u;                                      // frames are 16-byte aligned, rows may not be
    int n16 = 1;
    if (nrows > 0) {
        n16 = (3 * e->win_cols + 15 + 1 + 15) >> 4;  // + worst-case alignment slack + 1 byte for the dword reads
        if (n16 > 64) n16 = 64;
        const int cap = Hd.win_budget / (16 * n16);
        if (nrows > cap) nrows = cap;
    }
    const unsigned pitch = 16u * (unsigned)n16;
    const unsigned rb16 = rowbytes & 15u;  // row-to-row change of the alignment offset
    for (int f = 0; f < n_frames; ++f) {
        const uint8_t* s = src + (unsigned long long)f * src_stride;
        uint8_t* d = dst + (unsigned long long)f * dst_stride;
        if (nrows > 0) pb_issue_window_loads(s, win, lane, gbase, rowbytes, nrows, n16, safe_len);
        int rc[4][4];
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            PbRowModel R;
            pb_model_row(P, e, X0, Y0, yb + 8 * jr, 4 * xg, R);
#pragma unroll
            for (int k = 0; k < 4; ++k) rc[jr][k] = pb_model_px_rc<SRC_KIND>(P, R, 4 * xg, k);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pb_wave_sync();
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            const int y = Y0 + yb + 8 * jr;
            unsigned a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int v = rc[jr][k];
                unsigned px = 0;
                if (v >= 0) {
                    const unsigned r = (unsigned)v >> 16, c = (unsigned)v & 0xFFFFu;
                    const unsigned row = r - (unsigned)r0;
                    const unsigned g = r * rowbytes + 3u * c;
                    const unsigned off = 3u * (c - (unsigned)c0) + ((a0 + row * rb16) & 15u);
                    if (row < (unsigned)nrows && off + 4u <= pitch && g + 4u <= safe_len) {
                        const unsigned l = row * pitch + off;
                        px = __builtin_amdgcn_alignbyte(win[(l >> 2) + 1], win[l >> 2], l) & 0xFFFFFFu;
                    } else if (g + 4u <= frame_bytes) {
                        unsigned t;
                        __builtin_memcpy(&t, s + g, 4);
                        px = t & 0xFFFFFFu;
                    } else {
                        px = (unsigned)s[g] | ((unsigned)s[g + 1] << 8) | ((unsigned)s[g + 2] << 16);
                    }
                }
                a[k] = px;
            }
            if (y < H) {
                const unsigned long long off = 3ull * ((unsigned long long)y * W + x);
                if (x + 3 < W && (((uintptr_t)d + off) & 3u) == 0) {
                    pb_store3<NT>(pb_pack_px4(a[0], a[1], a[2], a[3]), d + off);
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
        pb_wave_sync();
    }
}

// The whole 256-byte tile entry in ONE scalar round trip (four s_load_dwordx16 in flight together), instead of
// flags -> window fields -> coefficients as the uses come up: the wave's first source load is issued one memory
// latency earlier.  The copy lives in SGPRs (every access below has a constant index).
typedef int pb_i32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void pb_load_entry(const PbTileEntry* __restrict__ e, PbTileEntry& L) {
    pb_i32x16 q0, q1, q2, q3;
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\t"
                 "s_load_dwordx16 %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3)
                 : "s"(e)
                 : "memory");
    int* w = reinterpret_cast<int*>(&L);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        w[i] = q0[i];
        w[16 + i] = q1[i];
        w[32 + i] = q2[i];
        w[48 + i] = q3[i];
    }
}

// Exact-index tables (built once per plan by pb_fix_tables_kernel from the faithful chain): what the tile models
// cannot reproduce is not recomputed per frame but LOOKED UP -
//   idx_tab  the int32 source index (-1 = black) of every pixel of every failed tile, 4 KiB per tile, slot
//            PbTileEntry::aux_off;
//   fix_idx  the source index of every pixel of the fix list (parallel to fix_px).
// With them the hot launch is the only launch of a frame: a failed tile is gathered by its own wave through its
// index slot, a tile's fix pixels are re-copied by its wave after its stores; no float64 in the kernel.
template <bool ONE>
__device__ __forceinline__ void pb_failed_tile(const PbHot& Hd, const PbTileEntry* __restrict__ e, const int tx, const int ty,
                                               const int lane, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                               const int n_frames, const unsigned long long src_stride,
                                               const unsigned long long dst_stride, const int32_t* __restrict__ idx_tab) {
    // failed tile: gather through the plan's exact indices (lane = 4 consecutive pixels x 4 rows)
    const int xg = lane & 7, yb = lane >> 3;
    const int W = Hd.dst_w, H = Hd.dst_h;
    const int x = tx * PB_TILE + 4 * xg;
    const int32_t* __restrict__ slot = idx_tab + (size_t)e->aux_off * (PB_TILE * PB_TILE);
    int id[4][4];
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        const int4 v = *reinterpret_cast<const int4*>(slot + (yb + 8 * jr) * PB_TILE + 4 * xg);
        id[jr][0] = v.x; id[jr][1] = v.y; id[jr][2] = v.z; id[jr][3] = v.w;
    }
    const unsigned last_px = (unsigned)Hd.src_h * (unsigned)Hd.src_w - 1u;
    for (int f = 0; f < n_frames; ++f) {
        const uint8_t* s = src + (unsigned long long)f * src_stride;
        uint8_t* d = dst + (unsigned long long)f * dst_stride;
        // branch-free, so that the 16 gathers of a lane are in flight together: a black pixel (-1) reads pixel 0
        // and is masked; the frame's very last pixel is read one byte early (a 4-byte read at its own address
        // would touch one byte past the buffer) and shifted
        unsigned a[4][4];
#pragma unroll
        for (int jr = 0; jr < 4; ++jr)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int v = id[jr][k];
                const bool last = (unsigned)v == last_px;
                const unsigned off = v < 0 ? 0u : 3u * (unsigned)v - (last ? 1u : 0u);
                unsigned t;
                __builtin_memcpy(&t, s + off, 4);
                a[jr][k] = v < 0 ? 0u : (last ? t >> 8 : t);
            }
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            const int y = ty * PB_TILE + yb + 8 * jr;
            if (y >= H) continue;
            const unsigned long long off = 3ull * ((unsigned long long)y * W + x);
            if (x + 3 < W && (((uintptr_t)d + off) & 3u) == 0) {
                pb_store3<true>(pb_pack_px4(a[jr][0], a[jr][1], a[jr][2], a[jr][3]), d + off);  // failed tiles are few: policy irrelevant
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x + k < W) {
                        d[off + 3 * k + 0] = (uint8_t)(a[jr][k] & 0xFF);
                        d[off + 3 * k + 1] = (uint8_t)((a[jr][k] >> 8) & 0xFF);
                        d[off + 3 * k + 2] = (uint8_t)((a[jr][k] >> 16) & 0xFF);
                    }
            }
        }
    }
}

// The hot kernel: one wave per 32x32 tile, one launch per pb_remap_u8 call whatever the number of frames.
// Frames of a batch are a GRID dimension (workgroup = (frame, 2x2 tile group); frame-major, so the XCD residue of a
// tile group is the same in every frame): the launch ramp, the drain of the last waves and the gap between dependent
// launches - about 5 of a c2 frame's 43 us (experiments/diag_trace.py) - are paid once per batch instead of once per
// frame, and the index math (0.35 us of a wave's 6.3) is simply repeated.  A wave that loops over the frames of its
// tile instead has a lifetime, and therefore a drain, n_frames times as long: measured no faster than single launches.
// (Also measured and rejected, see experiments/README.md: waves that work through 2-8 tiles with the next entry
// prefetched - the entry's scalar round trip disappears from the timeline but the frame gets slower, the workgroups
// become too coarse for the hardware's dealing; the four waves of a workgroup sharing one source window; aligned
// 8-byte loads in the direct-gather tiles.)
// The parameter block reaches this kernel through a pointer to the plan's device copy (120 bytes of kernel arguments instead of
// 1216): measured on MI355X (round 3, experiments/session_r3_g.sh, three alternating pairs of processes) c2 40.6-41.2 -> 39.8-40.2 us,
// c1 13.9-14.7 -> 13.7-13.9, c3 +-0, the all-tiles-skipped launch of 4096 workgroups 5.9-7.1 -> 5.9-6.7.  The double-fisheye kernel
// keeps the block by value: by pointer measured 1.4 % SLOWER there (58.4-58.9 -> 59.5-59.8 us).
// VEC (pb_remap_u8v): frame f of the batch is (vtab.src[f], vtab.dst[f]) instead of src / dst + f * stride.
template <int SRC_KIND, bool VEC = false>
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_hot_win_kernel(const PbParams* __restrict__ Pp, const PbHot Hd, const PbTileEntry* __restrict__ table,
                                                                         const uint8_t* __restrict__ src,
                                                                         uint8_t* __restrict__ dst, const unsigned groups_per_frame,
                                                                         unsigned long long src_stride,
                                                                         unsigned long long dst_stride,
                                                                         const int32_t* __restrict__ idx_tab,
                                                                         const int32_t* __restrict__ fix_px,
                                                                         const int32_t* __restrict__ fix_idx, const unsigned n_frames,
                                                                         const typename PbFrameTabOf<VEC>::type vtab) {
    const PbParams& P = *Pp;
    // every kernel argument the tile prologue needs, in one scalar round trip (the compiler would otherwise fetch
    // the table pointer only after the tile index is known: a second dependent trip per wave)
    asm volatile("" ::"s"(table), "s"(Hd.dst_w), "s"(Hd.dst_h), "s"(Hd.src_w), "s"(Hd.src_h), "s"(Hd.win_budget), "s"(groups_per_frame));
    const int lane = threadIdx.x & 63;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // A workgroup is `wpw` waves (1, 2 or 4; blockDim.x / 64).  LDS is released per WORKGROUP: with four waves the
    // window of a wave that has finished idles until the slowest of the four is done.  Tile groups (2x2 tiles) keep
    // their XCD: workgroup id -> (XCD residue, slot); the slot's waves continue the XCD's list of tile groups.
#ifdef PB_ABLATION
    const unsigned wpw = blockDim.x >> 6;
#else
    constexpr unsigned wpw = PB_TILE_WAVES;  // (blockDim.x is a hidden kernel argument of its own cache line: a round trip and a division saved)
#endif
    unsigned wg = blockIdx.x;
    const unsigned wgs_per_frame = groups_per_frame * (4u / wpw);
    if (VEC) {  // a batch of separately allocated frames: the frame's pointers from the kernel-argument table
        const unsigned f = wg / wgs_per_frame;
        wg -= f * wgs_per_frame;
        src = pb_frame_src(vtab, f, src);
        dst = pb_frame_dst(vtab, f, dst);
    } else if (wg >= wgs_per_frame) {  // a batch: which frame
        const unsigned f = wg / wgs_per_frame;
        wg -= f * wgs_per_frame;
        src += (unsigned long long)f * src_stride;
        dst += (unsigned long long)f * dst_stride;
    }
    // wave slot in launch order: `table` is the plan's LAUNCH-ORDER copy of the tile entries (pb_launch_table_kernel) -
    // workgroup ids keep their XCD residue, XCD x takes the super-tiles x, x + 8, ... of the output (neighbours in space
    // are neighbours in time and share an L2).  The entry itself says which tile it is: no index arithmetic.
    const unsigned flat = (wg >> 3) * wpw + (unsigned)wave_in_wg;
    const unsigned vslot = ((flat >> 2) * 8u + (wg & 7u)) * 4u + (flat & 3u);
    PbTileEntry entry;
    pb_load_entry(table + vslot, entry);
    const int tx = entry.tile_xy & 0xFFFF, ty = (int)((unsigned)entry.tile_xy >> 16);
    if (entry.flags & PB_TILE_SKIP) return;
    PB_TR(0);
#ifdef PB_TRACE
    if (lane == 0 && blockIdx.x / pb_trace_wpf == pb_trace_frame) {
        pb_trace[(size_t)(ty * pb_tiles_x(P) + tx) * 16 + 15] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
        pb_trace[(size_t)(ty * pb_tiles_x(P) + tx) * 16 + 14] = (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u) | ((unsigned long long)blockIdx.x << 8);  // HW_REG_XCC_ID, workgroup
    }
#endif
    const PbTileEntry* __restrict__ e = &entry;
    const int flags = e->flags;
    PB_TR(1);
    if (flags & PB_TILE_FAILED) {
        pb_failed_tile<true>(Hd, e, tx, ty, lane, src, dst, 1, src_stride, dst_stride, idx_tab);
#ifdef PB_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PB_TR(6);
        PB_TR(7);
#endif
        return;
    }
    pb_win_tile<SRC_KIND>(P, Hd, e, flags, tx, ty, lane, pb_dyn_lds + (size_t)wave_in_wg * ((Hd.win_budget >> 2) + 4), src, dst, 1, src_stride, dst_stride);
#ifdef PB_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PB_TR(6);
#endif
    // this tile's fix pixels (where the model's truncation differs from the faithful one): re-copied through
    // their exact indices after the wave's own stores have completed
    const int n_fix = e->fix_cnt;
    if (n_fix > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < n_fix) {
            const unsigned p = (unsigned)fix_px[e->fix_off + lane];
            const int id = fix_idx[e->fix_off + lane];
            const unsigned v = pb_load_px(src, id);
            uint8_t* o = dst + 3ull * p;
            o[0] = (uint8_t)(v & 0xFF);
            o[1] = (uint8_t)((v >> 8) & 0xFF);
            o[2] = (uint8_t)((v >> 16) & 0xFF);
        }
    }
#ifdef PB_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PB_TR(7);
#endif
}

// Plan creation: applies an LDS budget to the classification made with PB_WINLDS_MAX.  saved = the flags as
// classified (and certified); a LEAN tile whose window exceeds the budget takes the direct-gather path (same model,
// same anchors, same pixels).  counters[0] = LEAN tiles, [1] = DIRECT tiles after the change.
__global__ void pb_budget_kernel(PbTileEntry* __restrict__ table, const int32_t* __restrict__ saved, unsigned n_tiles, int budget,
                                 unsigned* __restrict__ counters) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    int flags = saved[t];
    if ((flags & PB_TILE_LEAN) && table[t].win_rows * 16 * table[t].win_n16 > budget) flags = (flags & ~PB_TILE_LEAN) | PB_TILE_DIRECT;
    table[t].flags = flags;
    if (flags & PB_TILE_LEAN) atomicAdd(&counters[0], 1u);
    if (flags & PB_TILE_DIRECT) atomicAdd(&counters[1], 1u);
}
// Plan creation / budget change: the launch-order copy of the tile table.  One wave per wave slot of the hot launch:
// slot v = virtual workgroup (v >> 2) x wave (v & 3).  Virtual workgroup B runs on XCD B & 7 (round-robin dispatch,
// speed only) as that XCD's (B >> 3)-th workgroup; unit_of[xcd * units_per_xcd + k] names the k-th unit (a 4x4 group of
// workgroups = a 256x256-px super-tile, or -1 = none) the host's ordering gave that XCD; without units (unit_of ==
// nullptr: grids that do not divide into super-tiles) virtual workgroup B is tile group B.
// -> the 2 x 2 tile group of virtual workgroup B (row-major over the grid of groups), or -1 = none
__device__ __forceinline__ long long pb_group_of_workgroup(const PbParams& P, unsigned B, const int* __restrict__ unit_of, int units_per_xcd, int unit_side,
                                                           int unit_side_y) {
    const int gx = (pb_tiles_x(P) + 1) / 2, gy = (pb_tiles_y(P) + 1) / 2;
    long long group = B;
    if (unit_of) {
        // a unit is U x UY workgroups (UY = U unless the caller says otherwise), walked row by row
        const unsigned U = (unsigned)unit_side, UY = unit_side_y ? (unsigned)unit_side_y : U, xcd = B & 7u, slot = B >> 3, k = slot / (U * UY), inner = slot - k * U * UY;
        const int S = (int)k < units_per_xcd ? unit_of[xcd * units_per_xcd + k] : -1;
        const int sgx = gx / (int)U;
        group = S < 0 ? -1 : (long long)((S / sgx) * (int)UY + (int)(inner / U)) * gx + (S % sgx) * (int)U + (int)(inner % U);
    }
    return (group >= 0 && group < (long long)gx * gy) ? group : -1;
}
// Double-fisheye plan (table = the left eye's entries): a tile that sees ONE eye - the other eye's tile BLACK, unit blend weights,
// nothing on either fix list - is a plain camera-source tile (PB_TILE_SOLO: the single-source tile code with the live eye's entry).
// fl / fr: the eyes' flags, nl / nr: their fix counts -> 0 both eyes, 1 the left eye alone, 2 the right eye alone
__device__ __forceinline__ int pb_eye_class(int fl, int fr, int nl, int nr) {
    const int plain = PB_TILE_LEAN | PB_TILE_DIRECT;
    const bool solo_ok = (fl & PB_TILE_W_UNIT_BIT) && nl == 0 && nr == 0;
    const bool solo_l = solo_ok && (fr & PB_TILE_BLACK) && (fl & (plain | PB_TILE_BLACK));
    const bool solo_r = solo_ok && !solo_l && (fl & PB_TILE_BLACK) && (fr & plain);
    return solo_l ? 1 : (solo_r ? 2 : 0);
}
#define PB_SOLO_KEEP (PB_TILE_LEAN | PB_TILE_DIRECT | PB_TILE_BLACK | PB_TILE_COARSE | PB_TILE_TD3 | PB_TILE_TAB_Y | PB_TILE_TAB_PLAIN)
__global__ __launch_bounds__(256) void pb_launch_table_kernel(const PbParams P, const PbTileEntry* __restrict__ table,
                                                              PbTileEntry* __restrict__ ltable, const int* __restrict__ unit_of,
                                                              int units_per_xcd, unsigned n_slots, int unit_side,
                                                              const PbTileEntry* __restrict__ table_r = nullptr, int unit_side_y = 0) {
    const unsigned v = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (v >= n_slots) return;
    const unsigned B = v >> 2, wave = v & 3u;
    const int gx = (pb_tiles_x(P) + 1) / 2;
    const long long group = pb_group_of_workgroup(P, B, unit_of, units_per_xcd, unit_side, unit_side_y);
    int tx = -1, ty = -1;
    if (group >= 0) {
        const int by = (int)(group / gx), bx = (int)(group - (long long)by * gx);
        tx = 2 * bx + (int)(wave & 1u);
        ty = 2 * by + (int)(wave >> 1);
    }
    int* out = reinterpret_cast<int*>(ltable + v);
    if (tx < 0 || tx >= pb_tiles_x(P) || ty >= pb_tiles_y(P)) {
        out[lane] = lane == 2 ? PB_TILE_SKIP : 0;  // flags is the third dword
        return;
    }
    const int* in = reinterpret_cast<const int*>(table + ((size_t)ty * pb_tiles_x(P) + tx));
    int w = in[lane];
    if (table_r) {
        // double-fisheye plan: a one-eye tile's slot carries the live eye's entry (PB_TILE_SOLO), which the hot kernel takes with
        // scalar loads and runs through the single-source tile code.  Every other slot only names its tile: the two-eye path reads
        // both entries itself.
        const int wr = reinterpret_cast<const int*>(table_r + ((size_t)ty * pb_tiles_x(P) + tx))[lane];
        const int FL = offsetof(PbTileEntry, flags) / 4, FC = offsetof(PbTileEntry, fix_cnt) / 4;
        const int cls = pb_eye_class(__shfl(w, FL), __shfl(wr, FL), __shfl(w, FC), __shfl(wr, FC));
        if (cls) {
            if (cls == 2) w = wr;
            if ((int)lane == FL) w = (w & PB_SOLO_KEEP) | PB_TILE_SOLO | (cls == 2 ? PB_TILE_EYE_R : 0);
        } else {
            w = 0;
        }
    }
    if (lane == offsetof(PbTileEntry, tile_xy) / 4) w = (ty << 16) | tx;
    out[lane] = w;
}

// ---- the PAIR layout: the launch-order table of a double-fisheye plan's bilinear mode (round 6) ---------------------------------
// A tile that samples BOTH eyes is served by a PAIR of waves of one workgroup - wave L the left eye, wave R the right eye, each with its
// eye's entry in its own slot, its own LDS region, the register state of a one-eye tile; R hands its sixteen samples per lane to L
// through LDS, L blends and stores (pb_bilinear_double_hot_kernel).  Round 5's single wave staged the two eyes' windows one after the
// other and kept both eyes' samples live: 168 VGPRs, three waves per SIMD, 6.6 ns per two-eye tile against 2.0 for a one-eye tile.
// Workgroups are homogeneous: a PAIR workgroup holds two pairs (slots L R L R; a missing pair = two PB_TILE_SKIP | PB_TILE_TWO pads,
// which still meet the workgroup's barrier), a SOLO workgroup up to four one-eye tiles - so a pair's barrier never waits for a
// stranger.  A 2 x 2 tile group with k two-eye tiles becomes ceil(k / 2) pair workgroups followed by one solo workgroup (k < 4),
// in the XCD's list where the group's single workgroup stood: neighbours in space stay neighbours in time on the same L2.
//   pb_pair_count_kernel   workgroups per virtual workgroup B of the plain layout (0 - 3)
//   pb_pair_scan_kernel    wave x: the exclusive scan of those counts over XCD x's list (B = x, x + 8, ...) and its total
//   pb_pair_table_kernel   one wave per B: writes its workgroups' slots at ((start[B] + j) * 8 + (B & 7)) * 4
// The table is pre-filled with PB_TILE_SKIP slots (pb_skip_fill_kernel): the XCDs' lists differ in length.
__device__ __forceinline__ void pb_group_classes(const PbParams& P, const PbTileEntry* __restrict__ table_l, const PbTileEntry* __restrict__ table_r, long long group,
                                                 int cls[4], int& n_two, int& n_solo) {
    const int gx = (pb_tiles_x(P) + 1) / 2;
    n_two = n_solo = 0;
    const int by = (int)(group / gx), bx = (int)(group - (long long)by * gx);
    for (int t = 0; t < 4; ++t) {
        const int tx = 2 * bx + (t & 1), ty = 2 * by + (t >> 1);
        cls[t] = -1;  // no such tile
        if (group < 0 || tx >= pb_tiles_x(P) || ty >= pb_tiles_y(P)) continue;
        const PbTileEntry& l = table_l[(size_t)ty * pb_tiles_x(P) + tx];
        const PbTileEntry& r = table_r[(size_t)ty * pb_tiles_x(P) + tx];
        cls[t] = pb_eye_class(l.flags, r.flags, l.fix_cnt, r.fix_cnt);
        if (cls[t]) ++n_solo; else ++n_two;
    }
}
__global__ void pb_pair_count_kernel(const PbParams P, const PbTileEntry* __restrict__ table_l, const PbTileEntry* __restrict__ table_r,
                                     const int* __restrict__ unit_of, int units_per_xcd, unsigned n_groups, int unit_side, int unit_side_y,
                                     unsigned* __restrict__ n_wgs) {
    const unsigned B = blockIdx.x * blockDim.x + threadIdx.x;
    if (B >= n_groups) return;
    int cls[4], n_two, n_solo;
    pb_group_classes(P, table_l, table_r, pb_group_of_workgroup(P, B, unit_of, units_per_xcd, unit_side, unit_side_y), cls, n_two, n_solo);
    n_wgs[B] = (unsigned)((n_two + 1) / 2 + (n_solo > 0 ? 1 : 0));
}
__global__ __launch_bounds__(512) void pb_pair_scan_kernel(const unsigned* __restrict__ n_wgs, unsigned n_groups, unsigned* __restrict__ start,
                                                          unsigned* __restrict__ totals) {
    const unsigned x = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    unsigned running = 0;
    for (unsigned base = 0; base * 8u < n_groups; base += 64u) {
        const unsigned B = (base + lane) * 8u + x;
        const unsigned v = B < n_groups ? n_wgs[B] : 0u;
        unsigned incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned up = __shfl_up(incl, d);
            if ((int)lane >= d) incl += up;
        }
        if (B < n_groups) start[B] = running + incl - v;
        running += __shfl(incl, 63);
    }
    if (lane == 0) totals[x] = running;
}
__global__ void pb_skip_fill_kernel(PbTileEntry* __restrict__ ltable, unsigned n_slots) {
    const unsigned v = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (v < n_slots) reinterpret_cast<int*>(ltable + v)[lane] = lane == offsetof(PbTileEntry, flags) / 4 ? PB_TILE_SKIP : 0;
}
__global__ __launch_bounds__(64) void pb_pair_table_kernel(const PbParams P, const PbTileEntry* __restrict__ table_l, const PbTileEntry* __restrict__ table_r,
                                                           PbTileEntry* __restrict__ ltable, const int* __restrict__ unit_of, int units_per_xcd, unsigned n_groups,
                                                           int unit_side, int unit_side_y, const unsigned* __restrict__ start) {
    const unsigned B = blockIdx.x, lane = threadIdx.x;
    if (B >= n_groups) return;
    const long long group = pb_group_of_workgroup(P, B, unit_of, units_per_xcd, unit_side, unit_side_y);
    int cls[4], n_two, n_solo;
    pb_group_classes(P, table_l, table_r, group, cls, n_two, n_solo);
    if (n_two + n_solo == 0) return;
    const int gx = (pb_tiles_x(P) + 1) / 2;
    const int by = (int)(group / gx), bx = (int)(group - (long long)by * gx);
    const int FL = offsetof(PbTileEntry, flags) / 4, TXY = offsetof(PbTileEntry, tile_xy) / 4;
    const unsigned pair_wgs = (unsigned)((n_two + 1) / 2);
    int seen_two = 0, seen_solo = 0;
    for (int t = 0; t < 4; ++t) {
        if (cls[t] < 0) continue;
        const int tx = 2 * bx + (t & 1), ty = 2 * by + (t >> 1);
        const size_t tile = (size_t)ty * pb_tiles_x(P) + tx;
        const int wl = reinterpret_cast<const int*>(table_l + tile)[lane], wr = reinterpret_cast<const int*>(table_r + tile)[lane];
        if (cls[t] == 0) {
            const unsigned wg = start[B] + (unsigned)(seen_two >> 1);
            int* out = reinterpret_cast<int*>(ltable + ((size_t)(wg * 8u + (B & 7u)) * 4u + 2u * (unsigned)(seen_two & 1)));
            int l = wl, r = wr;
            if ((int)lane == FL) { l |= PB_TILE_TWO; r |= PB_TILE_PAIR_R; }
            if ((int)lane == TXY) l = r = (ty << 16) | tx;
            out[lane] = l;
            out[64 + lane] = r;
            ++seen_two;
        } else {
            const unsigned wg = start[B] + pair_wgs;
            int* out = reinterpret_cast<int*>(ltable + ((size_t)(wg * 8u + (B & 7u)) * 4u + (unsigned)seen_solo));
            int w = cls[t] == 2 ? wr : wl;
            if ((int)lane == FL) w = (w & PB_SOLO_KEEP) | PB_TILE_SOLO | (cls[t] == 2 ? PB_TILE_EYE_R : 0);
            if ((int)lane == TXY) w = (ty << 16) | tx;
            out[lane] = w;
            ++seen_solo;
        }
    }
    if (n_two & 1) {  // the last pair workgroup holds one pair: two pads that still meet its barrier
        const unsigned wg = start[B] + pair_wgs - 1u;
        int* out = reinterpret_cast<int*>(ltable + ((size_t)(wg * 8u + (B & 7u)) * 4u + 2u));
        out[lane] = (int)lane == FL ? (PB_TILE_SKIP | PB_TILE_TWO) : 0;
        out[64 + lane] = (int)lane == FL ? (PB_TILE_SKIP | PB_TILE_TWO) : 0;
    }
}
// bil: the opt-in bilinear mode's launch order - a tile served from the exact coordinate table (bil_off >= 0: four clamped gathers per
// pixel) is the slowest class there
__device__ __forceinline__ float pb_tile_cost(const PbTileEntry& e, const bool bil = false) {
    const int f = e.flags;
    if (bil && e.bil_off >= 0) return 2.5f;
    return (f & PB_TILE_BLACK) ? 0.3f
         : (f & PB_TILE_LEAN) ? 0.55f + 0.055f * (float)(e.win_rows * 16 * e.win_n16) / 1024.0f
         : (f & PB_TILE_DIRECT) ? fminf(2.0f, 0.85f + 0.0027f * (float)e.win_cols)
         : (f & PB_TILE_FAILED) ? 1.5f : 1.0f;
}
// Plan creation: the cost of every super-tile for the launch order (fixed point, 1/1024: integer sums are order-independent).
// A wave's measured life (experiments/diag_trace.py): 3 us on a black tile, 4.2 + 0.4 per KiB of window on a window tile,
// 9.5 + 0.03 per source column on a direct-gather tile - here relative to a direct-gather tile of ordinary width.
// forward / n_units: the class counters the budget pass in front of this kernel has left on the device (4 words) travel behind the costs
// (unit_cost[n_units ..]), so that ONE copy brings both to the host (a round trip is 25 us of a 0.5 ms plan).
__global__ void pb_unit_cost_kernel(const PbTileEntry* __restrict__ table, unsigned n_tiles, unsigned tiles_x, unsigned unit_tiles,
                                    unsigned units_x, unsigned* __restrict__ unit_cost, const PbTileEntry* __restrict__ table_r = nullptr,
                                    unsigned unit_tiles_y = 0, int bil = 0, const unsigned* __restrict__ forward = nullptr, unsigned n_units = 0) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (forward && t < 4u) unit_cost[n_units + t] = forward[t];
    if (t >= n_tiles) return;
    float c = pb_tile_cost(table[t], bil != 0);
    if (table_r) c += pb_tile_cost(table_r[t], bil != 0) - 0.3f;  // double-fisheye source: both eyes' work (a one-eye tile costs its live eye's)
    const unsigned ty = t / tiles_x, tx = t - ty * tiles_x;
    atomicAdd(&unit_cost[(ty / (unit_tiles_y ? unit_tiles_y : unit_tiles)) * units_x + tx / unit_tiles], (unsigned)(c * 1024.0f + 0.5f));
}
// The parameter block as the launches read it, stored from the kernel's own arguments: no host copy, no round trip.
__global__ void pb_store_params_kernel(const PbParams P, PbParams* __restrict__ out) {
    const unsigned* in = reinterpret_cast<const unsigned*>(&P);
    unsigned* o = reinterpret_cast<unsigned*>(out);
    for (unsigned k = threadIdx.x; k < sizeof(PbParams) / 4u; k += blockDim.x) o[k] = in[k];
}
// A short host array into device memory from the kernel's own arguments, 256 words per launch: the launch order's unit list (a few hundred
// entries) without the synchronous staged copy a pageable hipMemcpy is (~12 us).
struct PbWordChunk {
    int v[256];
};
__global__ void pb_store_words_kernel(const PbWordChunk c, int* __restrict__ out, int n) {
    const int k = threadIdx.x;
    if (k < n) out[k] = c.v[k];
}
__global__ void pb_save_flags_kernel(const PbTileEntry* __restrict__ table, int32_t* __restrict__ saved, unsigned n_tiles) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tiles) saved[t] = table[t].flags;
}

// Plan creation: the exact-index tables of the hot kernel above.  Blocks [0, 4 * n_fail_tiles) take the failed
// tiles (256 px each; the tile's slot = its position in fail_tiles, recorded in its entry), the remaining blocks
// the fix list.
template <int SRC_KIND>
__global__ __launch_bounds__(PB_BLOCK) void pb_fix_tables_kernel(const PbParams P, PbTileEntry* __restrict__ table,
                                                                 const int32_t* __restrict__ fail_tiles, int n_fail_tiles,
                                                                 const int32_t* __restrict__ fix_px, int n_fix_px,
                                                                 int32_t* __restrict__ idx_tab, int32_t* __restrict__ fix_idx) {
    if ((int)blockIdx.x < 4 * n_fail_tiles) {
        const int s = blockIdx.x >> 2, t = fail_tiles[s];
        const int ty = t / pb_tiles_x(P), tx = t - ty * pb_tiles_x(P);
        const int local = (blockIdx.x & 3) * 256 + threadIdx.x;
        const int i = ty * PB_TILE + (local >> 5), j = tx * PB_TILE + (local & 31);
        idx_tab[(size_t)s * (PB_TILE * PB_TILE) + local] = (i < P.dst.height && j < P.dst.width) ? pb_exact_index<SRC_KIND>(P, i, j) : -1;
        if (local == 0) table[t].aux_off = s;
        return;
    }
    const unsigned item = (blockIdx.x - 4u * n_fail_tiles) * PB_BLOCK + threadIdx.x;
    if (item >= (unsigned)n_fix_px) return;
    const unsigned p = (unsigned)fix_px[item];
    const int i = (int)(p / (unsigned)P.dst.width), j = (int)(p - (unsigned)i * (unsigned)P.dst.width);
    fix_idx[item] = pb_exact_index<SRC_KIND>(P, i, j);
}

// The plan's fix list behind pb_hot_kernel (frames LDS-DMA cannot address, and the int32 index-map output): blocks
// [0, 4 * n_fail_tiles) take the failed tiles (256 px each), the remaining blocks single pixels; the faithful
// results come from the plan's exact-index tables (pb_fix_tables_kernel), nothing is recomputed.
template <int OUT>
__global__ __launch_bounds__(PB_BLOCK) void pb_fix_kernel(const PbParams P, const int32_t* __restrict__ fail_tiles,
                                                          int n_fail_tiles, const int32_t* __restrict__ fix_px,
                                                          int n_fix_px, const int32_t* __restrict__ idx_tab,
                                                          const int32_t* __restrict__ fix_idx, const uint8_t* __restrict__ src,
                                                          uint8_t* __restrict__ dst, int n_frames,
                                                          unsigned long long src_stride, unsigned long long dst_stride,
                                                          int32_t* __restrict__ idx_out) {
    int id;
    size_t p;
    if ((int)blockIdx.x < 4 * n_fail_tiles) {
        const int s = blockIdx.x >> 2, t = fail_tiles[s];
        const int ty = t / pb_tiles_x(P), tx = t - ty * pb_tiles_x(P);
        const int local = (blockIdx.x & 3) * 256 + threadIdx.x;
        const int i = ty * PB_TILE + (local >> 5), j = tx * PB_TILE + (local & 31);
        if (i >= P.dst.height || j >= P.dst.width) return;
        id = idx_tab[(size_t)s * (PB_TILE * PB_TILE) + local];
        p = (size_t)i * P.dst.width + j;
    } else {
        const unsigned item = (blockIdx.x - 4u * n_fail_tiles) * PB_BLOCK + threadIdx.x;
        if (item >= (unsigned)n_fix_px) return;
        id = fix_idx[item];
        p = (size_t)(unsigned)fix_px[item];
    }
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
// The first n in [lo, hi) where a monotone predicate (false ... false true ... true) holds, hi if nowhere: 64 probes per step, one per
// lane, instead of a bisection's one (3e7 candidates: 5 steps of one predicate's latency instead of 25 - the predicate is a chain of a
// float64 square root and an arcsine, and a plan's preparation waits for this kernel: 25-35 us -> ~10).
template <typename PRED>
__device__ __forceinline__ long long pb_first_true_64(long long lo, long long hi, const int lane, PRED pred) {
    while (lo < hi) {
        const long long step = (hi - lo + 63) / 64;        // >= 1; the last lane's probe is hi - 1 or beyond
        const long long p = lo + step * (lane + 1) - 1;
        const bool t = (p < hi) ? pred(p) : true;            // (beyond the interval: counts as true)
        const unsigned long long m = __builtin_amdgcn_ballot_w64(t);
        if (m == 0) return hi;                               // every probe false, hi - 1 among them
        const int first = __builtin_ctzll(m);
        const long long pf = lo + step * (first + 1) - 1;    // the first true probe; the one before it is false
        lo += step * first;
        hi = pf < hi ? pf : hi;                              // (at most step - 1 candidates left)
    }
    return lo;
}
// The destination-validity thresholds with the exact predicate: wave 0 the left / single image, wave 1 the right eye of a double destination.
__global__ __launch_bounds__(128) void pb_threshold_kernel(const PbParams P, long long* __restrict__ out) {
    const int side = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long wc = (P.dst.kind == PB_KIND_DOUBLE) ? P.dst_half_w : P.dst.width;
    const long long nmax = (wc - 1) * (wc - 1) + (long long)(P.dst.height - 1) * (P.dst.height - 1);
    // first n4 where the lens inverse leaves its domain (asin argument > 1); nmax + 1 if never
    const long long n_dom = pb_first_true_64(0, nmax + 1, lane, [&](long long n) {
        bool outside;
        pb_dst_inv_pred(P, n, side != 0, &outside);
        return outside;
    });
    // first n4 in [0, n_dom) where the pixel is invalid (monotone inside the domain)
    const long long lo = pb_first_true_64(0, n_dom, lane, [&](long long n) { return pb_dst_inv_pred(P, n, side != 0, nullptr); });
    if (lane == 0) {
        out[2 * side + 0] = lo;      // invalid  <=>  lo <= n4 < n_dom
        out[2 * side + 1] = n_dom;
    }
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
        e->c[lane][0] = any_bad ? 0.0f : (float)(lane == 0 ? c0 - a0 : c0);
        e->c[lane][1] = any_bad ? 0.0f : (float)(lane == 0 ? c1 - a1 : c1);
    }
    if (lane == 0) {
        e->anchor_r = any_bad ? 0 : (int)a0;
        e->anchor_c = any_bad ? 0 : (int)a1;
        e->flags = any_bad ? PB_TILE_FAILED : PB_TILE_HAS_MODEL;
        e->win_rows = 0;
        e->win_r0 = e->win_c0 = e->win_cols = e->win_n16 = e->win_a0 = 0;
        e->fix_off = e->fix_cnt = 0;
        e->aux_off = 0;
        e->bil_off = -1;
    }
}

// One wave per tile: evaluates the model for every pixel of the tile, finds the bounding box of its
// source samples and decides whether the tile is LEAN; LEAN tiles are re-anchored at the window origin.
template <int SRC_KIND>
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_window_kernel(const PbParams P, PbTileEntry* __restrict__ table) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tx, ty;
    if (!pb_tile_of_wave(P, wave, tx, ty)) return;
    PbTileEntry* e = table + ((size_t)ty * pb_tiles_x(P) + tx);
    if (e->flags & PB_TILE_FAILED) return;
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    const int y = lane & 31, xh = (lane >> 5) * 16;
    const int i = Y0 + y;
    const int h = P.src.height, w = P.src.width;
    PbRowModel R;
    pb_model_row(P, e, X0, Y0, y, xh, R);
    int rmin = 0x7fffffff, rmax = -0x7fffffff, cmin = 0x7fffffff, cmax = -0x7fffffff;   // raw (unwrapped)
    int wrmin = 0x7fffffff, wrmax = -1, wcmin = 0x7fffffff, wcmax = -1;               // as the generic path uses them
    int not_plain = 0;  // pixels outside the image, black or wrapping
    int n_invalid = 0;  // invalid destination pixels (inside the image)
    for (int k = 0; k < 16; ++k) {
        const int j = X0 + xh + k;
        if (i >= P.dst.height || j >= P.dst.width) { ++not_plain; continue; }
        int r, c;
        pb_f2 f;
        pb_model_px_raw(R, xh, k, r, c, f);
        if (pb_row_px_invalid(R, k)) { ++n_invalid; continue; }
        rmin = min(rmin, r); rmax = max(rmax, r);
        cmin = min(cmin, c); cmax = max(cmax, c);
        const int v = pb_model_px_rc<SRC_KIND>(P, R, xh, k);
        if (v < 0) { ++not_plain; continue; }
        const int wr = v >> 16, wc = v & 0xFFFF;
        if (wr != r || wc != c) ++not_plain;
        wrmin = min(wrmin, wr); wrmax = max(wrmax, wr);
        wcmin = min(wcmin, wc); wcmax = max(wcmax, wc);
    }
    rmin = pb_wave_min(rmin); rmax = pb_wave_max(rmax); cmin = pb_wave_min(cmin); cmax = pb_wave_max(cmax);
    wrmin = pb_wave_min(wrmin); wrmax = pb_wave_max(wrmax); wcmin = pb_wave_min(wcmin); wcmax = pb_wave_max(wcmax);
    not_plain = pb_wave_sum(not_plain);
    n_invalid = pb_wave_sum(n_invalid);
    if (lane != 0) return;
    if (wrmax < 0) {  // no pixel of the tile samples the source
        e->win_r0 = e->win_rows = e->win_c0 = e->win_cols = 0;
        e->flags |= PB_TILE_BLACK;
        return;
    }
    // generic window: bounding box of the wrapped samples
    e->win_r0 = wrmax < 0 ? 0 : wrmin;
    e->win_rows = wrmax < 0 ? 0 : wrmax - wrmin + 1;
    e->win_c0 = wrmax < 0 ? 0 : wcmin;
    e->win_cols = wrmax < 0 ? 0 : wcmax - wcmin + 1;
    // LEAN? one texel of margin on every side absorbs the last-bit effect of re-anchoring
    const unsigned rowbytes = 3u * (unsigned)w;
    const unsigned safe_len = (rowbytes * (unsigned)h) & ~15u;
    if (not_plain != 0 || w >= 32768 || h >= 32768) return;
    // invalid destination pixels only (the rest plain): the MASKED class, single sources only (the two-eye paths do not know it)
    const bool masked = n_invalid != 0;
    if (masked && SRC_KIND != PB_KIND_PANO && SRC_KIND != PB_KIND_CAMERA) return;
    int lr0 = rmin - 1, lc0 = cmin - 1, rows = rmax - rmin + 3, cols = cmax - cmin + 3;
    if (masked) {
        // a ring tile's valid pixels reach the LAST rows / columns of the source (a fisheye's rim is the panorama's pole row): the
        // margin texel is clipped at the frame's edge.  Safe: certification checks for every sampled pixel, in both evaluation
        // orders, that its window offsets lie inside [0, rows) x [0, cols) - no gather can leave the frame - and MASKED tiles never
        // take the LDS-window or the unguarded bilinear paths, the margin's other customers
        const int r1 = min(rmax + 1, h - 1), c1 = min(cmax + 1, w - 1);
        lr0 = max(lr0, 0);
        lc0 = max(lc0, 0);
        rows = r1 - lr0 + 1;
        cols = c1 - lc0 + 1;
    }
    // (an eye's margin texel may lie in the other eye's half: inside the frame, never sampled)
    if (lr0 < 0 || lc0 < 0 || lr0 + rows > h || lc0 + cols > w) return;
    // the last sample's 4-byte read must stay inside the frame
    if ((unsigned)(lr0 + rows - 1) * rowbytes + 3u * (unsigned)(lc0 + cols - 1) + 4u > rowbytes * (unsigned)h) return;
    const unsigned a0 = (3u * (unsigned)lc0) & 15u;
    const unsigned n16 = (a0 + 3u * (unsigned)cols + 1u + 15u) >> 4;
    const unsigned last_chunk_end = (((unsigned)(lr0 + rows - 1) * rowbytes + 3u * (unsigned)lc0) & ~15u) + 16u * n16;
    const bool stageable = (rowbytes & 15u) == 0 && n16 <= 64u && (unsigned)rows * 16u * n16 <= PB_WINLDS_BYTES &&
                           ((unsigned)rows + (64u / (n16 > 64u ? 64u : n16)) - 1u) / (64u / (n16 > 64u ? 64u : n16)) <= PB_LEAN_MAX_PASSES &&
                           last_chunk_end <= safe_len;
    e->c[0][0] += (float)(e->anchor_r - lr0);
    e->c[0][1] += (float)(e->anchor_c - lc0);
    e->anchor_r = lr0;
    e->anchor_c = lc0;
    e->win_r0 = lr0;
    e->win_c0 = lc0;
    e->win_rows = rows;
    e->win_cols = cols;
    e->win_n16 = (int)n16;
    e->win_a0 = (int)a0;
    e->flags |= masked ? (PB_TILE_DIRECT | PB_TILE_MASKED) : (stageable ? PB_TILE_LEAN : PB_TILE_DIRECT);
}

// Plan creation, after certification: how many tiles ended in each class.  counters: [4] LEAN, [5] BLACK, [6] DIRECT.
__global__ void pb_count_flags_kernel(const PbTileEntry* __restrict__ table, unsigned n_tiles, unsigned* __restrict__ counters) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const int f = table[t].flags;
    if (f & PB_TILE_LEAN) atomicAdd(&counters[4], 1u);
    if (f & PB_TILE_BLACK) atomicAdd(&counters[5], 1u);
    if (f & PB_TILE_DIRECT) atomicAdd(&counters[6], 1u);
}

// One wave per tile: compares the hot path's index with the faithful one for every pixel of the
// tile; differing pixels go to the fix list, tiles with more than PB_TILE_FAIL_LIMIT of them (or
// without a model) are marked failed.  A LEAN tile whose pixels do not all satisfy the lean
// invariants (non-negative offsets inside the window) loses the flag.  counters: [0] fix pixels,
// [1] failed tiles, [2] pixels differing in total (statistics).
// Plan creation, unrotated panorama destinations: an output pixel's longitude depends on its column alone (projection.py:502-512), so the
// correctly rounded sine / cosine the source stage takes of it (np.exp(lon * 1j), projection.py:252) is evaluated once per COLUMN -
// col_sc[2 j] = cos, col_sc[2 j + 1] = sin, the very bits a per-pixel evaluation returns - and certification looks them up (round 4:
// c5's two certification passes were 2.4 of its 3.7 ms of plan preparation, most of it this one function).
__global__ void pb_col_sincos_kernel(const PbParams P, double* __restrict__ col_sc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.dst.width) return;
    const PbCoord c = pb_dst_coord(P, 0, j);
    pb_expi_np(c.lon, &col_sc[2 * j + 1], &col_sc[2 * j]);
}

#ifndef PB_CERTIFY_WPE  // (the compiler's free choice: 87-116 VGPRs, four or five waves per SIMD, now that the build keeps the math kernels'
#define PB_CERTIFY_WPE 1  // constants out of long-lived registers - build.py; with them hoisted it was 212, and a forced 168 + scratch won 9-20 %)
#endif
#ifndef PB_CERTIFY_UNROLL
#define PB_CERTIFY_UNROLL 0
#endif
#ifndef PB_CERT_ABL  // timing experiments only (experiments/r4/build_f64.sh): 1 = no faithful chain, 2 = no column-first evaluation, 4 = no coarse measure
#define PB_CERT_ABL 0
#endif
template <int SRC_KIND, int ROT>
__global__ __launch_bounds__(64 * PB_TILE_WAVES, PB_CERTIFY_WPE) void pb_certify_kernel(const PbParams P, PbTileEntry* __restrict__ table,
                                                                         int32_t* __restrict__ fail_tiles,
                                                                         int32_t* __restrict__ fix_px, unsigned fix_capacity,
                                                                         unsigned* __restrict__ counters, const double* __restrict__ col_sc = nullptr) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (wave-uniform: the tile entry is read with scalar loads)
    int tx, ty;
    if (!pb_tile_of_wave(P, wave, tx, ty)) return;
    const int tile = ty * pb_tiles_x(P) + tx;
    PbTileEntry* e = table + tile;
    // the tile's entry as the kernel found it, in scalar registers: the model is evaluated from THIS copy.  (Read through `e` - memory
    // this kernel also writes - the 50 coefficients were fetched again for every one of a lane's 16 pixels: 79 scalar loads per wave,
    // each a dependent round trip; 74 % of the kernel's wave cycles were waits: experiments/r4/pmc_plan.sh.)
    PbTileEntry L;
    pb_load_entry(e, L);
    const PbTileEntry* __restrict__ ec = &L;
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    const int y = lane & 31, xh = (lane >> 5) * 16;
    const int i = Y0 + y;
    bool failed = (ec->flags & PB_TILE_FAILED) != 0;
    const bool lean = (ec->flags & (PB_TILE_LEAN | PB_TILE_DIRECT)) != 0;
    unsigned diff = 0;  // bit k: pixel xh + k of this lane's row differs
    if (!failed) {
        PbRowModel R;
        pb_model_row(P, ec, X0, Y0, y, xh, R);
        bool lean_ok = true;
        double coarse = 0.0;  // largest |model - faithful| pre-truncation coordinate over this lane's sampled pixels, px
        double coarse3 = 0.0;  // ... of the model's terms of total degree <= 3 alone (PB_TILE_TD3)
        pb_f2 a3[4];
        pb_collapse_row_td3(ec, y, a3);
#if PB_CERTIFY_UNROLL
        PB_UNROLL(PB_CERTIFY_UNROLL)
#endif
        for (int k = 0; k < 16; ++k) {
            const int j = X0 + xh + k;
            if (i < P.dst.height && j < P.dst.width) {
                const int fast = pb_model_px<SRC_KIND>(P, R, xh, k);
                PbCoord cc = {0.0, 0.0, false};
                if (!(PB_CERT_ABL & 1)) {
                    cc = pb_rotate_all<ROT>(P, pb_dst_coord(P, i, j));
                }
                // the faithful index and the faithful pre-truncation coordinate from ONE evaluation of the longitude's sine / cosine
                int exact;
                double f0 = 0.0, f1 = 0.0;
                if (PB_CERT_ABL & 1) {
                    exact = fast;
                } else if (SRC_KIND == PB_KIND_PANO) {
                    exact = pb_src_pano_index(P, cc);
                    if (exact >= 0) pb_src_pretrunc<PB_KIND_PANO>(P, cc, f0, f1);
                } else {
                    double sl, cl;
                    if (col_sc) {  // (unrotated panorama destination: the column's table entry IS pb_sincos_cr(cc.lon))
                        cl = col_sc[2 * j];
                        sl = col_sc[2 * j + 1];
                    } else {
                        pb_expi_np(cc.lon, &sl, &cl);
                    }
                    exact = pb_src_index_sc<SRC_KIND>(P, cc, sl, cl);
                    pb_src_pretrunc_sc<SRC_KIND>(P, cc, sl, cl, f0, f1);
                }
                diff |= (unsigned)(fast != exact) << k;
                if (exact >= 0 && !(PB_CERT_ABL & 4)) {  // (what the bilinear mode needs to know: PB_TILE_COARSE)
                    int mr, mc;
                    pb_f2 mf;
                    pb_model_px_raw(R, xh, k, mr, mc, mf);
                    double d0 = fabs(((double)R.anchor_r + (double)mf.x) - f0), d1 = fabs(((double)R.anchor_c + (double)mf.y) - f1);
                    if (SRC_KIND == PB_KIND_PANO) {  // the model runs on across the seam (columns) and the pole row
                        d0 = fmin(d0, fabs(d0 - (double)P.src.height));
                        d1 = fmin(d1, fabs(d1 - (double)P.src.width));
                    }
                    const double dm = fmax(d0, d1);
                    coarse = (dm == dm) ? fmax(coarse, dm) : 1.0;
                    const pb_f2 tf = pb_eval_row_td3(a3, pb_tile_coord(xh + k));
                    double e0 = fabs(((double)R.anchor_r + (double)tf.x) - f0), e1 = fabs(((double)R.anchor_c + (double)tf.y) - f1);
                    if (SRC_KIND == PB_KIND_PANO) {
                        e0 = fmin(e0, fabs(e0 - (double)P.src.height));
                        e1 = fmin(e1, fabs(e1 - (double)P.src.width));
                    }
                    const double em = fmax(e0, e1);
                    coarse3 = (em == em) ? fmax(coarse3, em) : 1.0;
                }
                if (lean && !pb_row_px_invalid(R, k)) {  // (a MASKED tile's invalid pixels are never sampled: the hot path masks them)
                    const pb_f2 f = pb_eval_row(R.a, pb_tile_coord(xh + k));
                    lean_ok = lean_ok && f.x >= 0.0f && f.y >= 0.0f && (int)f.x < ec->win_rows && (int)f.y < ec->win_cols;
                    // plain tiles may also be evaluated column-first by the hot kernel (a DIRECT tile whose gathers run
                    // down the columns collapses the model along u once per lane): same polynomial, another rounding
                    // order - certified as well, a pixel either order gets wrong goes on the fix list
                    pb_f2 bcol[5];
                    if (PB_CERT_ABL & 2) continue;
                    pb_collapse_col(ec, xh + k, bcol);
                    const pb_f2 g = pb_eval_row(bcol, pb_tile_coord(y));
                    lean_ok = lean_ok && g.x >= 0.0f && g.y >= 0.0f && (int)g.x < ec->win_rows && (int)g.y < ec->win_cols;
                    const int fast_col = (ec->anchor_r + (int)g.x) * P.src.width + ec->anchor_c + (int)g.y;
                    diff |= (unsigned)(fast_col != exact) << k;
                }
            }
        }
        if (lean && __builtin_amdgcn_ballot_w64(!lean_ok) != 0) {
            if (lane == 0) e->flags &= ~(PB_TILE_LEAN | PB_TILE_DIRECT);
        }
        if (__builtin_amdgcn_ballot_w64(coarse > PB_COARSE_PX) != 0 && lane == 0) e->flags |= PB_TILE_COARSE;
        else if (lean && __builtin_amdgcn_ballot_w64(coarse3 > PB_COARSE_PX) == 0 && lane == 0) e->flags |= PB_TILE_TD3;
        unsigned total = (unsigned)pb_wave_sum((int)__popc(diff));
        // NO per-tile bookkeeping atomics here (round 4): every wave used to add to the same three or four words - LEAN / DIRECT / BLACK
        // tile counts, pixel totals - and 32 768 waves x 3.5 device-scope atomics on one cache line WERE the kernel: 1.14 ms for c5 whatever
        // was computed in between (experiments/r4/plan_kernels.sh: the float64 chain, the second evaluation order and the coordinate
        // measure removed, one at a time: +-0).  The class counts come from pb_count_flags_kernel afterwards (one thread per tile, the
        // compiler's wave-level reduction in front of each atomic); the pixel total is added where it is not zero.
        if (lane == 0 && total) atomicAdd(&counters[2], total);
        if (total > PB_TILE_FAIL_LIMIT) failed = true;
        if (!failed && total) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&counters[0], total);
            base = __shfl(base, 0);
            if (base + total > fix_capacity) {
                failed = true;
            } else {
                unsigned mine = __popc(diff), pre = mine;  // inclusive prefix of per-lane counts
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned t = __shfl_up(pre, o);
                    if (lane >= o) pre += t;
                }
                unsigned pos = base + pre - mine;
                for (int k = 0; k < 16; ++k)
                    if (diff & (1u << k)) fix_px[pos++] = i * P.dst.width + (X0 + xh + k);
                if (lane == 0) {
                    e->fix_off = (int)base;
                    e->fix_cnt = (int)total;
                }
            }
        }
    }
    if (failed && lane == 0) {
        e->flags = PB_TILE_FAILED;
        fail_tiles[atomicAdd(&counters[1], 1u)] = tile;
    }
}

