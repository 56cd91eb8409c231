// pb_kernels_x.hpp - the hot kernel with wide stores (pb_hot_x_kernel).
//
// One wave per 32x32 tile, 4 tiles = a 64x64 block per workgroup, like pb_hot_win_kernel - but a wave that
// stores its own tile writes 96-byte row pieces: one and a half 64-byte memory lines per row, the other half
// written later by another wave.  Streaming (non-temporal) stores of such half lines cost about a quarter
// more write transactions and, next to the source stream, 10-15 % of the whole kernel
// (experiments/exp_window.hip).  Here every wave parks its packed pixels in its own - by then dead - LDS
// window and checks in at an LDS counter shared with its horizontal neighbour; whichever of the two arrives
// SECOND stores both tiles: 192 contiguous bytes per row, 4 rows per store instruction, only whole 64-byte
// lines.  Nobody waits: there is no barrier on the single-frame path (one per frame for batches, because the
// parked tiles live in the next frame's windows).
// The per-tile stages (window loads, model math, gather) are shared with nothing else: they are the
// pipeline stages of this kernel, written as functions of a tile descriptor held in scalar registers.
#pragma once
#include "pb_kernels_tile.hpp"

struct PbCuCtx {
    unsigned rowbytes, frame_bytes, safe_len;
    int lane, xg, yb, W, H;
    float u[4];
};

struct PbDesc {  // the scalar part of a work entry
    int flags, anchor_r, anchor_c, win_rows, win_r0, win_c0, win_cols, win_n16, win_a0, fix_off, fix_cnt;
    int tile_xy;  // (ty << 16) | tx, set by the kernel
};

__device__ __forceinline__ PbDesc pb_load_desc(const PbTileEntry* __restrict__ e) {
    PbDesc d;
    d.flags = e->flags; d.anchor_r = e->anchor_r; d.anchor_c = e->anchor_c; d.win_rows = e->win_rows;
    d.win_r0 = e->win_r0; d.win_c0 = e->win_c0; d.win_cols = e->win_cols; d.win_n16 = e->win_n16; d.win_a0 = e->win_a0;
    d.fix_off = e->fix_off; d.fix_cnt = e->fix_cnt; d.tile_xy = 0;
    return d;
}

#define PB_CU_PLAIN(flags) ((flags) & (PB_TILE_LEAN | PB_TILE_DIRECT))
#define PB_CU_GENERIC(flags) (!((flags) & (PB_TILE_LEAN | PB_TILE_DIRECT | PB_TILE_BLACK | PB_TILE_FAILED)))

// window geometry of a generic tile (same rule as pb_win_tile)
__device__ __forceinline__ void pb_cu_generic_window(const PbDesc& D, const PbCuCtx& C, int& nrows, int& n16, unsigned& gbase) {
    nrows = D.win_rows;
    gbase = (unsigned)D.win_r0 * C.rowbytes + 3u * (unsigned)D.win_c0;
    n16 = 1;
    if (nrows > 0) {
        n16 = (3 * D.win_cols + 15 + 1 + 15) >> 4;
        if (n16 > 64) n16 = 64;
        const int cap = PB_WINLDS_BYTES / (16 * n16);
        if (nrows > cap) nrows = cap;
    }
}

// stage A1: issue the LDS-DMA loads of the tile's source window (nothing for BLACK / DIRECT tiles)
__device__ __forceinline__ void pb_cu_issue(const PbDesc& D, const PbCuCtx& C, const uint8_t* __restrict__ s, unsigned* win) {
    if (D.flags & PB_TILE_LEAN) {
        const unsigned gbase = (unsigned)D.anchor_r * C.rowbytes + 3u * (unsigned)D.anchor_c;
        pb_issue_window_loads(s, win, C.lane, gbase, C.rowbytes, D.win_rows, D.win_n16, C.safe_len);
    } else if (PB_CU_GENERIC(D.flags)) {
        int nrows, n16;
        unsigned gbase;
        pb_cu_generic_window(D, C, nrows, n16, gbase);
        if (nrows > 0) pb_issue_window_loads(s, win, C.lane, gbase, C.rowbytes, nrows, n16, C.safe_len);
    }
}

// DIRECT tiles: 16 unaligned dword gathers per lane straight from the frame
__device__ __forceinline__ void pb_cu_direct_loads(const unsigned q[16], const uint8_t* __restrict__ s, unsigned ad[16]) {
#pragma unroll
    for (int n = 0; n < 16; ++n) __builtin_memcpy(&ad[n], s + q[n], 4);
}

// stage A2: the tile's model math -> per-pixel addresses q[jr * 4 + k]
//   LEAN: byte address in the LDS window; DIRECT: byte offset in the frame; generic: (row << 16 | col) or -1
template <int SRC_KIND>
__device__ __forceinline__ void pb_cu_math(const PbParams& P, const PbDesc& D, const PbCuCtx& C, const PbTileEntry* __restrict__ e,
                                           unsigned q[16]) {
    if (PB_CU_PLAIN(D.flags)) {
        const bool lean = (D.flags & PB_TILE_LEAN) != 0;
        const unsigned pitch = lean ? 16u * (unsigned)D.win_n16 : C.rowbytes;
        const unsigned off = lean ? (unsigned)D.win_a0 : (unsigned)D.anchor_r * C.rowbytes + 3u * (unsigned)D.anchor_c;
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            pb_f2 a[5];
            pb_collapse_row(e, C.yb + 8 * jr, a);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const pb_f2 fv = pb_eval_row(a, C.u[k]);
                const unsigned dr = (unsigned)(int)fv.x, dc = (unsigned)(int)fv.y;  // >= 0: truncation == floor
                // LEAN: dr * pitch < 12288; DIRECT: the product can exceed 24 bits
                q[jr * 4 + k] = (lean ? __umul24(dr, pitch) : dr * pitch) + (__umul24(dc, 3u) + off);
            }
        }
    } else if (PB_CU_GENERIC(D.flags)) {
        const int X0 = (D.tile_xy & 0xFFFF) * PB_TILE, Y0 = (D.tile_xy >> 16) * PB_TILE;
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            PbRowModel R;
            pb_model_row(P, e, X0, Y0, C.yb + 8 * jr, 4 * C.xg, R);
#pragma unroll
            for (int k = 0; k < 4; ++k) q[jr * 4 + k] = (unsigned)pb_model_px_rc<SRC_KIND>(P, R, 4 * C.xg, k);
        }
    }
}

// stage B1: the tile's 16 pixels per lane -> a[] (low 3 bytes valid).  All loads of the tile have landed.
__device__ __forceinline__ void pb_cu_gather(const PbDesc& D, const PbCuCtx& C, const unsigned q[16], const unsigned ad[16],
                                             const unsigned* win, const uint8_t* __restrict__ s, unsigned a[16]) {
    if (D.flags & PB_TILE_LEAN) {
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const unsigned l = q[n];
            a[n] = __builtin_amdgcn_alignbyte(win[(l >> 2) + 1], win[l >> 2], l);
        }
    } else if (D.flags & PB_TILE_DIRECT) {
#pragma unroll
        for (int n = 0; n < 16; ++n) a[n] = ad[n];
    } else if (PB_CU_GENERIC(D.flags)) {
        int nrows, n16;
        unsigned gbase;
        pb_cu_generic_window(D, C, nrows, n16, gbase);
        const unsigned a0 = gbase & 15u, pitch = 16u * (unsigned)n16, rb16 = C.rowbytes & 15u;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int v = (int)q[n];
            unsigned px = 0;
            if (v >= 0) {
                const unsigned r = (unsigned)v >> 16, c = (unsigned)v & 0xFFFFu;
                const unsigned row = r - (unsigned)D.win_r0;
                const unsigned g = r * C.rowbytes + 3u * c;
                const unsigned off = 3u * (c - (unsigned)D.win_c0) + ((a0 + row * rb16) & 15u);
                if (row < (unsigned)nrows && off + 4u <= pitch && g + 4u <= C.safe_len) {
                    const unsigned l = row * pitch + off;
                    px = __builtin_amdgcn_alignbyte(win[(l >> 2) + 1], win[l >> 2], l) & 0xFFFFFFu;
                } else if (g + 4u <= C.frame_bytes) {
                    unsigned t;
                    __builtin_memcpy(&t, s + g, 4);
                    px = t & 0xFFFFFFu;
                } else {
                    px = (unsigned)s[g] | ((unsigned)s[g + 1] << 8) | ((unsigned)s[g + 2] << 16);
                }
            }
            a[n] = px;
        }
    } else {
#pragma unroll
        for (int n = 0; n < 16; ++n) a[n] = 0u;  // BLACK
    }
}

// park 4 packed pixels (12 bytes) per row group in the wave's LDS area: tile row y, 4-px group g at dword (y * 8 + g) * 3
__device__ __forceinline__ void pb_x_park(const PbCuCtx& C, const unsigned a[16], unsigned* park) {
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        const pb_u32x3 o = pb_pack_px4(a[jr * 4 + 0], a[jr * 4 + 1], a[jr * 4 + 2], a[jr * 4 + 3]);
        unsigned* p = park + ((C.yb + 8 * jr) * 8 + C.xg) * 3;
        p[0] = o.x;
        p[1] = o.y;
        p[2] = o.z;
    }
}

// FUSED (single frame per launch only): the wave patches its tile's fix pixels (faithful chain) into the
// parked tile before the barrier, and a handful of failed tiles ride along as leading blocks; otherwise
// pb_fix_kernel follows.  (With a frame loop the per-pixel addresses stay live across the faithful chain and
// the kernel drops to two waves per SIMD; a batch amortises the extra launch anyway.)
template <int SRC_KIND, bool FUSED>
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_hot_x_kernel(const PbParams P, const PbTileEntry* __restrict__ table,
                                                                       const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                       int n_frames, unsigned long long src_stride,
                                                                       unsigned long long dst_stride, unsigned fail_blocks,
                                                                       const int32_t* __restrict__ fail_tiles,
                                                                       const int32_t* __restrict__ fix_px) {
    __shared__ __attribute__((aligned(16))) unsigned win_all[PB_TILE_WAVES][PB_WINLDS_BYTES / 4 + 4];
    __shared__ int tile_live[PB_TILE_WAVES];
    __shared__ int pair_flag[PB_TILE_WAVES];
    if (FUSED && blockIdx.x < fail_blocks) {
        // leading blocks: the plan's failed tiles, 256 pixels per block, faithful chain
        const unsigned b = blockIdx.x;
        const int t = fail_tiles[b >> 2];
        const int fty = t / pb_tiles_x(P), ftx = t - fty * pb_tiles_x(P);
        const int local = (int)(b & 3u) * 256 + (int)threadIdx.x;
        const int i = fty * PB_TILE + (local >> 5), j = ftx * PB_TILE + (local & 31);
        if (i >= P.dst.height || j >= P.dst.width) return;
        const int id = pb_exact_index<SRC_KIND>(P, i, j);
        const size_t p = (size_t)i * P.dst.width + j;
        const unsigned v = pb_load_px(src, id);
        uint8_t* o = dst + 3 * p;
        o[0] = (uint8_t)(v & 0xFF);
        o[1] = (uint8_t)((v >> 8) & 0xFF);
        o[2] = (uint8_t)((v >> 16) & 0xFF);
        return;
    }
    if (threadIdx.x < PB_TILE_WAVES) pair_flag[threadIdx.x] = 0;
    __syncthreads();  // the waves have just started; no other workgroup barrier on the single-frame path
    PbCuCtx C;
    C.lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned* win = win_all[wave];
    C.rowbytes = 3u * (unsigned)P.src.width;
    C.frame_bytes = C.rowbytes * (unsigned)P.src.height;  // < 2^31 (host check)
    C.safe_len = C.frame_bytes & ~15u;
    C.xg = C.lane & 7;
    C.yb = C.lane >> 3;
    C.W = P.dst.width;
    C.H = P.dst.height;
#pragma unroll
    for (int k = 0; k < 4; ++k) C.u[k] = pb_tile_coord(4 * C.xg + k);
    int tx, ty;
    const bool exists = pb_tile_of_wave(P, wave, tx, ty, blockIdx.x - (FUSED ? fail_blocks : 0u));
    // first tile of this wave's horizontal pair
    const int bx2 = tx & ~1;
    const PbTileEntry* __restrict__ e = table + (exists ? (size_t)ty * pb_tiles_x(P) + tx : 0);
    PbDesc D = pb_load_desc(e);
    D.tile_xy = (ty << 16) | tx;
    const bool live = exists && !(D.flags & PB_TILE_FAILED);
    unsigned q[16], ad[16], a[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) q[n] = ad[n] = 0u;
    if (live) {
        pb_cu_issue(D, C, src, win);
        pb_cu_math<SRC_KIND>(P, D, C, e, q);
        if (D.flags & PB_TILE_DIRECT) pb_cu_direct_loads(q, src, ad);
    }
    if (C.lane == 0) tile_live[wave] = live ? 1 : 0;  // read by the partner only after it saw this wave check in
    const int g = C.lane & 15, rs = C.lane >> 4;
    const int frames = FUSED ? 1 : n_frames;
    for (int f = 0; f < frames; ++f) {
        const uint8_t* s = src + (unsigned long long)f * src_stride;
        uint8_t* d = dst + (unsigned long long)f * dst_stride;
        // the per-pixel addresses are loop-invariant; keep the compiler from hoisting everything derived from
        // them out of the frame loop (hundreds of live registers for nothing)
#pragma unroll
        for (int n = 0; n < 16; ++n) asm volatile("" : "+v"(q[n]));
        if (live) {
            if (f > 0) {
                pb_cu_issue(D, C, s, win);
                if (D.flags & PB_TILE_DIRECT) pb_cu_direct_loads(q, s, ad);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the window / the gathers have landed
            pb_wave_sync();
            pb_cu_gather(D, C, q, ad, win, s, a);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // every lane has read its samples
            pb_wave_sync();
            pb_x_park(C, a, win);
            if (FUSED && D.fix_cnt > 0) {
                // this tile's fix pixels: the faithful sample replaces the parked one
                pb_wave_sync();
                if (C.lane < D.fix_cnt) {
                    unsigned p = (unsigned)fix_px[D.fix_off + C.lane];
                    asm volatile("" : "+v"(p));  // keep the (rare, register-hungry) faithful chain inside the frame loop
                    const int i = (int)(p / (unsigned)C.W), j = (int)(p - (unsigned)i * (unsigned)C.W);
                    const unsigned v = pb_load_px(s, pb_exact_index<SRC_KIND>(P, i, j));
                    const int yl = i - ty * PB_TILE, xl = j - tx * PB_TILE;
                    uint8_t* pb = reinterpret_cast<uint8_t*>(win) + (yl * 8 + (xl >> 2)) * 12 + (xl & 3) * 3;
                    pb[0] = (uint8_t)(v & 0xFF);
                    pb[1] = (uint8_t)((v >> 8) & 0xFF);
                    pb[2] = (uint8_t)((v >> 16) & 0xFF);
                }
            }
        }
#ifdef PB_X_OWN
        // (experiment) every wave stores its own parked tile: 96-byte row pieces, no exchange
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (live) {
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                const int yl = C.yb + 8 * jr;
                const int y = ty * PB_TILE + yl, x = tx * PB_TILE + 4 * C.xg;
                if (y >= C.H || x >= C.W) continue;
                const unsigned* p = win + (yl * 8 + C.xg) * 3;
                const unsigned o0 = p[0], o1 = p[1], o2 = p[2];
                const unsigned long long off = 3ull * ((unsigned long long)y * C.W + x);
                if (x + 3 < C.W && (((uintptr_t)d + off) & 3u) == 0) {
                    const pb_u32x3 o = {o0, o1, o2};
                    __builtin_nontemporal_store(o, reinterpret_cast<pb_u32x3*>(d + off));
                } else {
                    const unsigned w3[3] = {o0, o1, o2};
#pragma unroll
                    for (int k = 0; k < 12; ++k)
                        if (x + k / 3 < C.W) d[off + k] = (uint8_t)(w3[k >> 2] >> (8 * (k & 3)));
                }
            }
        }
#else
        // the pair (tiles 2m, 2m + 1 of one tile row) meets: each wave raises its flag and waits for its
        // partner's (both waves of a workgroup are resident, the partner always arrives), then stores half
        // of the pair's rows, 64 pixels = 192 contiguous bytes per row
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the parked tile is in LDS before the flag moves
        if (C.lane == 0) __hip_atomic_store(&pair_flag[wave], f + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&pair_flag[wave ^ 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < f + 1) __builtin_amdgcn_s_sleep(1);
        // lane -> row rs + 4 * s4 (+ 16 for the right-hand wave) of the tile row, 4-px group g of the 64-pixel pair
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int yl = 16 * (wave & 1) + rs + 4 * s4;
            const int from = (wave & ~1) + (g >> 3);
            const int y = ty * PB_TILE + yl, x = bx2 * PB_TILE + 4 * g;
            if (!tile_live[from] || y >= C.H || x >= C.W) continue;
            const unsigned* p = win_all[from] + (yl * 8 + (g & 7)) * 3;
            const unsigned o0 = p[0], o1 = p[1], o2 = p[2];
            const unsigned long long off = 3ull * ((unsigned long long)y * C.W + x);
            if (x + 3 < C.W && (((uintptr_t)d + off) & 3u) == 0) {
                const pb_u32x3 o = {o0, o1, o2};
                __builtin_nontemporal_store(o, reinterpret_cast<pb_u32x3*>(d + off));  // write-once output: keep it out of the source's cache space
            } else {
                const unsigned w3[3] = {o0, o1, o2};
#pragma unroll
                for (int k = 0; k < 12; ++k)
                    if (x + k / 3 < C.W) d[off + k] = (uint8_t)(w3[k >> 2] >> (8 * (k & 3)));
            }
        }
#endif
        if (f + 1 < frames) __syncthreads();  // the parked tiles are overwritten by the next frame's windows
    }
}
