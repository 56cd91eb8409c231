// pb_tile.hpp - per-tile polynomial models of the source-coordinate field.
//
// The remap's index math (inverse projection -> rotations -> forward projection) is
// a smooth map from output pixel (i, j) to a pre-truncation source coordinate
// (row-like, column-like) almost everywhere.  A PLAN therefore carries, for every
// 32x32 tile of the output, a degree-4 x degree-4 polynomial per coordinate
// (PbTileEntry, 256 bytes), built ON THE DEVICE at plan creation from 25 evaluations
// of the faithful float64 chain (pb_stages.hpp) at the tile's Chebyshev-Lobatto nodes.
// The hot kernel evaluates the model per pixel in float32 relative to an integer
// anchor (pb_model_row / pb_model_px below): 8 FMAs + 2 floors per pixel, no
// transcendental, no float64.
//
// Exactness is by construction, not by tolerance: plan creation runs the SAME device
// function (bit-reproducible: no contraction, explicit fmaf) next to the faithful
// chain for EVERY output pixel and records each pixel whose truncated index differs in
// the plan's fix list (tiles with many such pixels - a seam, a pole, the fisheye
// centre, a steep lens edge - are listed whole).  For exactly those pixels the plan
// stores the faithful result (source indices; for double sources taps and factors) and
// the hot kernel looks it up.  The union is bit-identical to the faithful path for every
// pixel, and no frame runs the float64 chain.
#pragma once
#include "pb_stages.hpp"

#define PB_TILE 32
#define PB_TILE_PITCH 33       // LDS row pitch of the index tile (ints): conflict-free column writes
#define PB_TILE_FAIL_LIMIT 48  // more differing pixels than this: the whole tile goes to the fix kernel

#define PB_TILE_HAS_MODEL 1
#define PB_TILE_FAILED 2
// LEAN: every pixel of the tile is a valid destination pixel whose sample lies inside the tile's source
// window, the window fits the LDS budget, and the model is anchored at the window origin - so the hot
// kernel needs no validity, bounds, wrap or fallback code for this tile (decided by the plan builder).
#define PB_TILE_LEAN 4
#define PB_TILE_DIRECT 16  // like LEAN (plain pixels, model anchored at the box origin) but the box is too sparse /
                           // large to stage: samples are fetched with unaligned global loads, no LDS
#define PB_TILE_BLACK 8  // every pixel of the tile (inside the image) is black: the hot kernel only stores zeros
#define PB_TILE_SKIP 32  // launch-order table only: an empty wave slot (beyond the image)
#define PB_TILE_SOLO 512  // launch-order table of a double-fisheye plan only: the tile sees ONE eye, the entry is that eye's (flags: its LEAN / DIRECT / BLACK)
// MASKED (with DIRECT; single sources): plain except that some of its pixels are INVALID destination pixels - the ring of tiles along
// the edge of a fisheye destination's image circle.  The model runs over the whole tile (the coordinate field continues smoothly
// past the validity boundary), the direct-gather path skips the loads of the invalid pixels (exact integer test) and paints them
// black.  Without the class these tiles take the generic path, three times as slow, and a single launch ENDS on them (c2: 586 tiles,
// 3.8 us of a 41.5 us launch - experiments/session_r3_k.sh).
#define PB_TILE_MASKED 1024
#define PB_TILE_EYE_R 2048  // launch-order table of a double-fisheye plan only: a SOLO entry that is the RIGHT eye's (the bilinear mode clamps taps to the eye)
// COARSE: somewhere on a sampled pixel the model's coordinate is more than PB_COARSE_PX from the faithful pre-truncation
// coordinate (a 32-px tile of a SMALL image spans tens of degrees; the edge of a lens's domain).  Harmless for the reference's
// nearest sampling - the fix list holds every pixel whose truncation differs - but the opt-in bilinear mode interpolates AT the
// coordinate: it leaves such tiles to its float64 pass (pb_certify_kernel measures, the bilinear kernels obey).
#define PB_TILE_COARSE 4096
#define PB_COARSE_PX 0x1p-10  // 1/1024 px: at most half an LSB of a channel on the steepest possible content
// TD3 (round 5; the bilinear mode only): the model's terms of TOTAL degree <= 3 (10 of the 25 coefficient pairs) alone stay within
// PB_COARSE_PX of the faithful coordinate on every sampled pixel - measured by pb_certify_kernel like COARSE, against the float64 chain,
// with the very evaluation the bilinear tile code then runs (pb_collapse_row_td3 / pb_eval_row_td3).  The coordinate field of a 32 x 32
// tile is smooth on the scale of the whole image: its quartic terms are ~1e-6 px away from a projection's singular points, and the
// coordinate costs 4.5 packed multiply-adds per pixel instead of 9 (experiments/r5/degree_study.py: every tile of c1 and c5, 83 % of
// c2's, 62 % of c3's by the coefficient bound alone).  The nearest mode never looks at the flag: its exactness is certified for the
// full model.
#define PB_TILE_TD3 8192
#define PB_TILE_TAB_Y 16384  // bilinear mode, tiles served from the exact coordinate table: the slot is stored transposed and walked by rows (pb_bilinear_orient_kernel)
// launch-order table of a double-fisheye plan's bilinear mode only (the PAIR layout, pb_kernels_tile.hpp): a two-eye tile takes two
// slots of one workgroup - TWO: the LEFT eye's entry (its wave blends and stores), PAIR_R: the RIGHT eye's entry (its wave hands its
// samples over through LDS); PB_TILE_SKIP | PB_TILE_TWO: an empty slot of a pair workgroup (it still meets the workgroup's barrier)
#define PB_TILE_TWO 32768
#define PB_TILE_PAIR_R 262144
// HALVES (bilinear launch table only, round 5): a plain tile whose source box exceeds the window budget but whose TOP and BOTTOM halves
// (output rows 0-15 / 16-31) each fit: the wave stages the two half windows one after the other in its LDS region and samples 8 pixels
// per lane from each - the window path's cost per pixel instead of the direct-gather path's (c2: 4 392 of its ~5 000 direct tiles).
// The entry's win_c0 and bil_off (made negative: such a tile has no coordinate-table slot) hold the two half windows (PB_HALF_* fields
// below), found at plan time with the hot path's own evaluation.
#define PB_TILE_HALVES 65536
#define PB_HALF_PACK(dr, dc, rows, n16) ((int)((unsigned)(dr) | ((unsigned)(dc) << 8) | ((unsigned)(rows) << 18) | ((unsigned)((n16) - 1) << 25)))
#define PB_HALF_DR(h) ((unsigned)(h) & 0xFFu)
#define PB_HALF_DC(h) (((unsigned)(h) >> 8) & 0x3FFu)
#define PB_HALF_ROWS(h) (((unsigned)(h) >> 18) & 0x7Fu)
#define PB_HALF_N16(h) ((((unsigned)(h) >> 25) & 0x3Fu) + 1u)
// TAB_PLAIN (bilinear mode, tiles served from the exact coordinate table, round 5): every live pixel's taps lie inside the frame's
// columns (and the eye's half), none at the panorama's seam, its rows at most one beyond the image, and an 8-byte load at either row of
// any tap stays inside the buffer (pb_bilinear_orient_kernel checks the slot's own coordinates): the table path then runs without
// column clamps, wrap, repeated-tap cases or end-of-frame handling (rows clamp with two instructions) - a third of its instructions (c3: its 1 506 table tiles took 3 100 vector instructions each, 38 % of the
// launch's, against 620 for a window tile).  Same taps, same weights, same arithmetic: the pixels do not depend on the flag.
#define PB_TILE_TAB_PLAIN 131072
#define PB_TILE_W_UNIT_BIT 64  // == PB_TILE_W_UNIT (pb_kernels_double.hpp): blend factors exactly 1.0 for every pixel of the tile
#define PB_LEAN_MAX_PASSES 24  // window rows / rows-per-load-instruction of a LEAN tile (register staging depth)

typedef float pb_f2 __attribute__((ext_vector_type(2)));

struct __attribute__((aligned(256))) PbTileEntry {
    int32_t anchor_r, anchor_c;  // integer anchors: coordinate = anchor + polynomial
    int32_t flags;
    int32_t win_rows;            // rows of the source window (after the LDS cap when LEAN)
    float c[25][2];              // c[m*5+n] = (row, col) coefficient pair of v^m u^n  (v: rows, u: columns, in [-1, 1])
    // source window = bounding box of the tile's samples (hot-path values, found by the plan builder):
    // rows [win_r0, win_r0 + win_rows), columns [win_c0, win_c0 + win_cols); LEAN: origin == anchors
    int32_t win_r0, win_c0, win_cols;
    int32_t win_n16, win_a0;     // LEAN: 16-byte chunks per row, byte offset of column win_c0 in its first chunk
    int32_t fix_off, fix_cnt;    // this tile's slice of the plan's fix-pixel list (<= PB_TILE_FAIL_LIMIT entries)
    int32_t aux_off;             // double sources, left-eye entry: the tile's slot in the plan's latitude table (PB_TILE_W_LAT)
    int32_t tile_xy;             // launch-order copy (pb_launch_table_kernel): the tile this entry belongs to, ty << 16 | tx
                                 // (16 bits each: tile plans exist for destinations up to 16384 px a side - pb_fast_possible)
    int32_t bil_off;             // opt-in bilinear mode: the tile's slot in the plan's exact coordinate table (pb_kernels_bilinear.hpp), -1 = the model serves it
};
static_assert(sizeof(PbTileEntry) == 256, "PbTileEntry must be 256 bytes");

// Lagrange -> monomial matrix for the nodes u = {-1, -sqrt(1/2), 0, sqrt(1/2), 1}:
// L_i(u) = sum_m PB_A[m][i] u^m
__device__ static const double PB_A[5][5] = {
    {0.0, 0.0, 1.0, 0.0, 0.0},
    {0.5, -1.4142135623730951, 0.0, 1.4142135623730951, -0.5},
    {-0.5, 2.0, -3.0, 2.0, -0.5},
    {-1.0, 1.4142135623730951, 0.0, -1.4142135623730951, 1.0},
    {1.0, -2.0, 2.0, -2.0, 1.0},
};
__device__ static const double PB_NODE[5] = {-1.0, -0.7071067811865476, 0.0, 0.7071067811865476, 1.0};

__device__ __forceinline__ void pb_wave_sync() {
    // LDS hand-off between lanes of ONE wave: order the ds ops, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// faithful chain at a real position, invalid flag ignored (model building only)
template <int SRC_KIND>
__device__ __forceinline__ void pb_chain_real(const PbParams& P, double fi, double fj, double& f0, double& f1) {
    PbCoord c = pb_dst_coord_real(P, fi, fj);
    c = pb_rotate_all(P, c);
    pb_src_pretrunc<SRC_KIND>(P, c, f0, f1);
}

// faithful chain at an integer pixel -> source index (-1 = black); THE reference path
template <int SRC_KIND>
__device__ __forceinline__ int pb_exact_index_of(const PbParams& P, const PbCoord& c) {
    if (SRC_KIND == PB_KIND_PANO) return pb_src_pano_index(P, c);
    double sl, cl;
    pb_expi_np(c.lon, &sl, &cl);
    return pb_src_index_sc<SRC_KIND>(P, c, sl, cl);  // (one eye of a double frame: that eye alone)
}
template <int SRC_KIND>
__device__ __forceinline__ int pb_exact_index(const PbParams& P, int i, int j) {
    PbCoord c = pb_dst_coord(P, i, j);
    c = pb_rotate_all(P, c);
    return pb_exact_index_of<SRC_KIND>(P, c);
}

// ---- the hot-path model evaluation (float32, bit-reproducible) ----------------------
// Row and column coordinate are evaluated as one float2 (v_pk_fma_f32): same IEEE results as two
// scalar fmaf chains, half the instructions.
__device__ __forceinline__ pb_f2 pb_fma2(pb_f2 a, float b, pb_f2 c) {
    const pb_f2 bb = {b, b};
    return __builtin_elementwise_fma(a, bb, c);
}

struct PbRowModel {
    pb_f2 a[5];      // per-row collapsed coefficients (polynomials in u), (row, col) pairs
    int va, vb, x2;  // validity window on x2^2 and the doubled x offset of the lane's first pixel
    int anchor_r, anchor_c;
};

__device__ __forceinline__ float pb_tile_coord(int t) {  // pixel offset 0..31 -> [-1, 1]
    const float half = 0.5f * (PB_TILE - 1), inv_half = 1.0f / (0.5f * (PB_TILE - 1));
    return ((float)t - half) * inv_half;
}

// collapse the tile model along v for row y (0..31): 5 float2 coefficients of the row polynomial in u
__device__ __forceinline__ void pb_collapse_row(const PbTileEntry* __restrict__ e, int y, pb_f2 a[5]) {
    const float v = pb_tile_coord(y);
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        pb_f2 s = {e->c[20 + n][0], e->c[20 + n][1]};
#pragma unroll
        for (int m = 3; m >= 0; --m) {
            const pb_f2 cm = {e->c[m * 5 + n][0], e->c[m * 5 + n][1]};
            s = pb_fma2(s, v, cm);
        }
        a[n] = s;
    }
}

// collapse the tile model along u for column x (0..31): 5 float2 coefficients of the column polynomial in v
__device__ __forceinline__ void pb_collapse_col(const PbTileEntry* __restrict__ e, int x, pb_f2 b[5]) {
    const float u = pb_tile_coord(x);
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        pb_f2 s = {e->c[m * 5 + 4][0], e->c[m * 5 + 4][1]};
#pragma unroll
        for (int n = 3; n >= 0; --n) {
            const pb_f2 cn = {e->c[m * 5 + n][0], e->c[m * 5 + n][1]};
            s = pb_fma2(s, u, cn);
        }
        b[m] = s;
    }
}

__device__ __forceinline__ pb_f2 pb_eval_row(const pb_f2 a[5], float u) {
    pb_f2 f = a[4];
#pragma unroll
    for (int n = 3; n >= 0; --n) f = pb_fma2(f, u, a[n]);
    return f;
}

// The same with the terms of total degree <= 3 only (PB_TILE_TD3 tiles, bilinear mode): c[m][n], m + n <= 3.
__device__ __forceinline__ void pb_collapse_row_td3(const PbTileEntry* __restrict__ e, int y, pb_f2 a[4]) {
    const float v = pb_tile_coord(y);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        pb_f2 s = {e->c[(3 - n) * 5 + n][0], e->c[(3 - n) * 5 + n][1]};
#pragma unroll
        for (int m = 2 - n; m >= 0; --m) {
            const pb_f2 cm = {e->c[m * 5 + n][0], e->c[m * 5 + n][1]};
            s = pb_fma2(s, v, cm);
        }
        a[n] = s;
    }
}
__device__ __forceinline__ void pb_collapse_col_td3(const PbTileEntry* __restrict__ e, int x, pb_f2 b[4]) {
    const float u = pb_tile_coord(x);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        pb_f2 s = {e->c[m * 5 + (3 - m)][0], e->c[m * 5 + (3 - m)][1]};
#pragma unroll
        for (int n = 2 - m; n >= 0; --n) {
            const pb_f2 cn = {e->c[m * 5 + n][0], e->c[m * 5 + n][1]};
            s = pb_fma2(s, u, cn);
        }
        b[m] = s;
    }
}
__device__ __forceinline__ pb_f2 pb_eval_row_td3(const pb_f2 a[4], float u) {
    pb_f2 f = a[3];
#pragma unroll
    for (int n = 2; n >= 0; --n) f = pb_fma2(f, u, a[n]);
    return f;
}

// Collapses the tile model along v for the row `y` (0..31) of tile (X0, Y0) and prepares the
// exact integer validity test of that row: invalid <=> va <= (x2 + 2k)^2 < vb.
__device__ __forceinline__ void pb_model_row(const PbParams& P, const PbTileEntry* __restrict__ e, int X0, int Y0, int y,
                                             int xh, PbRowModel& R) {
    pb_collapse_row(e, y, R.a);
    R.anchor_r = e->anchor_r;
    R.anchor_c = e->anchor_c;
    R.va = 0x7fffffff;
    R.vb = 0x7fffffff;
    R.x2 = 0;
    if (P.dst.kind != PB_KIND_PANO) {
        // a double-destination tile that straddles the two eyes is listed as failed at plan
        // creation (the side chosen here would be wrong for part of it)
        const int side = (P.dst.kind == PB_KIND_DOUBLE) && (X0 >= P.dst_half_w);
        const int wc = (P.dst.kind == PB_KIND_DOUBLE) ? P.dst_half_w : P.dst.width;
        const long long y2 = (long long)(P.dst.height - 1) - 2ll * (Y0 + y);
        const long long A = P.inv_lo[side] - y2 * y2, B = P.inv_hi[side] - y2 * y2;
        R.va = (int)(A < 0 ? 0 : (A > 0x7fffffffll ? 0x7fffffffll : A));
        R.vb = (int)(B < 0 ? 0 : (B > 0x7fffffffll ? 0x7fffffffll : B));
        R.x2 = 2 * (X0 + xh - (side ? P.dst_half_w : 0)) - (wc - 1);
    }
}

__device__ __forceinline__ bool pb_row_px_invalid(const PbRowModel& R, int k) {
    const int xx = R.x2 + 2 * k, q = (int)__mul24(xx, xx);
    return q >= R.va && q < R.vb;
}

// raw (unwrapped) source row / column of pixel x = xh + k of the row prepared in R
__device__ __forceinline__ void pb_model_px_raw(const PbRowModel& R, int xh, int k, int& r, int& c, pb_f2& f) {
    f = pb_eval_row(R.a, pb_tile_coord(xh + k));
    r = R.anchor_r + (int)floorf(f.x);
    c = R.anchor_c + (int)floorf(f.y);
}

// The reference converts a fisheye source's coordinates to pixels by TRUNCATION (astype(int), projection.py:254-259):
// anything in (-1, 0) becomes pixel 0 and is sampled, where the floor of the model says -1.  (An exact -1.0
// differs the other way; such pixels end on the plan's fix list like every other model miss.)
template <int SRC_KIND>
__device__ __forceinline__ void pb_model_trunc_edge(const PbParams& P, int& r, int& c) {
    if (SRC_KIND == PB_KIND_PANO) return;
    if (r == -1) r = 0;
    if (SRC_KIND == PB_KIND_EYE_R) {
        if (c == P.src.width) c = P.src.width - 1;  // the mirrored column of x in (-1, 0)
    } else if (c == -1) {
        c = 0;
    }
}

// Source (row, col) of pixel x = xh + k packed as (r << 16 | c), or -1 = black.  Needs src dims < 32768.
template <int SRC_KIND>
__device__ __forceinline__ int pb_model_px_rc(const PbParams& P, const PbRowModel& R, int xh, int k) {
    int r, c;
    pb_f2 f;
    pb_model_px_raw(R, xh, k, r, c, f);
    const int h = P.src.height, w = P.src.width;
    if (SRC_KIND == PB_KIND_PANO) {
        if (r >= h) r -= h;  // lat = pi wraps to row 0
        if (c >= w) c -= w;
    }
    pb_model_trunc_edge<SRC_KIND>(P, r, c);
    int cmin, cmax;
    pb_src_col_range<SRC_KIND>(P, cmin, cmax);
    int id = ((unsigned)r < (unsigned)h && c >= cmin && c < cmax) ? ((r << 16) | c) : -1;
    if (pb_row_px_invalid(R, k)) id = -1;  // invalid destination pixel -> black
    return id;
}

// Source index of pixel x = xh + k of the row prepared in R (or -1 = black).
template <int SRC_KIND>
__device__ __forceinline__ int pb_model_px(const PbParams& P, const PbRowModel& R, int xh, int k) {
    int r, c;
    pb_f2 f;
    pb_model_px_raw(R, xh, k, r, c, f);
    const int h = P.src.height, w = P.src.width;
    if (SRC_KIND == PB_KIND_PANO) {
        if (r >= h) r -= h;
        if (c >= w) c -= w;
    }
    pb_model_trunc_edge<SRC_KIND>(P, r, c);
    int cmin, cmax;
    pb_src_col_range<SRC_KIND>(P, cmin, cmax);
    int id = ((unsigned)r < (unsigned)h && c >= cmin && c < cmax) ? (int)__umul24(r, w) + c : -1;
    if (pb_row_px_invalid(R, k)) id = -1;
    return id;
}
