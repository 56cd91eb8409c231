// pb_tile.hpp - the per-tile fast path of the fused remap.
//
// One WAVE owns one 32x32 tile of output pixels.  The source coordinate field
// (pre-truncation row/column of the source sample) is smooth inside a tile almost
// everywhere, so instead of running the float64 transcendental chain 1024 times the
// wave
//   1. runs it 41 times, at the 5x5 Chebyshev-Lobatto nodes of the tile plus 16 check
//      points, with the SAME faithful stage functions (pb_stages.hpp) evaluated at real
//      pixel positions ("node pass": one instruction stream, 41 of 64 lanes busy);
//   2. turns the 25 node values into a degree-4 x degree-4 polynomial per coordinate
//      and accepts the model only if it reproduces the 16 check points to PB_TILE_TOL
//      (tiles that straddle a seam, a pole, the fisheye centre, a NaN region or a
//      steep lens edge fail and take the per-pixel faithful path);
//   3. evaluates the model per pixel in float64 (8 FMAs after a per-row collapse),
//      truncates, and sends every pixel whose coordinate lies within PB_TILE_EPS of an
//      integer - where a 1e-7 px model error could flip the truncation - back through
//      the faithful chain ("fragile" pixels, ~1e-5 of all);
//   4. decides validity of camera/double destination pixels with exact integer
//      thresholds on (2x)^2 + (2y)^2 found at plan creation by bisection with the
//      faithful predicate.
// Exactness does not rest on the tolerances alone: pb_plan_create certifies the plan
// by comparing this path's integer index map with the faithful one for EVERY pixel
// and disables the fast path if a single index differs.
#pragma once
#include "pb_stages.hpp"

#define PB_TILE 32
#define PB_TILE_PITCH 33        // LDS row pitch of the index tile (ints): conflict-free column writes
#define PB_TILE_TOL 2.5e-7      // model accepted if |model - exact| <= TOL px at all check points
// pixels closer than 2^-fx_shift px (7.6e-6 for sources up to 8192 px) to an integer
// coordinate are re-done exactly; must exceed the model error bound 8 * PB_TILE_TOL
#define PB_IDX_FRAGILE (-2)

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
__device__ static const double PB_CHECK[4] = {-0.9, -0.38, 0.38, 0.9};

struct PbWaveLds {
    double F[2][25];               // node values, then (in place) monomial coefficients C[coord][m*5+n]
    int idx[PB_TILE * PB_TILE_PITCH];
};

__device__ __forceinline__ void pb_wave_sync() {
    // LDS hand-off between lanes of ONE wave: order the ds ops, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// faithful chain at a real position, invalid flag ignored
template <int SRC_KIND>
__device__ __forceinline__ void pb_chain_real(const PbParams& P, double fi, double fj, double& f0, double& f1) {
    PbCoord c = pb_dst_coord_real(P, fi, fj);
    for (int k = 0; k < P.n_rot; ++k) c = pb_rotate(P.R[k], c);
    pb_src_pretrunc<SRC_KIND>(P, c, f0, f1);
}

// faithful chain at an integer pixel -> source index (-1 = black); the reference path
template <int SRC_KIND>
__device__ __forceinline__ int pb_exact_index(const PbParams& P, int i, int j) {
    PbCoord c = pb_dst_coord(P, i, j);
    for (int k = 0; k < P.n_rot; ++k) c = pb_rotate(P.R[k], c);
    return (SRC_KIND == PB_KIND_PANO) ? pb_src_pano_index(P, c) : pb_src_camera_index(P, c);
}

// exact validity of destination pixel (i, j): integer thresholds when the plan has them
__device__ __forceinline__ bool pb_dst_invalid(const PbParams& P, int i, int j) {
    if (P.dst.kind == PB_KIND_PANO) return false;
    int side = 0, jj = j, wc = P.dst.width;
    if (P.dst.kind == PB_KIND_DOUBLE) {
        side = j >= P.dst_half_w;
        jj = side ? j - P.dst_half_w : j;
        wc = P.dst_half_w;
    }
    const long long x2 = 2ll * jj - (wc - 1), y2 = (long long)(P.dst.height - 1) - 2ll * i;
    const long long n4 = x2 * x2 + y2 * y2;
    if (P.thresholds_ready) return n4 >= P.inv_lo[side] && n4 < P.inv_hi[side];
    return pb_dst_inv_pred(P, n4, side != 0, nullptr);
}

// Builds the tile model.  Returns true (wave-uniform) when the model may be used; the
// coefficients are then in L.F[coord][m*5+n] (value = sum_mn C v^m u^n, u along x).
template <int SRC_KIND>
__device__ __forceinline__ bool pb_tile_model(const PbParams& P, PbWaveLds& L, int lane, int X0, int Y0) {
    const double half = 0.5 * (PB_TILE - 1);
    double u = 0.0, v = 0.0;
    const bool node = lane < 25, check = lane >= 25 && lane < 41;
    if (node) {
        v = PB_NODE[lane / 5];
        u = PB_NODE[lane % 5];
    } else if (check) {
        v = PB_CHECK[(lane - 25) >> 2];
        u = PB_CHECK[(lane - 25) & 3];
    }
    double f0 = 0.0, f1 = 0.0;
    if (node || check) pb_chain_real<SRC_KIND>(P, (double)Y0 + half + half * v, (double)X0 + half + half * u, f0, f1);
    bool bad = (node || check) && !(fabs(f0) < 1.0e9 && fabs(f1) < 1.0e9);  // NaN / inf / absurd
    if (node) {
        L.F[0][lane] = f0;
        L.F[1][lane] = f1;
    }
    pb_wave_sync();
    double c0 = 0.0, c1 = 0.0;
    if (node) {  // lane = m*5+n: C_mn = sum_ij A[m][i] A[n][j] F[i][j]   (i: rows / v, j: columns / u)
        const int m = lane / 5, n = lane % 5;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double am = PB_A[m][i];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const double w = am * PB_A[n][j];
                c0 = fma(w, L.F[0][i * 5 + j], c0);
                c1 = fma(w, L.F[1][i * 5 + j], c1);
            }
        }
    }
    pb_wave_sync();
    if (node) {
        L.F[0][lane] = c0;
        L.F[1][lane] = c1;
    }
    pb_wave_sync();
    if (check) {
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int m = 4; m >= 0; --m) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int n = 4; n >= 0; --n) {
                a0 = fma(a0, u, L.F[0][m * 5 + n]);
                a1 = fma(a1, u, L.F[1][m * 5 + n]);
            }
            p0 = fma(p0, v, a0);
            p1 = fma(p1, v, a1);
        }
        bad = bad || !(fabs(p0 - f0) <= PB_TILE_TOL && fabs(p1 - f1) <= PB_TILE_TOL);
    }
    return !__builtin_amdgcn_ballot_w64(bad);
}

// Truncates one modelled coordinate pair to the source index; flags fragile pixels.
// The coordinate is converted to fixed point with P.fx_shift fractional bits (one
// v_cvt_i32_f64, saturating, NaN -> 0): the integer part is the truncated coordinate,
// and a fraction field of all zeros or all ones means "within 2^-fx_shift px of an
// integer" = fragile.  Negative values are sent to the faithful path as well.
template <int SRC_KIND>
__device__ __forceinline__ int pb_model_index(const PbParams& P, double f0, double f1) {
    const int q0 = (int)(f0 * P.fx_scale), q1 = (int)(f1 * P.fx_scale);
    const unsigned mask = (unsigned)P.fx_mask;
    const bool safe = (((unsigned)(q0 + 1) & mask) > 1u) && (((unsigned)(q1 + 1) & mask) > 1u) && ((q0 | q1) >= 0);
    int t0 = q0 >> P.fx_shift, t1 = q1 >> P.fx_shift;
    const int h = P.src.height, w = P.src.width;
    if (SRC_KIND == PB_KIND_PANO) {
        if (t0 >= h) t0 -= h;  // lat = pi wraps to row 0
        if (t1 >= w) t1 -= w;
        const bool in = (t0 < h) && (t1 < w);
        return (safe && in) ? (int)__umul24(t0, w) + t1 : PB_IDX_FRAGILE;
    } else {
        if (!safe) return PB_IDX_FRAGILE;
        const bool in = (t0 < h) && (t1 < w);
        return in ? (int)__umul24(t0, w) + t1 : -1;
    }
}

// Fills L.idx with the source index of every pixel of the tile (X0, Y0).
// MODE 0: fast path allowed; MODE 1: faithful path only.
template <int SRC_KIND>
__device__ __forceinline__ bool pb_tile_indices(const PbParams& P, PbWaveLds& L, int lane, int X0, int Y0, bool allow_fast,
                                                unsigned* n_exact = nullptr) {
    const int y = lane & 31, xh = (lane >> 5) * 16;
    const int i = Y0 + y;
    const bool row_in = i < P.dst.height;
    bool ok = false;
    if (allow_fast && !(P.dst.kind == PB_KIND_DOUBLE && X0 < P.dst_half_w && X0 + PB_TILE > P.dst_half_w))
        ok = pb_tile_model<SRC_KIND>(P, L, lane, X0, Y0);
    if (ok) {
        const double half = 0.5 * (PB_TILE - 1), inv_half = 1.0 / (0.5 * (PB_TILE - 1));
        const double v = ((double)y - half) * inv_half;
        double a[2][5];
#pragma unroll
        for (int n = 0; n < 5; ++n) {  // collapse the v direction for this lane's row
            double s0 = L.F[0][20 + n], s1 = L.F[1][20 + n];
#pragma unroll
            for (int m = 3; m >= 0; --m) {
                s0 = fma(s0, v, L.F[0][m * 5 + n]);
                s1 = fma(s1, v, L.F[1][m * 5 + n]);
            }
            a[0][n] = s0;
            a[1][n] = s1;
        }
        // validity of this row's pixels: lo <= x2^2 + y2^2 < hi  <=>  va <= x2^2 < vb  (32-bit, clamped)
        int va = 0x7fffffff, vb = 0x7fffffff, x2 = 0;
        if (P.dst.kind != PB_KIND_PANO) {
            const int side = (P.dst.kind == PB_KIND_DOUBLE) && (X0 >= P.dst_half_w);
            const int wc = (P.dst.kind == PB_KIND_DOUBLE) ? P.dst_half_w : P.dst.width;
            const long long y2 = (long long)(P.dst.height - 1) - 2ll * i;
            const long long A = P.inv_lo[side] - y2 * y2, B = P.inv_hi[side] - y2 * y2;
            va = (int)(A < 0 ? 0 : (A > 0x7fffffffll ? 0x7fffffffll : A));
            vb = (int)(B < 0 ? 0 : (B > 0x7fffffffll ? 0x7fffffffll : B));
            x2 = 2 * (X0 + xh - (side ? P.dst_half_w : 0)) - (wc - 1);
        }
        const double u0 = ((double)xh - half) * inv_half;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int x = xh + k;
            const double u = fma((double)k, inv_half, u0);
            double f0 = a[0][4], f1 = a[1][4];
#pragma unroll
            for (int n = 3; n >= 0; --n) {
                f0 = fma(f0, u, a[0][n]);
                f1 = fma(f1, u, a[1][n]);
            }
            int id = pb_model_index<SRC_KIND>(P, f0, f1);
            const int xx = x2 + 2 * k, q = (int)__mul24(xx, xx);
            if (q >= va && q < vb) id = -1;                       // invalid destination pixel -> black
            if (!row_in || X0 + x >= P.dst.width) id = -1;        // outside the image
            L.idx[y * PB_TILE_PITCH + x] = id;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int x = xh + k;
            L.idx[y * PB_TILE_PITCH + x] = (row_in && X0 + x < P.dst.width) ? PB_IDX_FRAGILE : -1;
        }
    }
    // one copy of the faithful chain serves both the fragile pixels of a modelled tile
    // (rare: the ballot skips the body) and every pixel of a tile without a model
#pragma unroll 1
    for (int k = 0; k < 16; ++k) {
        const int x = xh + k;
        const bool need = L.idx[y * PB_TILE_PITCH + x] == PB_IDX_FRAGILE;
        if (__builtin_amdgcn_ballot_w64(need)) {
            if (need) {
                L.idx[y * PB_TILE_PITCH + x] = pb_exact_index<SRC_KIND>(P, i, X0 + x);
                if (n_exact) ++*n_exact;
            }
        }
    }
    pb_wave_sync();
    return ok;
}
