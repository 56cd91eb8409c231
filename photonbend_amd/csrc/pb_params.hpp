// pb_params.hpp - the parameter block one remap launch carries (kernel argument,
// passed by value) and the host code that derives it.  Every derived constant is
// computed on the host in IEEE double with the SAME operation order the reference
// uses in Python (file:line cited per field), so that kernels only ever replay
// per-pixel arithmetic.
#pragma once
#include <cstdint>
#include <cstring>

#include "../../include/photonbend_hip.h"

#define PB_PI 3.141592653589793  // == numpy.pi == M_PI

// kernel variant switches (PbParams::exp_flags; PB_EXP environment variable at plan creation): used only by the
// -DPB_ABLATION diagnostic build (experiments/), which skips tile classes / loads / stores to attribute time.

struct PbEnd {
    int32_t kind, lens, height, width;
    double fov, f_distance;
};

struct PbParams {
    PbEnd dst, src;
    int32_t n_rot;
    int32_t win_budget;  // bytes of LDS window per wave (hot kernels; chosen per plan, multiple of 16, <= PB_WINLDS_MAX)
    double R[PB_MAX_ROTATIONS][9];

    // ---- destination side -------------------------------------------------
    double dst_half_fov;   // fov / 2            projection.py:160, :357
    double dst_right_min;  // pi - fov / 2.0     projection.py:358-360 (double)
    double dst_x0, dst_y0; // first linspace samples, projection.py:177-180, :390-400
    int32_t dst_half_w;    // W // 2 (double)    projection.py:355
    int32_t exp_flags;     // kernel variant switches (PB_EXP_*): paths, never pixels
    // pano destination: np.linspace(start, stop, num) = k*step + start, last = stop
    double pano_lon_start, pano_lon_stop, pano_lon_step;  // projection.py:500-504
    double pano_lat_step;                                 // projection.py:505

    // ---- source side --------------------------------------------------------
    double src_hseg, src_wseg, src_half_w;  // pi/h, pi/(w/2), w/2   projection.py:539-543
    int64_t src_nan_row, src_nan_col;       // INT64_MIN floor-mod h / w  (NaN -> int quirk)
    double src_cy, src_cx, src_cx_r;        // h/2-0.5, w/2-0.5  projection.py:274 (per eye for double)
    int32_t src_eye_w;                      // width of one eye (w // 2)  projection.py:413
    int32_t src_eye_w_right;                // w - w // 2
    double rect_max;                        // to_radians(89)  lens.py:91,98
    // destination validity as integer thresholds on n4 = (2x)^2 + (2y)^2, found at plan
    // creation by bisection with the exact predicate (pb_threshold_kernel):
    // invalid  <=>  inv_lo[side] <= n4 < inv_hi[side]   (side 0 = left / single, 1 = right eye)
    int64_t inv_lo[2], inv_hi[2];
    // fixed point used by the tile fast path: coordinate * 2^fx_shift fits an int32
    double fx_scale;
    int32_t fx_shift, fx_mask;
    int32_t thresholds_ready;               // 0: kernels must evaluate the predicate per pixel
    int32_t fast_tiles;                     // 1: per-tile polynomial models allowed (certified plan)
    double mrg_min, mrg_max, mrg_range, mrg_max_safe;  // projection.py:414-418
};

static inline int64_t pb_floor_mod_i64(int64_t a, int64_t n) {
    int64_t r = a % n;
    return (r < 0) ? r + n : r;
}

// np.linspace(start, stop, num): step = (stop - start) / (num - 1)
static inline double pb_linspace_step(double start, double stop, int num) {
    return (num > 1) ? (stop - start) / (double)(num - 1) : 0.0;
}

static inline void pb_derive(PbParams& P) {
    const double pi = PB_PI;
    // destination
    {
        const PbEnd& d = P.dst;
        const double W = (double)d.width, H = (double)d.height;
        P.dst_half_fov = d.fov / 2;
        P.dst_right_min = pi - (d.fov / 2.0);
        P.dst_half_w = d.width / 2;
        P.dst_y0 = H / 2 - 0.5;
        if (d.kind == PB_KIND_DOUBLE) {
            const double half = (double)P.dst_half_w;
            P.dst_x0 = -half / 2 + 0.5;
        } else {
            P.dst_x0 = -W / 2 + 0.5;
        }
        const double q = pi / W / 2;
        P.pano_lon_start = -pi + q;
        P.pano_lon_stop = pi - q;
        P.pano_lon_step = pb_linspace_step(P.pano_lon_start, P.pano_lon_stop, d.width);
        P.pano_lat_step = pb_linspace_step(0.0, pi, d.height);
    }
    // source
    {
        const PbEnd& s = P.src;
        const double w = (double)s.width, h = (double)s.height;
        P.src_wseg = pi / (w / 2);
        P.src_hseg = pi / h;
        P.src_half_w = w / 2;
        P.src_nan_row = pb_floor_mod_i64(INT64_MIN, s.height);
        P.src_nan_col = pb_floor_mod_i64(INT64_MIN, s.width);
        P.src_eye_w = s.width / 2;
        P.src_eye_w_right = s.width - s.width / 2;
        P.src_cy = h / 2 - 0.5;
        if (s.kind == PB_KIND_DOUBLE) {
            P.src_cx = (double)P.src_eye_w / 2 - 0.5;
            P.src_cx_r = (double)P.src_eye_w_right / 2 - 0.5;
        } else {
            P.src_cx = w / 2 - 0.5;
            P.src_cx_r = P.src_cx;
        }
        const double ref = (s.fov / 2) - (pi / 2);
        P.mrg_min = pi / 2 - ref;
        P.mrg_max = pi / 2 + ref;
        P.mrg_range = 2.0 * ref;
        P.mrg_max_safe = P.mrg_max + (0.5 / 180 * pi);
    }
    P.rect_max = 89.0 / 180 * pi;
    {
        int maxdim = P.src.height > P.src.width ? P.src.height : P.src.width;
        int bits = 1;
        while ((1ll << bits) < (long long)maxdim + 1) ++bits;  // 2^bits >= maxdim + 1
        int shift = 30 - bits;                                  // coordinate < 2 * maxdim stays below 2^31
        if (shift > 20) shift = 20;
        P.fx_shift = shift;
        P.fx_mask = (1 << shift) - 1;
        P.fx_scale = (double)(1ll << shift);
    }
}
