// pb_stages.hpp - the three per-pixel stages of photonbend's remap, as device
// functions for ONE output pixel, in "faithful" float64: every IEEE-exact
// operation (add, mul, div, sqrt, fma) is replayed in the reference's order with
// contraction disabled (-ffp-contract=off), and the NumPy cast quirks are spelled out.
// Transcendentals, each the very function the reference's NumPy runs on the machine that made the goldens (x86-64, FMA, AVX512_SKX,
// glibc 2.35, NumPy 2.2.6; SURVEY 2, primitive table), restated operation for operation and bit-equal on every argument tested:
//   np.sin, np.cos             glibc's sin / cos, `_fma` build                      pb_sin_np, pb_cos_np        (pb_math_glibc.hpp)
//   np.exp(lon * 1j)           glibc's internal sincos, plain build                 pb_expi_np
//   np.log(complex).imag       glibc's atan2, `_fma` build                          pb_arg_np
//   np.arcsin / arccos / arctan / tan   NumPy's own AVX-512 kernels (Intel SVML)    pb_asin_np ... pb_tan_np    (pb_math_np.hpp)
// None of these is correctly rounded (0.1 % to 17 % of their results are not), so nothing but their own algorithms returns their bits.
//
//   stage A  dst_coord()      pixel (i, j) -> (lat, lon, invalid)
//            CameraImage._compute_latitude_longitude  projection.py:171-194
//            DoubleCameraImage._compute_latitude_longitude  :370-406
//            PanoramaImage.get_coordinate_map  :487-513
//   stage B  rotate()         Rotation.rotate_coordinate_map  rotation.py:102-176
//   stage C  src_*_index()    process_coordinate_map minus the gather
//            projection.py:197-260 (camera), :408-462 (double), :515-547 (pano)
#pragma once
#include <hip/hip_runtime.h>

#include "pb_math.hpp"
#include "pb_params.hpp"

struct PbCoord {
    double lat, lon;
    bool inv;
};

// ---- NumPy / x86 cast semantics ------------------------------------------------
// float64 -> int64 as cvttsd2si does it: truncate toward zero; NaN, +-inf and
// anything outside int64 give INT64_MIN (SURVEY 2 "to-pixel", probe).
__device__ __forceinline__ long long pb_cvt_i64(double v) {
    if (!(fabs(v) < 9223372036854775808.0)) return (long long)0x8000000000000000ull;
    return (long long)v;
}
// float64 -> uint8 as NumPy's astype(np.uint8) does it on x86-64: truncate to a
// 32-bit integer (invalid -> 0x80000000), keep the low 8 bits (SURVEY 8 a-6 probe:
// 300.0 -> 44, -1.0 -> 255, NaN / +-inf -> 0).
__device__ __forceinline__ unsigned pb_cvt_u8(double v) {
    if (!(fabs(v) < 2147483648.0)) return 0u;
    return ((unsigned)(int)v) & 0xFFu;
}
// Python floor-mod of an int64 by a small positive n, fast when 0 <= a < 2n.
__device__ __forceinline__ int pb_floor_mod(long long a, int n) {
    if (a >= 0 && a < n) return (int)a;
    if (a >= n && a < 2ll * n) return (int)(a - n);
    long long r = a % (long long)n;
    return (int)((r < 0) ? r + n : r);
}

// ---- a-1 lens functions (array semantics of core/lens.py) --------------------
// np.tan / np.arcsin / np.arctan are NumPy's own SIMD kernels, restated bit for bit in pb_math_np.hpp; np.sin is glibc's (pb_math.hpp).
__device__ __forceinline__ double pb_lens_forward(int lens, double theta, double rect_max) {
    switch (lens) {
        case PB_LENS_EQUIDISTANT: return theta;                          // lens.py:187
        case PB_LENS_EQUISOLID: return 2.0 * pb_sin_np(theta / 2.0);     // lens.py:240-243
        case PB_LENS_STEREOGRAPHIC: return 2.0 * pb_tan_np(theta / 2.0); // lens.py:142-145
        case PB_LENS_ORTHOGRAPHIC: return pb_sin_np(theta);              // lens.py:285
        case PB_LENS_THOBY: return 1.47 * pb_sin_np(0.713 * theta);      // lens.py:332-335
        default: {                                                       // lens.py:97-103
            double t = pb_tan_np(theta);
            return (theta < 0.0 || theta > rect_max) ? __builtin_nan("") : t;
        }
    }
}
__device__ __forceinline__ double pb_lens_inverse(int lens, double r) {
    switch (lens) {
        case PB_LENS_EQUIDISTANT: return r;                              // lens.py:165
        case PB_LENS_EQUISOLID: {                                        // lens.py:206-220
            double t = 2.0 * pb_asin_np(r / 2.0);
            return (t != t) ? 0.0 : t;
        }
        case PB_LENS_STEREOGRAPHIC: return 2.0 * pb_atan_np(r / 2.0);    // lens.py:121-124
        case PB_LENS_ORTHOGRAPHIC: return pb_asin_np(r);                 // lens.py:261
        case PB_LENS_THOBY: return pb_asin_np(r / 1.47) / 0.713;         // lens.py:305
        default: return pb_atan_np(r);                                   // lens.py:71
    }
}

// atan2 as np.log(complex).imag gives it (glibc's atan2), bit for bit (pb_math_glibc.hpp) - including the octant lines
// |x| == |y| and the axes, where the pre-truncation longitude coordinate of a pano source is an exact integer (SURVEY 7 hard part 2).
__device__ __forceinline__ double pb_atan2(double y, double x) { return pb_arg_np(y, x); }

// ---- stage A ---------------------------------------------------------------------
__device__ __forceinline__ PbCoord pb_dst_coord(const PbParams& P, int i, int j) {
    PbCoord c;
    const PbEnd& d = P.dst;
    if (d.kind == PB_KIND_PANO) {
        // linspace: k * step + start (two roundings), last sample = stop
        c.lat = (i == d.height - 1 && d.height > 1) ? PB_PI : ((double)i * P.pano_lat_step + 0.0);
        c.lon = (j == d.width - 1 && d.width > 1) ? P.pano_lon_stop : ((double)j * P.pano_lon_step + P.pano_lon_start);
        c.inv = false;
        return c;
    }
    double x, y;
    y = P.dst_y0 - (double)i;  // exact: i * (-1.0) + (H/2 - 0.5)
    bool right = false;
    if (d.kind == PB_KIND_DOUBLE) {
        right = j >= P.dst_half_w;
        const int jj = right ? j - P.dst_half_w : j;
        x = (double)jj + P.dst_x0;
        if (right) x = -x;  // projection.py:394
    } else {
        x = (double)j + P.dst_x0;
    }
    const double dist = sqrt(x * x + y * y) / d.f_distance;  // projection.py:186, :375
    double lat = pb_lens_inverse(d.lens, dist);
    if (d.kind == PB_KIND_DOUBLE && right) {
        lat = (lat * -1.0) + PB_PI;           // projection.py:381-382
        c.inv = lat < P.dst_right_min;        // projection.py:358-360
    } else {
        c.inv = lat > P.dst_half_fov;         // projection.py:160, :357
    }
    c.lat = lat;
    c.lon = pb_atan2(y, x);                  // projection.py:193, :383
    return c;
}


// ---- smooth continuation of stages A..C at REAL pixel positions -------------------
// Used only to build per-tile polynomial models (pb_tile.hpp): same formulas, but the
// pixel position is a real number, the invalid flag is ignored and the linspace
// "last sample = stop" override is dropped (a 1-ulp kink that a model accurate to
// 1e-7 px cannot see).  Never used to produce an output value directly.
__device__ __forceinline__ PbCoord pb_dst_coord_real(const PbParams& P, double fi, double fj) {
    PbCoord c;
    const PbEnd& d = P.dst;
    c.inv = false;
    if (d.kind == PB_KIND_PANO) {
        c.lat = fi * P.pano_lat_step;
        c.lon = fj * P.pano_lon_step + P.pano_lon_start;
        return c;
    }
    double x;
    const double y = P.dst_y0 - fi;
    bool right = false;
    if (d.kind == PB_KIND_DOUBLE) {
        right = fj >= (double)P.dst_half_w;
        x = (right ? fj - (double)P.dst_half_w : fj) + P.dst_x0;
        if (right) x = -x;
    } else {
        x = fj + P.dst_x0;
    }
    const double dist = sqrt(x * x + y * y) / d.f_distance;
    double lat = pb_lens_inverse(d.lens, dist);
    if (right) lat = (lat * -1.0) + PB_PI;
    c.lat = lat;
    c.lon = atan2(y, x);
    return c;
}

// One eye of a double-fisheye source seen as a source of its own (plans only; never part of the ABI):
// PB_KIND_EYE_L samples columns [0, w // 2) of the side-by-side frame, PB_KIND_EYE_R columns [w // 2, w),
// mirrored (projection.py:430-431).
#define PB_KIND_EYE_L 3
#define PB_KIND_EYE_R 4

// columns of the frame a source kind may sample: [cmin, cmax)
template <int SRC_KIND>
__device__ __forceinline__ void pb_src_col_range(const PbParams& P, int& cmin, int& cmax) {
    cmin = (SRC_KIND == PB_KIND_EYE_R) ? P.src_eye_w : 0;
    cmax = (SRC_KIND == PB_KIND_EYE_L) ? P.src_eye_w : P.src.width;
}

// pre-truncation source coordinates (row-like, column-like) of a pano / camera source (or one eye)
template <int SRC_KIND>
__device__ __forceinline__ void pb_src_pretrunc(const PbParams& P, const PbCoord& c, double& f0, double& f1) {
    if (SRC_KIND == PB_KIND_PANO) {
        f0 = c.lat / P.src_hseg;
        f1 = c.lon / P.src_wseg + P.src_half_w;
    } else if (SRC_KIND == PB_KIND_EYE_R) {
        // the right eye looks backwards (projection.py:426-427) and is mirrored: sampled column =
        // eye_w + (eye_w_right - 1 - x) with x = trunc(re + cx_r), i.e. floor(w - (re + cx_r)) wherever that
        // is not an exact integer (those pixels end on the plan's fix list like every other model miss)
        const double lat_r = (c.lat * -1.0) + PB_PI;
        const double dist = pb_lens_forward(P.src.lens, lat_r, P.rect_max) * P.src.f_distance;
        double sl, cl;
        sincos(c.lon, &sl, &cl);
        f0 = ((sl * dist) * -1.0) + P.src_cy;
        f1 = (double)P.src.width - ((cl * dist) + P.src_cx_r);
    } else {
        const double dist = pb_lens_forward(P.src.lens, c.lat, P.rect_max) * P.src.f_distance;
        double sl, cl;
        sincos(c.lon, &sl, &cl);
        f0 = ((sl * dist) * -1.0) + P.src_cy;
        f1 = (cl * dist) + P.src_cx;
    }
}

// exact invalid predicate of a camera / double destination pixel as a function of the
// integer n4 = (2x)^2 + (2y)^2 (x*x + y*y == n4 / 4 exactly), projection.py:160, :357-360
__device__ __forceinline__ bool pb_dst_inv_pred(const PbParams& P, long long n4, bool right, bool* outside_domain) {
    const double dist = sqrt((double)n4 * 0.25) / P.dst.f_distance;
    double lat;
    bool nan_region = false;
    switch (P.dst.lens) {
        case PB_LENS_EQUISOLID: {
            const double t = 2.0 * pb_asin_np(dist / 2.0);
            nan_region = (t != t);
            lat = nan_region ? 0.0 : t;
        } break;
        case PB_LENS_ORTHOGRAPHIC: lat = pb_asin_np(dist); nan_region = (lat != lat); break;
        case PB_LENS_THOBY: lat = pb_asin_np(dist / 1.47) / 0.713; nan_region = (lat != lat); break;
        default: lat = pb_lens_inverse(P.dst.lens, dist);
    }
    if (outside_domain) *outside_domain = nan_region;
    if (right) {
        lat = (lat * -1.0) + PB_PI;
        return lat < P.dst_right_min;
    }
    return lat > P.dst_half_fov;
}

// ---- stage B ---------------------------------------------------------------------
__device__ __forceinline__ PbCoord pb_rotate(const double* __restrict__ R, PbCoord c) {
    if (c.inv) {  // rotation.py:125, :168-175
        c.lat = 0.0;
        c.lon = 0.0;
        return c;
    }
    double s, yy, sl, cl;
    s = pb_sin_np(c.lat);           // np.sin(lat), np.cos(lat): two calls in the reference, two functions in libm   rotation.py:129-131
    yy = pb_cos_np(c.lat);
    pb_expi_np(c.lon, &sl, &cl);  // np.exp(lon * 1j)
    const double x = cl * s, z = sl * s;  // rotation.py:130-132
    // accumulation order of the BLAS behind np.matmul (probe)
    const double vx = fma(R[2], z, fma(R[0], x, R[1] * yy));
    const double vy = fma(R[5], z, fma(R[3], x, R[4] * yy));
    const double vz = fma(R[8], z, fma(R[6], x, R[7] * yy));
    c.lat = pb_acos_np(vy);   // rotation.py:158 (NumPy's SIMD arccos, bit for bit: pb_math_np.hpp)
    c.lon = pb_atan2(vz, vx); // rotation.py:159-164
    return c;
}

// The whole rotation chain of one pixel.  ROT = the number of rotations when the caller knows it at COMPILE time (0 or 1: what the
// reference's CLI and nearly every caller use), PB_ROT_ANY otherwise.  History (round 4, experiments/README.md): the float64 remap kernel
// needed 123 VGPRs (4 waves per SIMD) and ran 20-25 % slower on EVERY geometry, rotated or not, because the compiler's machine-level
// loop-invariant code motion hoisted the hundred-odd float64 constants of the five transcendental kernels out of this loop (and out of
// certification's pixel loop) and kept them live across the caller.  The build switches that pass off (build.py): 42-61 VGPRs with the
// plain loop.  Instantiating the three float64 kernels per count on top of that is worth another 3-4 % on unrotated geometries.
#define PB_ROT_ANY (-1)
template <int ROT = PB_ROT_ANY>
__device__ __forceinline__ PbCoord pb_rotate_all(const PbParams& P, PbCoord c) {
    if (ROT == 0) return c;
    if (ROT == 1) return pb_rotate(P.R[0], c);
    for (int k = 0; k < P.n_rot; ++k) c = pb_rotate(P.R[k], c);
    return c;
}
// launches kernel<KIND, ROT> for the plan's rotation count
#define PB_LAUNCH_BY_ROT(n_rot, kernel, KIND, ...)                                      \
    do {                                                                                \
        if ((n_rot) == 0) hipLaunchKernelGGL((kernel<KIND, 0>), __VA_ARGS__);           \
        else if ((n_rot) == 1) hipLaunchKernelGGL((kernel<KIND, 1>), __VA_ARGS__);      \
        else hipLaunchKernelGGL((kernel<KIND, PB_ROT_ANY>), __VA_ARGS__);               \
    } while (0)

// ---- stage C ---------------------------------------------------------------------
// pano source: linear index or -1   (projection.py:533-546)
__device__ __forceinline__ int pb_src_pano_index(const PbParams& P, const PbCoord& c) {
    if (c.inv) return -1;
    const double tr = c.lat / P.src_hseg;
    const double tc = c.lon / P.src_wseg + P.src_half_w;
    const int r = pb_floor_mod(pb_cvt_i64(tr), P.src.height);
    const int col = pb_floor_mod(pb_cvt_i64(tc), P.src.width);
    return r * P.src.width + col;
}

// one fisheye of size h x w centred at (cy, cx): (py, px) or ok == false
// (projection.py:247-260, :223-231), from the distance and the sine / cosine of the longitude (np.exp(lon * 1j), projection.py:252)
__device__ __forceinline__ bool pb_src_camera_pos_sc(double dist, double sl, double cl, int h, int w, double cy, double cx, int& py, int& px) {
    const double re = cl * dist, im = sl * dist;
    const long long y = pb_cvt_i64((im * -1.0) + cy);
    const long long x = pb_cvt_i64(re + cx);
    if (y >= h || y < 0 || x >= w || x < 0) return false;
    py = (int)y;
    px = (int)x;
    return true;
}
__device__ __forceinline__ bool pb_src_camera_pos(const PbParams& P, double lat, double lon, int h, int w, double cy,
                                                  double cx, int& py, int& px) {
    const double dist = pb_lens_forward(P.src.lens, lat, P.rect_max) * P.src.f_distance;
    double sl, cl;
    pb_expi_np(lon, &sl, &cl);  // np.exp(lon * 1j)   projection.py:252
    return pb_src_camera_pos_sc(dist, sl, cl, h, w, cy, cx, py, px);
}

__device__ __forceinline__ int pb_src_camera_index(const PbParams& P, const PbCoord& c) {
    int py, px;
    const bool ok = pb_src_camera_pos(P, c.lat, c.lon, P.src.height, P.src.width, P.src_cy, P.src_cx, py, px);
    return (ok && !c.inv) ? py * P.src.width + px : -1;
}

// the index of ONE source (SRC_KIND camera, or one eye of a double frame: into the full side-by-side frame) from the sine / cosine of
// the longitude: both eyes of a stitch see the same longitude (round 4: its correctly rounded sincos is evaluated once per pixel, not
// once per eye - the float64 chain of c5 spent half its transcendental work on the repeat)
template <int SRC_KIND>
__device__ __forceinline__ int pb_src_index_sc(const PbParams& P, const PbCoord& c, double sl, double cl) {
    int py, px;
    if (SRC_KIND == PB_KIND_EYE_R) {
        const double lat_r = (c.lat * -1.0) + PB_PI;  // projection.py:426-427
        const double dist = pb_lens_forward(P.src.lens, lat_r, P.rect_max) * P.src.f_distance;
        const bool ok = pb_src_camera_pos_sc(dist, sl, cl, P.src.height, P.src_eye_w_right, P.src_cy, P.src_cx_r, py, px);
        // the right eye is mirrored before it is sampled (projection.py:430-431)
        return (ok && !c.inv) ? py * P.src.width + (P.src_eye_w + (P.src_eye_w_right - 1 - px)) : -1;
    }
    const double dist = pb_lens_forward(P.src.lens, c.lat, P.rect_max) * P.src.f_distance;
    const int we = (SRC_KIND == PB_KIND_EYE_L) ? P.src_eye_w : P.src.width;
    const bool ok = pb_src_camera_pos_sc(dist, sl, cl, P.src.height, we, P.src_cy, P.src_cx, py, px);
    return (ok && !c.inv) ? py * P.src.width + px : -1;
}
// ... and its pre-truncation coordinates (what pb_src_pretrunc gives, with the caller's sine / cosine)
template <int SRC_KIND>
__device__ __forceinline__ void pb_src_pretrunc_sc(const PbParams& P, const PbCoord& c, double sl, double cl, double& f0, double& f1) {
    if (SRC_KIND == PB_KIND_EYE_R) {
        const double lat_r = (c.lat * -1.0) + PB_PI;
        const double dist = pb_lens_forward(P.src.lens, lat_r, P.rect_max) * P.src.f_distance;
        f0 = ((sl * dist) * -1.0) + P.src_cy;
        f1 = (double)P.src.width - ((cl * dist) + P.src_cx_r);
    } else {
        const double dist = pb_lens_forward(P.src.lens, c.lat, P.rect_max) * P.src.f_distance;
        f0 = ((sl * dist) * -1.0) + P.src_cy;
        f1 = (cl * dist) + P.src_cx;
    }
}

struct PbDoubleTap {
    int il, ir;     // indices into the full side-by-side frame, or -1
    double fl, fr;  // blend factors
};

__device__ __forceinline__ double pb_merge_factor_of(double mrg_min, double mrg_max_safe, double mrg_max, double mrg_range, double lat) {
    const bool band = (lat >= mrg_min) && (lat <= mrg_max_safe);  // projection.py:440-443
    const double f = (lat - mrg_max) / mrg_range * -1.0;         // projection.py:444
    return band ? f : 1.0;
}
__device__ __forceinline__ double pb_merge_factor(const PbParams& P, double lat) {
    return pb_merge_factor_of(P.mrg_min, P.mrg_max_safe, P.mrg_max, P.mrg_range, lat);
}

__device__ __forceinline__ PbDoubleTap pb_src_double_taps(const PbParams& P, const PbCoord& c) {
    PbDoubleTap t;
    const double lat_r = (c.lat * -1.0) + PB_PI;  // projection.py:426-427
    double sl, cl;
    pb_expi_np(c.lon, &sl, &cl);  // np.exp(lon * 1j): ONE evaluation serves both eyes (same argument, same bits)
    t.il = pb_src_index_sc<PB_KIND_EYE_L>(P, c, sl, cl);
    t.ir = pb_src_index_sc<PB_KIND_EYE_R>(P, c, sl, cl);
    t.fl = pb_merge_factor(P, c.lat);
    t.fr = pb_merge_factor(P, lat_r);
    return t;
}

// (left * fl + right * fr).astype(uint8)   projection.py:447-459
__device__ __forceinline__ unsigned pb_blend_u8(unsigned l, unsigned r, double fl, double fr) {
    return pb_cvt_u8((double)l * fl + (double)r * fr);
}
