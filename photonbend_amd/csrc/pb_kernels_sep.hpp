// pb_kernels_sep.hpp - separable fast path for a DOUBLE-fisheye source seen from an unrotated
// panorama destination (the Gear-360 stitch, BASELINE config c5; projection.py:408-462 fed by :487-513).
//
// With no rotation the destination latitude depends only on the output ROW and the longitude only on
// the COLUMN.  Everything transcendental therefore lives in two small tables built ONCE per plan with
// the faithful device functions (pb_sep_tables_kernel):
//   row table  [H]: dist_left, dist_right (= forward_lens(lat) * f_distance per eye), blend factors
//   col table  [W]: cos(lon), sin(lon)   (the cexp(i*lon) of projection.py:252)
// and a pixel costs four float64 multiplies + four adds + four truncations, in exactly the reference's
// operation order - bit-identical to the faithful kernel by construction (and checked for every pixel at
// plan creation, pb_sep_check_kernel).
#pragma once
#include "pb_kernels_tile.hpp"

struct PbSepRow {
    double dist_l, dist_r, f_l, f_r;
};
struct PbSepCol {
    double cl, sl;
};

__global__ __launch_bounds__(PB_BLOCK) void pb_sep_tables_kernel(const PbParams P, PbSepRow* __restrict__ rows,
                                                                 PbSepCol* __restrict__ cols) {
    const int t = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (t < P.dst.height) {
        const PbCoord c = pb_dst_coord(P, t, 0);
        const double lat_r = (c.lat * -1.0) + PB_PI;  // projection.py:426-427
        PbSepRow r;
        r.dist_l = pb_lens_forward(P.src.lens, c.lat, P.rect_max) * P.src.f_distance;  // projection.py:251
        r.dist_r = pb_lens_forward(P.src.lens, lat_r, P.rect_max) * P.src.f_distance;
        r.f_l = pb_merge_factor(P, c.lat);
        r.f_r = pb_merge_factor(P, lat_r);
        rows[t] = r;
    }
    const int j = t - P.dst.height;
    if (j >= 0 && j < P.dst.width) {
        const PbCoord c = pb_dst_coord(P, 0, j);
        PbSepCol q;
        pb_expi_np(c.lon, &q.sl, &q.cl);
        cols[j] = q;
    }
}

// float64 -> int64 of cvttsd2si restricted to what matters here: anything that is not a finite value
// inside int32 range can never be a valid pixel position
__device__ __forceinline__ bool pb_sep_pos(double pre_y, double pre_x, int h, int w, int& py, int& px) {
    if (!(fabs(pre_y) < 2147483648.0) || !(fabs(pre_x) < 2147483648.0)) return false;
    py = (int)pre_y;
    px = (int)pre_x;
    return py >= 0 && py < h && px >= 0 && px < w;
}

__device__ __forceinline__ void pb_sep_taps(const PbParams& P, const PbSepRow& R, const PbSepCol& C, int& il, int& ir) {
    int py, px;
    // left eye (projection.py:252-259 with the centre of the left half)
    bool ok = pb_sep_pos(((C.sl * R.dist_l) * -1.0) + P.src_cy, (C.cl * R.dist_l) + P.src_cx, P.src.height, P.src_eye_w, py, px);
    il = ok ? py * P.src.width + px : -1;
    // right eye, mirrored before it is sampled (projection.py:430-431)
    ok = pb_sep_pos(((C.sl * R.dist_r) * -1.0) + P.src_cy, (C.cl * R.dist_r) + P.src_cx_r, P.src.height, P.src_eye_w_right, py, px);
    ir = ok ? py * P.src.width + (P.src_eye_w + (P.src_eye_w_right - 1 - px)) : -1;
}

// per-channel (l * fl + r * fr).astype(uint8) on packed RGB
__device__ __forceinline__ unsigned pb_sep_blend(unsigned l, unsigned r, double fl, double fr) {
    if (fl == 1.0 && fr == 1.0) {
        // l * 1.0 + r * 1.0 is the exact integer l + r; astype(uint8) keeps its low 8 bits
        return (((l & 0x00FF00FFu) + (r & 0x00FF00FFu)) & 0x00FF00FFu) | (((l & 0x0000FF00u) + (r & 0x0000FF00u)) & 0x0000FF00u);
    }
    return pb_blend_u8(l & 0xFF, r & 0xFF, fl, fr) | (pb_blend_u8((l >> 8) & 0xFF, (r >> 8) & 0xFF, fl, fr) << 8) |
           (pb_blend_u8((l >> 16) & 0xFF, (r >> 16) & 0xFF, fl, fr) << 16);
}

// one wave per 32x32 tile, lane = 4 consecutive pixels x 4 rows (as the hot kernel's gather phase)
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_sep_double_kernel(const PbParams P, const PbSepRow* __restrict__ rows,
                                                                            const PbSepCol* __restrict__ cols,
                                                                            const uint8_t* __restrict__ src,
                                                                            uint8_t* __restrict__ dst, int n_frames,
                                                                            unsigned long long src_stride,
                                                                            unsigned long long dst_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tx, ty;
    if (!pb_tile_of_wave(P, wave, tx, ty)) return;
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    const int xg = lane & 7, yb = lane >> 3;
    const int W = P.dst.width, H = P.dst.height;
    const int x = X0 + 4 * xg;
    PbSepCol C[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) C[k] = cols[min(x + k, W - 1)];
    int il[4][4], ir[4][4];
    double fl[4], fr[4];
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        const PbSepRow R = rows[min(Y0 + yb + 8 * jr, H - 1)];
        fl[jr] = R.f_l;
        fr[jr] = R.f_r;
#pragma unroll
        for (int k = 0; k < 4; ++k) pb_sep_taps(P, R, C[k], il[jr][k], ir[jr][k]);
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
                const unsigned l = ((unsigned)il[jr][k] == last_px) ? pb_load_px(s, il[jr][k]) : pb_load_px32(s, il[jr][k]);
                const unsigned r = ((unsigned)ir[jr][k] == last_px) ? pb_load_px(s, ir[jr][k]) : pb_load_px32(s, ir[jr][k]);
                a[k] = pb_sep_blend(l, r, fl[jr], fr[jr]);  // a pano destination has no invalid pixels
            }
            if (y < H) {
                const unsigned long long off = 3ull * ((unsigned long long)y * W + x);
                if (x + 3 < W && (((uintptr_t)d + off) & 3u) == 0) {
                    __builtin_nontemporal_store(pb_pack_px4(a[0], a[1], a[2], a[3]), reinterpret_cast<pb_u32x3*>(d + off));
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

// plan creation: the separable taps / factors against the faithful chain, for every pixel
__global__ __launch_bounds__(PB_BLOCK) void pb_sep_check_kernel(const PbParams P, const PbSepRow* __restrict__ rows,
                                                                const PbSepCol* __restrict__ cols,
                                                                unsigned* __restrict__ mismatches) {
    const unsigned total = (unsigned)P.dst.height * (unsigned)P.dst.width;
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    const unsigned i = p / (unsigned)P.dst.width, j = p - i * (unsigned)P.dst.width;
    PbCoord c = pb_dst_coord(P, (int)i, (int)j);
    const PbDoubleTap t = pb_src_double_taps(P, c);
    int il, ir;
    const PbSepRow R = rows[i];
    pb_sep_taps(P, R, cols[j], il, ir);
    const bool same = il == t.il && ir == t.ir && (R.f_l == t.fl || (R.f_l != R.f_l && t.fl != t.fl)) &&
                      (R.f_r == t.fr || (R.f_r != R.f_r && t.fr != t.fr));
    if (!same) atomicAdd(mismatches, 1u);
}
