// pb_kernels_bilinear.hpp - OPT-IN bilinear sampling.
//
// The reference samples nearest-by-truncation only (projection.py:254-259, :545); this mode has NO reference
// behaviour to be at parity with ("parity unpinned": its oracle is oracle/reference_path.py:remap_bilinear,
// our own definition).  Definition: the continuous source coordinate is the reference's PRE-truncation
// coordinate f (pixel k covers [k, k+1), centre k + 0.5).  s = f - 0.5, i0 = floor(s), t = s - i0; the four taps
// (i0, i0 + 1) x (j0, j0 + 1) are clamped to the image (pano columns wrap), weights (1-t, t); the channel value is
// rounded half-to-even.  Pixels the nearest mode paints black (invalid destination pixel, camera source position
// outside [0, h) x [0, w)) stay black.  A double-fisheye source is the reference's own blend (projection.py:439-460) of
// the two eyes' BILINEAR samples: each eye is sampled like a camera source on its half of the frame (the right eye on
// the mirrored half, taps clamped to the eye), rounded to uint8, then (l * fl + r * fr).astype(uint8) with the
// reference's factors.
//
// Round 4: ONE launch per call, no float64 in it.  A tile is served either by its float32 tile model (certified to
// 1/1024 px against the faithful coordinate, pb_certify_kernel) or - failed tiles, coarse models, tiles with invalid
// pixels, an image edge or an eye's rim inside - by the plan's EXACT COORDINATE TABLE: the faithful tap coordinate of each of
// its pixels in 1/4096 px, 8 KiB per tile, built once per plan from the float64 chain (like the nearest mode's exact-index
// tables: what the models cannot reproduce is looked up, not recomputed).  Round 3 recomputed those tiles per frame in
// float64 (c3: 1 506 of 16 384 tiles, 61 us of a 115 us frame; c5: 63 us).
//
// Round 5: tile models evaluated on their certified low-degree part (PB_TILE_TD3); table tiles walked along the line of
// constant source row (pb_bilinear_orient_kernel) and read without guards where their taps allow (PB_TILE_TAB_PLAIN); direct-gather
// tiles staged as two half windows (PB_TILE_HALVES); an LDS pool per workgroup (pb_bilinear_pool_kernel); two-eye slots that carry
// the left entry (PB_TILE_TWO); one instance of the tile code for both eyes.  None of these moves a bit of the output
// (tests/test_hip_bilinear_invariance.py).
//
//   pb_bilinear_hot_kernel          pano / camera sources: one wave per tile over the plan's launch-order table
//   pb_bilinear_double_hot_kernel   double-fisheye sources: one-eye (SOLO) tiles through the same tile code, two-eye tiles
//                                   sample both eyes and blend with the tile's weight class
//   pb_bilinear_fix_kernel, pb_bilinear_double_kernel, pb_bilinear_double_fix_kernel
//                                   the float64 chain per pixel: the mode's definition on the device (PB_MODE_FAITHFUL, plans
//                                   without tile tables) and the fallback for plans whose coordinate table would not fit
#pragma once
#include "pb_kernels_tile.hpp"

#define PB_BIL_WPE 3           // waves per SIMD the bilinear tile kernels are compiled for (the register budget: 512 / PB_BIL_WPE VGPRs)
#define PB_BIL_WPE_DBL 4       // ... and the double-fisheye kernel, whose pair layout leaves every wave a single-source tile's state (128 VGPRs)
#define PB_BIL_HALVES_MAX 24576  // bytes both half windows of a PB_TILE_HALVES tile may have in sum

// ---- float64 passes: tap arithmetic ------------------------------------------------------------------------------------------
template <int SRC_KIND>
__device__ __forceinline__ unsigned pb_bilinear_taps(const PbParams& P, const uint8_t* __restrict__ s, float sy, float sx, int by,
                                                     int bx) {
    // sy / sx: s = f - 0.5 relative to the integer bases (by, bx); all taps are clamped / wrapped into the image
    const int h = P.src.height, w = P.src.width;
    const float fy0 = floorf(sy), fx0 = floorf(sx);
    const float ty = sy - fy0, tx = sx - fx0;
    int r0 = by + (int)fy0, c0 = bx + (int)fx0;
    int r1 = r0 + 1, c1 = c0 + 1;
    r0 = min(max(r0, 0), h - 1);
    r1 = min(max(r1, 0), h - 1);
    if (SRC_KIND == PB_KIND_PANO) {
        c0 = c0 < 0 ? c0 + w : (c0 >= w ? c0 - w : c0);
        c1 = c1 < 0 ? c1 + w : (c1 >= w ? c1 - w : c1);
        c0 = min(max(c0, 0), w - 1);
        c1 = min(max(c1, 0), w - 1);
    } else {
        c0 = min(max(c0, 0), w - 1);
        c1 = min(max(c1, 0), w - 1);
    }
    // byte loads: measured faster here than guarded unaligned dword loads (c3 199 vs 327 us)
    const unsigned p00 = pb_load_px(s, r0 * w + c0), p01 = pb_load_px(s, r0 * w + c1);
    const unsigned p10 = pb_load_px(s, r1 * w + c0), p11 = pb_load_px(s, r1 * w + c1);
    unsigned out = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float a = (float)((p00 >> (8 * ch)) & 0xFF), b = (float)((p01 >> (8 * ch)) & 0xFF);
        const float c = (float)((p10 >> (8 * ch)) & 0xFF), d = (float)((p11 >> (8 * ch)) & 0xFF);
        const float top = fmaf(tx, b - a, a), bot = fmaf(tx, d - c, c);
        const float v = fmaf(ty, bot - top, top);
        out |= ((unsigned)(int)rintf(fminf(fmaxf(v, 0.0f), 255.0f))) << (8 * ch);
    }
    return out;
}

// ---- the mode on a MATERIALISED map, any image (round 5: f-4's API complete) ----------------------------------------------------
// pb_sample_map_bilinear_px: the definition itself, per pixel, in float64 - the coordinate comes from the caller's (H, W, 3) map (a map
// that was looked at or edited between the stages; a destination Lens of user callables), the source stage is the reference's
// (projection.py:247-260, :533-545; a source Lens of user callables through the host-evaluated distance planes, like
// pb_index_from_map_i32), the image any layout the reference's fancy indexing accepts: C channels of 8- or 16-bit samples.  The three
// lerps are the oracle's float64 expressions with the build's -ffp-contract=off: on equal coordinates the bytes ARE
// oracle.remap_bilinear's.  One work-item per output pixel; off the hot path (uint8 RGB + built-in lenses + a lazy map take the tile
// kernels above).
template <typename SAMPLE>
__device__ __forceinline__ double pb_bil64_tap(const SAMPLE* __restrict__ img, long long r, long long c, int w, int channels, int ch) {
    return (double)img[((unsigned long long)r * (unsigned)w + (unsigned long long)c) * (unsigned)channels + (unsigned)ch];
}
// one source's sample of channel ch at pre-truncation coordinate (fy, fx): taps clamped to rows [0, h) and columns [cmin, cmax) of a
// frame `w` wide (`mirror`: the right eye's image is its half mirrored - projection.py:430-431 - eye column x = frame column
// cmax - 1 - x); WRAP: a panorama's columns wrap first.  Rounded half to even and clamped to the sample type's range.
template <typename SAMPLE, bool WRAP>
__device__ __forceinline__ double pb_bil64_sample(const SAMPLE* __restrict__ img, double fy, double fx, int h, int w, int cmin, int cmax, bool mirror,
                                                  int channels, int ch) {
    const double sy = fy - 0.5, sx = fx - 0.5;
    const double ry = floor(sy), rx = floor(sx);
    const double ty = sy - ry, tx = sx - rx;
    long long r0 = (long long)ry, c0 = (long long)rx;
    long long r1 = r0 + 1, c1 = c0 + 1;
    r0 = r0 < 0 ? 0 : (r0 > h - 1 ? h - 1 : r0);
    r1 = r1 < 0 ? 0 : (r1 > h - 1 ? h - 1 : r1);
    const long long we = cmax - cmin;
    if (WRAP) {
        c0 %= we; if (c0 < 0) c0 += we;
        c1 %= we; if (c1 < 0) c1 += we;
    }
    c0 = c0 < 0 ? 0 : (c0 > we - 1 ? we - 1 : c0);
    c1 = c1 < 0 ? 0 : (c1 > we - 1 ? we - 1 : c1);
    const long long g0 = mirror ? (cmax - 1 - c0) : (cmin + c0), g1 = mirror ? (cmax - 1 - c1) : (cmin + c1);
    const double a = pb_bil64_tap(img, r0, g0, w, channels, ch), b = pb_bil64_tap(img, r0, g1, w, channels, ch);
    const double c = pb_bil64_tap(img, r1, g0, w, channels, ch), d = pb_bil64_tap(img, r1, g1, w, channels, ch);
    const double top = a + tx * (b - a), bot = c + tx * (d - c);
    const double v = rint(top + ty * (bot - top));
    const double vmax = (double)(SAMPLE)~(SAMPLE)0;
    return v < 0.0 ? 0.0 : (v > vmax ? vmax : v);
}
template <int SRC_KIND, typename SAMPLE>
__global__ __launch_bounds__(PB_BLOCK) void pb_sample_map_bilinear_kernel(const PbParams P, double* __restrict__ map, unsigned total,
                                                                          const double* __restrict__ dist_l, const double* __restrict__ dist_r,
                                                                          const SAMPLE* __restrict__ img, void* __restrict__ out, int channels) {
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    double* a = map + 3ull * p;
    const bool inv = a[2] != 0.0;
    if (SRC_KIND == PB_KIND_PANO && inv) {
        a[0] = 0.0;  // polar_map[invalid_map] = 0 writes through the view, projection.py:534-536
        a[1] = 0.0;
    }
    const double lat = a[0], lon = a[1];
    const int h = P.src.height, w = P.src.width;
    if (SRC_KIND == PB_KIND_PANO) {
        const double fy = lat / P.src_hseg, fx = lon / P.src_wseg + P.src_half_w;
        const bool live = !inv && fabs(fy) < 1.0e300 && fabs(fx) < 1.0e300 && fy == fy && fx == fx;
        SAMPLE* o = static_cast<SAMPLE*>(out) + (unsigned long long)p * (unsigned)channels;
        for (int ch = 0; ch < channels; ++ch) o[ch] = live ? (SAMPLE)pb_bil64_sample<SAMPLE, true>(img, fy, fx, h, w, 0, w, false, channels, ch) : (SAMPLE)0;
        return;
    }
    double sl, cl;
    pb_expi_np(lon, &sl, &cl);  // np.exp(lon * 1j), projection.py:252
    if (SRC_KIND == PB_KIND_CAMERA) {
        const double dist = dist_l ? dist_l[p] : pb_lens_forward(P.src.lens, lat, P.rect_max) * P.src.f_distance;
        const double fy = ((sl * dist) * -1.0) + P.src_cy, fx = (cl * dist) + P.src_cx;
        const bool live = !inv && fy == fy && fx == fx && fabs(fy) < 1.0e300 && fabs(fx) < 1.0e300 && fy >= 0.0 && fy < (double)h && fx >= 0.0 && fx < (double)w;
        SAMPLE* o = static_cast<SAMPLE*>(out) + (unsigned long long)p * (unsigned)channels;
        for (int ch = 0; ch < channels; ++ch) o[ch] = live ? (SAMPLE)pb_bil64_sample<SAMPLE, false>(img, fy, fx, h, w, 0, w, false, channels, ch) : (SAMPLE)0;
        return;
    }
    // two eyes (projection.py:408-462): each sampled like a camera source on its half (the right one mirrored), then the reference's blend
    const double lat_r = (lat * -1.0) + PB_PI;
    const double dl = dist_l ? dist_l[p] : pb_lens_forward(P.src.lens, lat, P.rect_max) * P.src.f_distance;
    const double dr = dist_r ? dist_r[p] : pb_lens_forward(P.src.lens, lat_r, P.rect_max) * P.src.f_distance;
    const int wl = P.src_eye_w, wr = P.src_eye_w_right;
    const double fyl = ((sl * dl) * -1.0) + P.src_cy, fxl = (cl * dl) + P.src_cx;
    const double fyr = ((sl * dr) * -1.0) + P.src_cy, fxr = (cl * dr) + P.src_cx_r;
    const bool live_l = !inv && fyl == fyl && fxl == fxl && fabs(fyl) < 1.0e300 && fabs(fxl) < 1.0e300 && fyl >= 0.0 && fyl < (double)h && fxl >= 0.0 && fxl < (double)wl;
    const bool live_r = !inv && fyr == fyr && fxr == fxr && fabs(fyr) < 1.0e300 && fabs(fxr) < 1.0e300 && fyr >= 0.0 && fyr < (double)h && fxr >= 0.0 && fxr < (double)wr;
    const double fl = pb_merge_factor(P, lat), fr = pb_merge_factor(P, lat_r);
    uint8_t* o = static_cast<uint8_t*>(out) + (unsigned long long)p * (unsigned)channels;  // (left * fl + right * fr).astype(np.uint8): uint8 whatever the samples
    for (int ch = 0; ch < channels; ++ch) {
        const double l = live_l ? pb_bil64_sample<SAMPLE, false>(img, fyl, fxl, h, w, 0, wl, false, channels, ch) : 0.0;
        const double r = live_r ? pb_bil64_sample<SAMPLE, false>(img, fyr, fxr, h, w, wl, wl + wr, true, channels, ch) : 0.0;
        o[ch] = inv ? (uint8_t)0 : (uint8_t)pb_cvt_u8(l * fl + r * fr);
    }
}

// ---- exact coordinate tables ------------------------------------------------------------------------------------------------
// The faithful tap coordinate s = f - 0.5 of one pixel for one source (or eye), in 1/4096 px, in FRAME space: the right eye's
// column runs over the mirrored half (w - 1 - s_eye: bilinear interpolation commutes with the mirror, the taps are the frame's
// texels floor(s), floor(s) + 1 of either eye, clamped to the eye's own columns).  y == PB_BIL_DEAD: the pixel is black for this
// source (invalid destination pixel, position outside the image / the eye, not finite).  2^-13 px of quantisation = 0.06 LSB on
// the steepest possible content.
struct PbBilCoord {
    int32_t y, x;
};
// PbTileEntry::bil_off of a tile with a slot: bits 0-19 the slot, bits 20-27 the walk's shear (signed, 1/64 pixel per pixel; the entry's
// flag PB_TILE_TAB_Y says which way the walk runs); -1 = no slot
#define PB_BIL_SLOT_MASK 0xFFFFF
__host__ __device__ __forceinline__ int pb_bil_slot_shift(int bil_off, int p) {
    const int q = (int)(signed char)((bil_off >> 20) & 0xFF);
    return (int)rintf((float)q * (1.0f / 64.0f) * ((float)p - 15.5f));
}
#define PB_BIL_SHIFT 12
#define PB_BIL_DEAD ((int32_t)0x80000000)
#define PB_BIL_MAX_DIM (1 << 18)  // source sides the fixed point holds (the plan keeps the float64 pass beyond)

template <int SRC_KIND>
__device__ __forceinline__ PbBilCoord pb_bil_coord_of(const PbParams& P, const PbCoord& c) {
    PbBilCoord q = {PB_BIL_DEAD, 0};
    if (c.inv) return q;
    double f0, f1;
    bool live;
    if (SRC_KIND == PB_KIND_PANO) {
        pb_src_pretrunc<PB_KIND_PANO>(P, c, f0, f1);
        live = f0 == f0 && f1 == f1 && fabs(f0) < 1.0e9 && fabs(f1) < 1.0e9;
    } else {
        // one fisheye (or one eye, sampled like a camera source on its half: projection.py:429-434) in ITS pixel space
        const double lat = (SRC_KIND == PB_KIND_EYE_R) ? (c.lat * -1.0) + PB_PI : c.lat;  // projection.py:426-427
        const int we = (SRC_KIND == PB_KIND_EYE_L) ? P.src_eye_w : (SRC_KIND == PB_KIND_EYE_R) ? P.src_eye_w_right : P.src.width;
        const double cx = (SRC_KIND == PB_KIND_EYE_R) ? P.src_cx_r : P.src_cx;
        const double dist = pb_lens_forward(P.src.lens, lat, P.rect_max) * P.src.f_distance;
        double sl, cl;
        pb_expi_np(c.lon, &sl, &cl);  // np.exp(lon * 1j)
        f0 = ((sl * dist) * -1.0) + P.src_cy;
        f1 = (cl * dist) + cx;
        live = f0 == f0 && f1 == f1 && fabs(f0) < 1.0e9 && fabs(f1) < 1.0e9 && f0 >= 0.0 && f0 < (double)P.src.height && f1 >= 0.0 && f1 < (double)we;
        if (SRC_KIND == PB_KIND_EYE_R) f1 = (double)P.src.width - f1;  // the mirrored half of the frame (projection.py:430-431)
    }
    if (!live) return q;
    q.y = (int32_t)rint((f0 - 0.5) * (double)(1 << PB_BIL_SHIFT));
    q.x = (int32_t)rint((f1 - 0.5) * (double)(1 << PB_BIL_SHIFT));
    return q;
}

// ---- the arithmetic of one pixel -------------------------------------------------------------------------------------------
// Two horizontally adjacent taps are 6 consecutive bytes; (lo, hi) = the 8 bytes from the left tap's first byte on.  The four weights
// (1-tx)(1-ty), tx(1-ty), (1-tx)ty, tx ty are formed once per pixel in float32.  (The definition nests three lerps in float64.)
//
// Round 6: the weighted sum runs in INTEGERS.  The four weights, formed in float32 as before, become 16-bit fixed point (v_cvt_pknorm_u16_f32
// packs two per instruction; the factor 65536 / 65535 rides on the (1 - ty, ty) pair so that a weight w comes out as round(65536 w)),
// a channel's two taps of a row are lifted into one dword of two 16-bit lanes by ONE v_perm_b32 straight from the 8 bytes, and
// v_dot2_u32_u16 takes a row's two products and the running sum in one instruction: per pixel 6 v_perm + 6 v_dot2 + 2 v_perm to pack
// where the float32 form took 12 byte-to-float conversions, 6 packed multiply-adds and 3 pack-converts - a quarter of a window tile's
// vector instructions, and the tile code is bound by vector issue.  The sum is exact in 32 bits (255 x 65538 < 2^24);
// +0.5 rides in the accumulator, the channel is bits 16-23.  Against the float32 form: the weights carry 16 bits instead of 24 - a value
// error of at most 0.008 LSB before rounding (typically 0.001) - and an exact tie rounds up instead of to even.
typedef unsigned short pb_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pb_udot2(unsigned a, unsigned b, unsigned c) {
    pb_us2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    return __builtin_amdgcn_udot2(x, y, c, false);
}
__device__ __forceinline__ unsigned pb_pknorm_u16(float lo, float hi) {
    const pb_us2 w = __builtin_amdgcn_cvt_pknorm_u16(lo, hi);
    unsigned u;
    __builtin_memcpy(&u, &w, 4);
    return u;
}
#define PB_BIL_WSCALE (65536.0f / 65535.0f)
// one pixel: (lo, hi) of the two rows, W0 = (w00, w01), W1 = (w10, w11) as 16-bit fixed point
__device__ __forceinline__ unsigned pb_bil_dot(unsigned lo0, unsigned hi0, unsigned lo1, unsigned hi1, unsigned W0, unsigned W1) {
    // v_perm_b32(S0, S1, sel): selector 0-3 = bytes of S1, 4-7 = bytes of S0, 0x0c = 0x00: (tap a's byte, 0, tap b's byte, 0)
    const unsigned s0 = pb_udot2(__builtin_amdgcn_perm(hi1, lo1, 0x0c030c00u), W1, pb_udot2(__builtin_amdgcn_perm(hi0, lo0, 0x0c030c00u), W0, 0x8000u));
    const unsigned s1 = pb_udot2(__builtin_amdgcn_perm(hi1, lo1, 0x0c040c01u), W1, pb_udot2(__builtin_amdgcn_perm(hi0, lo0, 0x0c040c01u), W0, 0x8000u));
    const unsigned s2 = pb_udot2(__builtin_amdgcn_perm(hi1, lo1, 0x0c050c02u), W1, pb_udot2(__builtin_amdgcn_perm(hi0, lo0, 0x0c050c02u), W0, 0x8000u));
    return __builtin_amdgcn_perm(s2, __builtin_amdgcn_perm(s1, s0, 0x0c0c0602u), 0x0c060100u);  // bytes 2 of the three sums
}
__device__ __forceinline__ unsigned pb_bil_mix64(unsigned lo0, unsigned hi0, unsigned lo1, unsigned hi1, float tx, float ty) {
    const float ux = 1.0f - tx, uy = fmaf(-ty, PB_BIL_WSCALE, PB_BIL_WSCALE), vy = ty * PB_BIL_WSCALE;
    return pb_bil_dot(lo0, hi0, lo1, hi1, pb_pknorm_u16(ux * uy, tx * uy), pb_pknorm_u16(ux * vy, tx * vy));
}
// Two pixels at once: their weights are formed packed (v_pk_*).
__device__ __forceinline__ void pb_bil_mix64x2(const unsigned lo0[2], const unsigned hi0[2], const unsigned lo1[2], const unsigned hi1[2], const pb_f2 tx,
                                               const pb_f2 ty, unsigned out[2]) {
    const pb_f2 one = {1.0f, 1.0f}, k = {PB_BIL_WSCALE, PB_BIL_WSCALE};
    const pb_f2 ux = one - tx, uy = __builtin_elementwise_fma(-ty, k, k), vy = ty * k;
    const pb_f2 w00 = ux * uy, w01 = tx * uy, w10 = ux * vy, w11 = tx * vy;
#pragma unroll
    for (int i = 0; i < 2; ++i) out[i] = pb_bil_dot(lo0[i], hi0[i], lo1[i], hi1[i], pb_pknorm_u16(w00[i], w01[i]), pb_pknorm_u16(w10[i], w11[i]));
}
// the same from four separate taps (low 3 bytes of each)
__device__ __forceinline__ unsigned pb_bil_mix4(unsigned p00, unsigned p01, unsigned p10, unsigned p11, float tx, float ty) {
    return pb_bil_mix64((p00 & 0xFFFFFFu) | (p01 << 24), p01 >> 8, (p10 & 0xFFFFFFu) | (p11 << 24), p11 >> 8, tx, ty);
}

// a * b + c on 24-bit operands in ONE instruction (the compiler forms v_mul_u32_u24 + v_add3_u32 from __umul24(a, b) + c)
__device__ __forceinline__ unsigned pb_umad24(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// The tile model as the bilinear tile code evaluates it: TD3 = the terms of total degree <= 3 only (PB_TILE_TD3 tiles: certified within
// PB_COARSE_PX of the faithful coordinate by themselves), 4.5 packed multiply-adds per pixel instead of 9.
template <bool TD3>
__device__ __forceinline__ void pb_bil_collapse(const PbTileEntry* __restrict__ e, const bool col, const int t, pb_f2 c[5]) {
    if (TD3) {
        if (col) pb_collapse_col_td3(e, t, c);
        else pb_collapse_row_td3(e, t, c);
    } else {
        if (col) pb_collapse_col(e, t, c);
        else pb_collapse_row(e, t, c);
    }
}
template <bool TD3>
__device__ __forceinline__ pb_f2 pb_bil_eval(const pb_f2 c[5], const float t) {
    return TD3 ? pb_eval_row_td3(c, t) : pb_eval_row(c, t);
}

// Four pixels from the wave's LDS window: the four taps around s = f - 0.5 (WINDOW coordinates; LEAN tiles carry one texel of margin
// on every side - exactly the taps' reach - and lie strictly inside the image, so nothing is clamped or wrapped, and s >= 0.5:
// truncation is floor).  Three aligned dwords per row (ds_read2_b32 + ds_read_b32) and two v_alignbyte give the 8 bytes from the left
// tap on (an unaligned ds_read_b64 is legal here but measured three times slower: experiments/r4/exp_isa.hip).  Four pixels per call:
// their 24 LDS reads are in flight together and their arithmetic interleaves - one pixel at a time left a wave waiting for LDS sixteen
// times per tile, with a wait state after every dependent v_pk_fma of the coordinate polynomial (c5: a wave lived 19 us).
// a0w: the byte offset of the window's first sample from the START OF LDS (the wave's window offset folded in: a multiple of 16, so
// the byte phase of an address is the sample's) - an address is two multiply-adds and one AND.
typedef const __attribute__((address_space(3))) unsigned* pb_lds_cptr;
struct PbLdsTaps {  // four pixels' taps on their way out of LDS: the dwords, the byte phases, the weights
    unsigned l0[4], w[4][6];
    float tx[4], ty[4];
};
__device__ __forceinline__ void pb_bil_lds4_issue(const pb_f2 sv[4], unsigned pitch, unsigned a0w, PbLdsTaps& T) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        T.ty[k] = __builtin_amdgcn_fractf(sv[k].x);
        T.tx[k] = __builtin_amdgcn_fractf(sv[k].y);
        T.l0[k] = pb_umad24((unsigned)(int)sv[k].x, pitch, pb_umad24((unsigned)(int)sv[k].y, 3u, a0w));
        const unsigned b0 = T.l0[k] & ~3u;  // (the pitch is a multiple of 16: both rows share the byte phase)
        const pb_lds_cptr r0 = (pb_lds_cptr)(uintptr_t)b0, r1 = (pb_lds_cptr)(uintptr_t)(b0 + pitch);
        T.w[k][0] = r0[0];
        T.w[k][1] = r0[1];
        T.w[k][2] = r0[2];
        T.w[k][3] = r1[0];
        T.w[k][4] = r1[1];
        T.w[k][5] = r1[2];
    }
}
__device__ __forceinline__ void pb_bil_lds4_blend(const PbLdsTaps& T, unsigned out[4]) {
#pragma unroll
    for (int k = 0; k < 4; k += 2) {
        unsigned lo0[2], hi0[2], lo1[2], hi1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lo0[i] = __builtin_amdgcn_alignbyte(T.w[k + i][1], T.w[k + i][0], T.l0[k + i]);
            hi0[i] = __builtin_amdgcn_alignbyte(T.w[k + i][2], T.w[k + i][1], T.l0[k + i]);
            lo1[i] = __builtin_amdgcn_alignbyte(T.w[k + i][4], T.w[k + i][3], T.l0[k + i]);
            hi1[i] = __builtin_amdgcn_alignbyte(T.w[k + i][5], T.w[k + i][4], T.l0[k + i]);
        }
        const pb_f2 tx2 = {T.tx[k], T.tx[k + 1]}, ty2 = {T.ty[k], T.ty[k + 1]};
        pb_bil_mix64x2(lo0, hi0, lo1, hi1, tx2, ty2, &out[k]);
    }
}
__device__ __forceinline__ void pb_bil_lds4(const pb_f2 sv[4], unsigned pitch, unsigned a0w, unsigned out[4]) {
    PbLdsTaps T;
    pb_bil_lds4_issue(sv, pitch, a0w, T);
    pb_bil_lds4_blend(T, out);
}

// the three bytes of a pixel at byte offset o of the frame as the low bytes of a dword; the frame's very last pixel is read one byte
// early and shifted (a dword at its address would touch one byte past the buffer).  Needs frame_bytes >= 4 (host check).
__device__ __forceinline__ unsigned pb_bil_load_px(const uint8_t* __restrict__ s, unsigned o, unsigned frame_bytes) {
    const unsigned oo = min(o, frame_bytes - 4u);
    unsigned t;
    __builtin_memcpy(&t, s + oo, 4);
    return t >> (8u * (o - oo));
}

// one pixel from its exact tap coordinate (fix pixels): taps clamped to the frame's rows and to the columns [cmin, cmax) of the
// source (an eye's half), a panorama's columns wrap
template <bool WRAP>
__device__ __forceinline__ unsigned pb_bil_table_px(const uint8_t* __restrict__ s, int qy, int qx, int h, int w, int cmin, int cmax,
                                                    unsigned frame_bytes) {
    const bool dead = qy == PB_BIL_DEAD;
    if (dead) qy = 0;
    const float ty = (float)(qy & ((1 << PB_BIL_SHIFT) - 1)) * (1.0f / (float)(1 << PB_BIL_SHIFT));
    const float tx = (float)(qx & ((1 << PB_BIL_SHIFT) - 1)) * (1.0f / (float)(1 << PB_BIL_SHIFT));
    int r0 = qy >> PB_BIL_SHIFT, c0 = qx >> PB_BIL_SHIFT;  // arithmetic shifts: floor
    int r1 = r0 + 1, c1 = c0 + 1;
    r0 = min(max(r0, 0), h - 1);
    r1 = min(max(r1, 0), h - 1);
    if (WRAP) {
        c0 = c0 < 0 ? c0 + w : (c0 >= w ? c0 - w : c0);
        c1 = c1 < 0 ? c1 + w : (c1 >= w ? c1 - w : c1);
    }
    c0 = min(max(c0, cmin), cmax - 1);
    c1 = min(max(c1, cmin), cmax - 1);
    const unsigned b0 = (unsigned)r0 * (unsigned)w, b1 = (unsigned)r1 * (unsigned)w;
    const unsigned p00 = pb_bil_load_px(s, 3u * (b0 + (unsigned)c0), frame_bytes), p01 = pb_bil_load_px(s, 3u * (b0 + (unsigned)c1), frame_bytes);
    const unsigned p10 = pb_bil_load_px(s, 3u * (b1 + (unsigned)c0), frame_bytes), p11 = pb_bil_load_px(s, 3u * (b1 + (unsigned)c1), frame_bytes);
    const unsigned out = pb_bil_mix4(p00, p01, p10, p11, tx, ty);
    return dead ? 0u : out;
}

// The table path of a tile: the lane's 16 pixels from their exact tap coordinates, eight at a time.  A pixel's two taps of a row
// are 6 consecutive bytes wherever the right tap is the left one's neighbour: ONE 8-byte load per row (moved back and shifted where
// it would run past the frame's end).  A right tap clamped onto the left one (an image / eye edge) is that same tap again; a
// panorama's wrapped right tap (column 0 after column w - 1) is loaded by itself, the only divergent case.  Needs frame_bytes >= 8.
struct PbBilTap {
    unsigned o0, o1;  // byte offsets of the left taps of the two rows
    unsigned c1off;   // WRAP only: byte offset of the right tap's column relative to the left one's row start, or ~0u: neighbour / same
    float tx, ty;
    int kind;         // 0: neighbours, 1: the right tap is the left tap, 2: wrapped, -1: black
};
template <bool WRAP, bool PLAIN>
__device__ __forceinline__ PbBilTap pb_bil_table_tap(int qy, int qx, int h, int w, int cmin, int cmax) {
    PbBilTap t;
    const bool dead = qy == PB_BIL_DEAD;
    if (dead) qy = 0;
    if (PLAIN) {  // (PB_TILE_TAB_PLAIN: the four taps of every live pixel are inside the frame and the column range)
        if (dead) qx = 0;
        t.ty = (float)(qy & ((1 << PB_BIL_SHIFT) - 1)) * (1.0f / (float)(1 << PB_BIL_SHIFT));
        t.tx = (float)(qx & ((1 << PB_BIL_SHIFT) - 1)) * (1.0f / (float)(1 << PB_BIL_SHIFT));
        // (rows clamp to the image like the definition's - the rim of a fisheye output samples a panorama's last row -, columns do not)
        const int r0 = qy >> PB_BIL_SHIFT;
        const unsigned cb = __umul24((unsigned)(qx >> PB_BIL_SHIFT), 3u);
        t.o0 = __umul24((unsigned)max(r0, 0), 3u * (unsigned)w) + cb;
        t.o1 = __umul24((unsigned)min(r0 + 1, h - 1), 3u * (unsigned)w) + cb;
        t.kind = dead ? -1 : 0;
        t.c1off = 3u;
        return t;
    }
    t.ty = (float)(qy & ((1 << PB_BIL_SHIFT) - 1)) * (1.0f / (float)(1 << PB_BIL_SHIFT));
    t.tx = (float)(qx & ((1 << PB_BIL_SHIFT) - 1)) * (1.0f / (float)(1 << PB_BIL_SHIFT));
    int r0 = qy >> PB_BIL_SHIFT, c0 = qx >> PB_BIL_SHIFT;
    int r1 = r0 + 1, c1 = c0 + 1;
    r0 = min(max(r0, 0), h - 1);
    r1 = min(max(r1, 0), h - 1);
    if (WRAP) {
        c0 = c0 < 0 ? c0 + w : (c0 >= w ? c0 - w : c0);
        c1 = c1 < 0 ? c1 + w : (c1 >= w ? c1 - w : c1);
    }
    c0 = min(max(c0, cmin), cmax - 1);
    c1 = min(max(c1, cmin), cmax - 1);
    t.o0 = 3u * ((unsigned)r0 * (unsigned)w + (unsigned)c0);
    t.o1 = 3u * ((unsigned)r1 * (unsigned)w + (unsigned)c0);
    t.kind = dead ? -1 : (c1 == c0 + 1 ? 0 : (c1 == c0 ? 1 : 2));
    t.c1off = 3u * (unsigned)(c1 - c0);  // (kind 2: negative, as an unsigned wrap-around - added to o0 / o1)
    return t;
}
// 8 bytes at byte offset o of the frame (moved back and shifted at the frame's end)
__device__ __forceinline__ unsigned long long pb_bil_load8(const uint8_t* __restrict__ s, unsigned o, unsigned frame_bytes) {
    const unsigned oo = min(o, frame_bytes - 8u);
    unsigned long long t;
    __builtin_memcpy(&t, s + oo, 8);
    return t >> (8u * (o - oo));
}
template <bool WRAP, bool PLAIN>
__device__ __forceinline__ void pb_bil_table8(const uint8_t* __restrict__ s, const int4 q[4], int h, int w, int cmin, int cmax, unsigned frame_bytes,
                                              unsigned out[8]) {
    PbBilTap t[8];
    unsigned long long r0[8], r1[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int4 qq = q[n >> 1];
        t[n] = pb_bil_table_tap<WRAP, PLAIN>((n & 1) ? qq.z : qq.x, (n & 1) ? qq.w : qq.y, h, w, cmin, cmax);
        if (PLAIN) {  // (both 8-byte loads end inside the buffer: checked at plan time)
            __builtin_memcpy(&r0[n], s + t[n].o0, 8);
            __builtin_memcpy(&r1[n], s + t[n].o1, 8);
        } else {
            r0[n] = pb_bil_load8(s, t[n].o0, frame_bytes);
            r1[n] = pb_bil_load8(s, t[n].o1, frame_bytes);
        }
    }
#pragma unroll
    for (int n = 0; n < 8; n += 2) {
        unsigned lo0[2], hi0[2], lo1[2], hi1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const PbBilTap& tt = t[n + i];
            lo0[i] = (unsigned)r0[n + i];
            hi0[i] = (unsigned)(r0[n + i] >> 32);
            lo1[i] = (unsigned)r1[n + i];
            hi1[i] = (unsigned)(r1[n + i] >> 32);
            if (!PLAIN && tt.kind == 1) {  // the right tap is the left tap again: bytes 0-2 repeated as bytes 3-5
                hi0[i] = lo0[i] >> 8;
                lo0[i] = (lo0[i] & 0xFFFFFFu) | (lo0[i] << 24);
                hi1[i] = lo1[i] >> 8;
                lo1[i] = (lo1[i] & 0xFFFFFFu) | (lo1[i] << 24);
            }
            if (!PLAIN && WRAP && tt.kind == 2) {  // the panorama's seam: the right tap by itself
                const unsigned p01 = pb_bil_load_px(s, tt.o0 + tt.c1off, frame_bytes), p11 = pb_bil_load_px(s, tt.o1 + tt.c1off, frame_bytes);
                lo0[i] = (lo0[i] & 0xFFFFFFu) | (p01 << 24);
                hi0[i] = p01 >> 8;
                lo1[i] = (lo1[i] & 0xFFFFFFu) | (p11 << 24);
                hi1[i] = p11 >> 8;
            }
        }
        const pb_f2 tx2 = {t[n].tx, t[n + 1].tx}, ty2 = {t[n].ty, t[n + 1].ty};
        pb_bil_mix64x2(lo0, hi0, lo1, hi1, tx2, ty2, &out[n]);
        if (t[n].kind < 0) out[n] = 0u;
        if (t[n + 1].kind < 0) out[n + 1] = 0u;
    }
}

// The direct-gather path's loads and arithmetic for one lane: its 16 pixels q(n) = (2 n + q0) & 31 along the collapsed polynomial
// cf, eight pixels' loads in flight together (WIDE: two 8-byte loads per pixel, else four dwords), each blended pixel parked in the
// wave's LDS at park_p + q * park_q.
template <bool WIDE, bool TD3>
__device__ __forceinline__ void pb_bil_direct_gather(const uint8_t* __restrict__ s, unsigned* win, const pb_f2 cf[5], const unsigned gbase,
                                                     const unsigned rowbytes, const int q0, const int park_p, const int park_q) {
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        unsigned lo[8][2], hi[8][2];
        float wy[8], wx[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int q = (2 * (8 * grp + m) + q0) & 31;
            const pb_f2 sv = pb_bil_eval<TD3>(cf, pb_tile_coord(q));
            wy[m] = __builtin_amdgcn_fractf(sv.x);
            wx[m] = __builtin_amdgcn_fractf(sv.y);
            unsigned g = gbase + (unsigned)(int)sv.x * rowbytes + __umul24((unsigned)(int)sv.y, 3u);  // (s >= 0.5: truncation is floor)
            if (WIDE) {
                unsigned long long t0, t1;
                __builtin_memcpy(&t0, s + g, 8);
                __builtin_memcpy(&t1, s + g + rowbytes, 8);
                lo[m][0] = (unsigned)t0;
                hi[m][0] = (unsigned)(t0 >> 32);
                lo[m][1] = (unsigned)t1;
                hi[m][1] = (unsigned)(t1 >> 32);
            } else {
                unsigned t00, t01, t10, t11;
                __builtin_memcpy(&t00, s + g, 4);
                __builtin_memcpy(&t01, s + g + 3u, 4);
                __builtin_memcpy(&t10, s + g + rowbytes, 4);
                __builtin_memcpy(&t11, s + g + rowbytes + 3u, 4);
                lo[m][0] = (t00 & 0xFFFFFFu) | (t01 << 24);
                hi[m][0] = t01 >> 8;
                lo[m][1] = (t10 & 0xFFFFFFu) | (t11 << 24);
                hi[m][1] = t11 >> 8;
            }
        }
#pragma unroll
        for (int m = 0; m < 8; m += 2) {
            const unsigned l0[2] = {lo[m][0], lo[m + 1][0]}, h0[2] = {hi[m][0], hi[m + 1][0]}, l1[2] = {lo[m][1], lo[m + 1][1]}, h1[2] = {hi[m][1], hi[m + 1][1]};
            const pb_f2 tx2 = {wx[m], wx[m + 1]}, ty2 = {wy[m], wy[m + 1]};
            unsigned o[2];
            pb_bil_mix64x2(l0, h0, l1, h1, tx2, ty2, o);
            win[park_p + ((2 * (8 * grp + m) + q0) & 31) * park_q] = o[0];
            win[park_p + ((2 * (8 * grp + m + 1) + q0) & 31) * park_q] = o[1];
        }
    }
}

// The HALVES path of a plain tile (PB_TILE_HALVES): the two half windows staged one after the other in the wave's LDS region, the
// lane's eight pixels of each half sampled from it like a window tile's.  The model is evaluated in the order the direct path would use
// for the tile (the slot's window geometry was measured with exactly this evaluation: pb_bilinear_halves_kernel).
template <bool TD3>
__device__ __forceinline__ void pb_bil_halves_vals(const PbHot& Hd, const PbTileEntry* __restrict__ e, const int lane, unsigned* win,
                                                   const uint8_t* __restrict__ s, unsigned v[16]) {
    const int xg = lane & 7, yb = lane >> 3;
    const unsigned rowbytes = 3u * (unsigned)Hd.src_w, safe_len = (rowbytes * (unsigned)Hd.src_h) & ~15u;
    const bool along_x = fabsf(e->c[1][0]) <= fabsf(e->c[5][0]);
    const pb_f2 half = {0.5f, 0.5f};
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)win;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const unsigned hw = part ? ((unsigned)e->bil_off & 0x7FFFFFFFu) : (unsigned)e->win_c0;
        const unsigned dr = PB_HALF_DR(hw), dc = PB_HALF_DC(hw), rows = PB_HALF_ROWS(hw), n16 = PB_HALF_N16(hw);
        const unsigned pitch = 16u * n16;
        const unsigned gbase = ((unsigned)e->anchor_r + dr) * rowbytes + 3u * ((unsigned)e->anchor_c + dc);
        pb_issue_window_loads(s, win, lane, gbase, rowbytes, (int)rows, (int)n16, safe_len);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pb_wave_sync();
        // LDS address of a tap = (row - dr) * pitch + (col - dc) * 3 + (the first sample's byte phase) + the region's base
        const unsigned a0w = (gbase & 15u) + base - dr * pitch - 3u * dc;
        if (along_x) {
#pragma unroll
            for (int kk = 0; kk < 4; kk += 2) {
                pb_f2 b0[5], b1[5], sv[4];
                unsigned o[4];
                pb_bil_collapse<TD3>(e, true, 4 * xg + kk, b0);
                pb_bil_collapse<TD3>(e, true, 4 * xg + kk + 1, b1);
                b0[0] = b0[0] - half;
                b1[0] = b1[0] - half;
                sv[0] = pb_bil_eval<TD3>(b0, pb_tile_coord(yb + 8 * (2 * part)));
                sv[1] = pb_bil_eval<TD3>(b0, pb_tile_coord(yb + 8 * (2 * part + 1)));
                sv[2] = pb_bil_eval<TD3>(b1, pb_tile_coord(yb + 8 * (2 * part)));
                sv[3] = pb_bil_eval<TD3>(b1, pb_tile_coord(yb + 8 * (2 * part + 1)));
                pb_bil_lds4(sv, pitch, a0w, o);
                v[(2 * part) * 4 + kk] = o[0];
                v[(2 * part + 1) * 4 + kk] = o[1];
                v[(2 * part) * 4 + kk + 1] = o[2];
                v[(2 * part + 1) * 4 + kk + 1] = o[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int jr = 2 * part + j;
                pb_f2 a[5], sv[4];
                pb_bil_collapse<TD3>(e, false, yb + 8 * jr, a);
                a[0] = a[0] - half;
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[k] = pb_bil_eval<TD3>(a, pb_tile_coord(4 * xg + k));
                pb_bil_lds4(sv, pitch, a0w, &v[jr * 4]);
            }
        }
        pb_wave_sync();  // every lane has read its taps: the region may be refilled
    }
}

// The window and the direct-gather path of a plain tile (LEAN / DIRECT), on the full tile model or on its TD3 part.
template <bool TD3>
__device__ __forceinline__ void pb_bil_model_vals(const PbHot& Hd, const PbTileEntry* __restrict__ e, const int flags, const int lane, unsigned* win,
                                                  const int windows, const uint8_t* __restrict__ s, unsigned v[16]) {
    const int xg = lane & 7, yb = lane >> 3;
    const unsigned rowbytes = 3u * (unsigned)Hd.src_w, frame_bytes = rowbytes * (unsigned)Hd.src_h, safe_len = frame_bytes & ~15u;
    const unsigned gbase = (unsigned)e->anchor_r * rowbytes + 3u * (unsigned)e->anchor_c;
    const bool along_x = fabsf(e->c[1][0]) <= fabsf(e->c[5][0]);  // |d row / du| <= |d row / dv|
    if ((flags & PB_TILE_LEAN) && windows) {
        const unsigned pitch = 16u * (unsigned)e->win_n16, a0 = (unsigned)e->win_a0;
        pb_issue_window_loads(s, win, lane, gbase, rowbytes, e->win_rows, e->win_n16, safe_len);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pb_wave_sync();
        const pb_f2 half = {0.5f, 0.5f};
        const unsigned a0w = a0 + (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)win;  // (LDS addresses are 32-bit offsets)
        if (along_x) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pb_f2 b[5], sv[4];
                unsigned o[4];
                pb_bil_collapse<TD3>(e, true, 4 * xg + k, b);
                b[0] = b[0] - half;  // s = f - 0.5, folded into the constant term (window path and direct path alike)
#pragma unroll
                for (int jr = 0; jr < 4; ++jr) sv[jr] = pb_bil_eval<TD3>(b, pb_tile_coord(yb + 8 * jr));
                pb_bil_lds4(sv, pitch, a0w, o);
#pragma unroll
                for (int jr = 0; jr < 4; ++jr) v[jr * 4 + k] = o[jr];
            }
        } else {
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                pb_f2 a[5], sv[4];
                pb_bil_collapse<TD3>(e, false, yb + 8 * jr, a);
                a[0] = a[0] - half;
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[k] = pb_bil_eval<TD3>(a, pb_tile_coord(4 * xg + k));
                pb_bil_lds4(sv, pitch, a0w, &v[jr * 4]);
            }
        }
        pb_wave_sync();  // every lane has read its taps: the window may be refilled (the other eye, the next path)
        return;
    }
    if ((flags & PB_TILE_HALVES) && windows) {
        pb_bil_halves_vals<TD3>(Hd, e, lane, win, s, v);
        return;
    }
    // direct gathers.  wide: an 8-byte load takes both taps of a row - allowed when even the box's last tap has 8 bytes of frame
    // behind it.
    const bool wide = gbase + (unsigned)(e->win_rows - 1) * rowbytes + 3u * (unsigned)(e->win_cols - 1) + 8u <= frame_bytes;
    const int p = lane & 31, hh = lane >> 5;
    const float num = along_x ? e->c[1][0] : e->c[5][0], den = along_x ? e->c[5][0] : e->c[1][0];
    const float slope = (den != 0.0f) ? -num / den : 0.0f;
    const int shift = (int)rintf(slope * ((float)p - 15.5f));
    pb_f2 cf[5];
    pb_bil_collapse<TD3>(e, along_x, p, cf);
    {
        const pb_f2 half = {0.5f, 0.5f};
        cf[0] = cf[0] - half;  // s = f - 0.5 (as in the window path)
    }
    const int park_p = along_x ? p : p * 33, park_q = along_x ? 33 : 1;  // the lane's pixel (p, q) or (q, p) parks at [y][x], 33-dword pitch
    if (wide) pb_bil_direct_gather<true, TD3>(s, win, cf, gbase, rowbytes, hh + shift, park_p, park_q);
    else pb_bil_direct_gather<false, TD3>(s, win, cf, gbase, rowbytes, hh + shift, park_p, park_q);
    pb_wave_sync();
#pragma unroll
    for (int jr = 0; jr < 4; ++jr)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[jr * 4 + k] = win[(yb + 8 * jr) * 33 + 4 * xg + k];
    pb_wave_sync();
}


// ---- one source's (or eye's) values for a tile -----------------------------------------------------------------------------
// v[jr * 4 + k] = the bilinear sample of pixel (4 xg + k, yb + 8 jr) of the tile (xg = lane & 7, yb = lane >> 3: the lane's four
// 12-byte stores), packed RGB.  Which path:
//   table   the entry names a slot of the exact coordinate table (bil_off >= 0): failed tiles, coarse models, tiles with invalid
//           pixels, an image edge or an eye's rim inside - every pixel from its stored coordinate, taps clamped;
//   black   nothing of the tile samples this source;
//   window  LEAN tile: the nearest mode's LDS window (same plan, same LDS-DMA loads), taps from LDS;
//   direct  plain tile whose window exceeds the LDS budget: both taps of a row in one 8-byte load straight from the frame,
//           unguarded (the tile's box, margin texel included, lies inside the frame and - an eye - inside its half).  The gathers
//           run like the nearest mode's (pb_win_tile, DIRECT): lane = pixel column (or row), sheared along the line of constant
//           source row, regrouped for the stores through the wave's LDS.
// The model is evaluated column-first when the source row changes least along x, row-first otherwise, in the window path and in
// the direct path alike: the LDS budget moves a tile between the two and must not move a bit of its coordinates.
// Registers: the paths compute four pixels at a time and keep only the packed results (round 3 held all coordinates and all taps:
// 136 / 202 VGPRs, 3 / 2 waves per SIMD).
template <bool WRAP>
__device__ __forceinline__ void pb_bil_vals(const PbHot& Hd, const PbTileEntry* __restrict__ e, const int flags, const int lane, unsigned* win,
                                            const int windows, const uint8_t* __restrict__ s, const PbBilCoord* __restrict__ bil_xy, const int cmin,
                                            const int cmax, unsigned v[16]) {
    const int xg = lane & 7, yb = lane >> 3;
    const int h = Hd.src_h, w = Hd.src_w;
    const unsigned frame_bytes = 3u * (unsigned)w * (unsigned)h;
    if (e->bil_off >= 0) {
        // Lane = one column (or, PB_TILE_TAB_Y, one row) of the tile, 16 pixels down the other direction, SHEARED along the line of
        // constant source row like the direct-gather path: the 32 lanes of a half-wave take 32 neighbouring pixels along the direction
        // in which the source position moves least.  Direction and shear are decided per slot at plan time by
        // pb_bilinear_orient_kernel, which stores the slot in walk order: the lane's n-th coordinate is t[(2 n + hh) * 32 + p],
        // 256 contiguous bytes per half-wave, and belongs to pixel (p, (2 n + hh + shift(p)) & 31) - or its transpose.  These are the tiles the models cannot follow - the rim of a
        // fisheye destination, where one output pixel step along the radius is tens of source rows but a step along the rim almost
        // none: with 4 x 4 pixels per lane every tap of a load instruction sat in a line of its own (64 lines per instruction, 32
        // instructions per lane) and c3's 1 506 such tiles cost 10 of its 52 us in the CUs' L1 line rate alone (PB_BIL_ABL=1024).
        // The blended pixels are regrouped for the 12-byte stores through the wave's LDS, like the direct-gather path's.
        const PbBilCoord* __restrict__ t = bil_xy + (size_t)(e->bil_off & PB_BIL_SLOT_MASK) * (PB_TILE * PB_TILE);
        const int p = lane & 31, hh = lane >> 5;
        const bool by_rows = (flags & PB_TILE_TAB_Y) != 0;
        const int shift = pb_bil_slot_shift(e->bil_off, p);
        int2 c[16];  // all sixteen coordinates first: one round trip
#pragma unroll
        for (int n = 0; n < 16; ++n) c[n] = *reinterpret_cast<const int2*>(t + (2 * n + hh) * PB_TILE + p);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            int4 q[4];
            unsigned o[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = make_int4(c[8 * half + 2 * i].x, c[8 * half + 2 * i].y, c[8 * half + 2 * i + 1].x, c[8 * half + 2 * i + 1].y);
            if (flags & PB_TILE_TAB_PLAIN) pb_bil_table8<WRAP, true>(s, q, h, w, cmin, cmax, frame_bytes, o);
            else pb_bil_table8<WRAP, false>(s, q, h, w, cmin, cmax, frame_bytes, o);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int a = (2 * (8 * half + m) + hh + shift) & 31;  // (the slot is stored in walk order: entry [2 n + hh][p] is this pixel's)
                win[by_rows ? p * 33 + a : a * 33 + p] = o[m];  // parked at [y][x], 33-dword pitch
            }
        }
        pb_wave_sync();
#pragma unroll
        for (int jr = 0; jr < 4; ++jr)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[jr * 4 + k] = win[(yb + 8 * jr) * 33 + 4 * xg + k];
        pb_wave_sync();
        return;
    }
    if (!(flags & (PB_TILE_LEAN | PB_TILE_DIRECT))) {  // BLACK (every other class has a table slot)
#pragma unroll
        for (int n = 0; n < 16; ++n) v[n] = 0u;
        return;
    }
    if (flags & PB_TILE_TD3) pb_bil_model_vals<true>(Hd, e, flags, lane, win, windows, s, v);
    else pb_bil_model_vals<false>(Hd, e, flags, lane, win, windows, s, v);
}

// the lane's four 12-byte stores (4 consecutive pixels x 4 rows); tiles on the image's edge are clipped
template <bool NT>
__device__ __forceinline__ void pb_bil_store(const unsigned v[16], uint8_t* __restrict__ d, const int X0, const int Y0, const int lane, const int W,
                                             const int H) {
    const int xg = lane & 7, yb = lane >> 3, x = X0 + 4 * xg;
    const bool inside = X0 + PB_TILE <= W && Y0 + PB_TILE <= H;
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        const int y = Y0 + yb + 8 * jr;
        if (!inside && y >= H) continue;
        const unsigned long long off = 3ull * ((unsigned long long)y * W + x);
        if ((inside || x + 3 < W) && (((uintptr_t)d + off) & 3u) == 0) {
            pb_store3<NT>(pb_pack_px4(v[jr * 4], v[jr * 4 + 1], v[jr * 4 + 2], v[jr * 4 + 3]), d + off);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (inside || x + k < W) {
                    d[off + 3 * k + 0] = (uint8_t)(v[jr * 4 + k] & 0xFF);
                    d[off + 3 * k + 1] = (uint8_t)((v[jr * 4 + k] >> 8) & 0xFF);
                    d[off + 3 * k + 2] = (uint8_t)((v[jr * 4 + k] >> 16) & 0xFF);
                }
        }
    }
}

// The launch table is laid out in VIRTUAL workgroups of four slots, virtual workgroup B on XCD B & 7 (pb_launch_table_kernel).  A REAL
// workgroup is WAVES waves - FOUR (a whole virtual workgroup) or TWO (round 6), chosen per plan with its LDS pool (pb_build_bilinear_launch):
// two-wave workgroups release their LDS and their wave slots in finer grain (a wave that has finished waits for one neighbour, not
// three; the chip's last workgroups are half as long) and a pair of a stitch meets a barrier of its own - c2 50.2 -> 47.8 us, c5 78.5 ->
// 77.2 - but pool their regions over two slots instead of four, which costs c1 its small pool (it keeps four waves).  Real workgroup w of
// a frame runs on XCD w & 7 (round-robin dispatch: speed only) and takes part (w >> 3) % PARTS of that XCD's (w >> 3) / PARTS-th
// virtual workgroup, PARTS = 4 / WAVES: a slot keeps the XCD the plan gave it.
template <int WAVES>
__device__ __forceinline__ unsigned pb_bil_slot_of(unsigned wg, unsigned wave) {
    constexpr unsigned PARTS = 4u / (unsigned)WAVES;
    const unsigned idx = wg >> 3, B = ((idx / PARTS) << 3) | (wg & 7u), part = idx % PARTS;
    return B * 4u + part * (unsigned)WAVES + wave;
}

// One wave per tile, launched like pb_hot_win_kernel: `table` is the plan's LAUNCH-ORDER table (the entry says which tile it is),
// frames of a batch are a grid dimension.  The tile's fix pixels - the model's truncation differs from the faithful one: mostly a
// coordinate a hair from an integer, harmless here, but also the genuine discontinuities a polynomial cannot follow inside an
// otherwise modelled tile (the edge of a lens inverse's domain, a validity or image boundary) - are redone from their exact
// coordinates by the tile's wave after its stores, like the nearest mode's.  bil_xy == nullptr: the plan has no coordinate table
// (it would not fit); tiles that need one, and the fix pixels, are then left to pb_bilinear_fix_kernel.
template <int SRC_KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES, PB_BIL_WPE) void pb_bilinear_hot_kernel(const PbHot Hd, const PbTileEntry* __restrict__ table,
                                                                              const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                              const unsigned groups_per_frame, unsigned long long src_stride,
                                                                              unsigned long long dst_stride, int windows,
                                                                              const PbBilCoord* __restrict__ bil_xy, const int32_t* __restrict__ fix_px,
                                                                              const PbBilCoord* __restrict__ fix_xy) {
    asm volatile("" ::"s"(table), "s"(Hd.dst_w), "s"(Hd.dst_h), "s"(Hd.src_w), "s"(Hd.src_h), "s"(Hd.win_budget), "s"(groups_per_frame));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned wg = blockIdx.x;
    if (wg >= groups_per_frame) {  // a batch: which frame (groups_per_frame: REAL workgroups per frame)
        const unsigned f = wg / groups_per_frame;
        wg -= f * groups_per_frame;
        src += (unsigned long long)f * src_stride;
        dst += (unsigned long long)f * dst_stride;
    }
    PbTileEntry entry;
    const unsigned vslot = (unsigned)__builtin_amdgcn_readfirstlane((int)pb_bil_slot_of<WAVES>(wg, (unsigned)wave));
    pb_load_entry(table + vslot, entry);
    const PbTileEntry* __restrict__ e = &entry;
    const int flags = e->flags;
    if (flags & PB_TILE_SKIP) return;
    if (e->bil_off >= 0 && !bil_xy) return;  // (no coordinate table: the float64 pass owns the tile)
    const int tx = e->tile_xy & 0xFFFF, ty = (int)((unsigned)e->tile_xy >> 16);
    unsigned v[16];
    // (the wave's LDS region: its slot says where in the workgroup's pool - pb_bilinear_pool_kernel)
    pb_bil_vals<SRC_KIND == PB_KIND_PANO>(Hd, e, flags, lane, pb_dyn_lds + ((unsigned)e->win_r0 >> 2), windows, src, bil_xy, 0, Hd.src_w, v);
    pb_bil_store<SRC_KIND == PB_KIND_CAMERA>(v, dst, tx * PB_TILE, ty * PB_TILE, lane, Hd.dst_w, Hd.dst_h);
    const int n_fix = e->fix_cnt;
    if (n_fix > 0 && fix_xy && e->bil_off < 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's own stores have completed
        if (lane < n_fix) {
            const unsigned p = (unsigned)fix_px[e->fix_off + lane];
            const PbBilCoord q = fix_xy[e->fix_off + lane];
            const unsigned px = pb_bil_table_px<SRC_KIND == PB_KIND_PANO>(src, q.y, q.x, Hd.src_h, Hd.src_w, 0, Hd.src_w, 3u * (unsigned)Hd.src_w * (unsigned)Hd.src_h);
            uint8_t* o = dst + 3ull * p;
            o[0] = (uint8_t)(px & 0xFF);
            o[1] = (uint8_t)((px >> 8) & 0xFF);
            o[2] = (uint8_t)((px >> 16) & 0xFF);
        }
    }
}

// float64 faithful coordinates; mode 0: the plan's failed tiles (4 blocks each), mode 1: every pixel (no plan state)
// mode 0 also takes the plan's fix list (blocks beyond the failed tiles): a pixel is on it because the model's index
// differs from the faithful one - mostly a coordinate a hair from an integer, harmless here, but also the genuine
// discontinuities a polynomial cannot follow inside an otherwise modelled tile (the edge of a lens inverse's domain,
// a validity or image boundary): those pixels take the float64 coordinates like the failed tiles.
template <int SRC_KIND>
__global__ __launch_bounds__(PB_BLOCK) void pb_bilinear_fix_kernel(const PbParams P, const int32_t* __restrict__ fail_tiles,
                                                                   int all_pixels, const uint8_t* __restrict__ src,
                                                                   uint8_t* __restrict__ dst, int n_frames,
                                                                   unsigned long long src_stride, unsigned long long dst_stride,
                                                                   int n_fail_tiles = 0, const int32_t* __restrict__ fix_px = nullptr,
                                                                   int n_fix_px = 0, const int32_t* __restrict__ more_tiles = nullptr,
                                                                   int n_fail_only = 0) {
    int i, j;
    if (all_pixels) {
        const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
        if (p >= (unsigned)P.dst.height * (unsigned)P.dst.width) return;
        i = p / (unsigned)P.dst.width;
        j = p - (unsigned)i * (unsigned)P.dst.width;
    } else if ((int)blockIdx.x >= 4 * n_fail_tiles && fix_px) {
        const unsigned item = (blockIdx.x - 4u * (unsigned)n_fail_tiles) * PB_BLOCK + threadIdx.x;
        if (item >= (unsigned)n_fix_px) return;
        const unsigned p = (unsigned)fix_px[item];
        i = p / (unsigned)P.dst.width;
        j = p - (unsigned)i * (unsigned)P.dst.width;
    } else {
        // n_fail_tiles: the plan's failed tiles (the first n_fail_only, from fail_tiles) and its COARSE tiles (more_tiles) together
        const int k = blockIdx.x >> 2;
        const int t = (more_tiles && k >= n_fail_only) ? more_tiles[k - n_fail_only] : fail_tiles[k];
        const int ty = t / pb_tiles_x(P), tx = t - ty * pb_tiles_x(P);
        const int local = (blockIdx.x & 3) * 256 + threadIdx.x;
        i = ty * PB_TILE + (local >> 5);
        j = tx * PB_TILE + (local & 31);
        if (i >= P.dst.height || j >= P.dst.width) return;
    }
    PbCoord c = pb_dst_coord(P, i, j);
    c = pb_rotate_all(P, c);
    double f0, f1;
    pb_src_pretrunc<SRC_KIND>(P, c, f0, f1);
    bool live = !c.inv && f0 == f0 && f1 == f1 && fabs(f0) < 1.0e9 && fabs(f1) < 1.0e9;
    if (SRC_KIND == PB_KIND_CAMERA) live = live && f0 >= 0.0 && f0 < (double)P.src.height && f1 >= 0.0 && f1 < (double)P.src.width;
    const size_t p = (size_t)i * P.dst.width + j;
    // integer bases keep the float32 tap arithmetic exact enough: s - base is in [-1, 1)
    const double sy = f0 - 0.5, sx = f1 - 0.5;
    const int by = live ? (int)floor(sy) : 0, bx = live ? (int)floor(sx) : 0;
    for (int f = 0; f < n_frames; ++f) {
        unsigned v = 0;
        if (live) v = pb_bilinear_taps<SRC_KIND>(P, src + (unsigned long long)f * src_stride, (float)(sy - by), (float)(sx - bx), by, bx);
        uint8_t* o = dst + (unsigned long long)f * dst_stride + 3 * p;
        o[0] = (uint8_t)(v & 0xFF);
        o[1] = (uint8_t)((v >> 8) & 0xFF);
        o[2] = (uint8_t)((v >> 16) & 0xFF);
    }
}

// ---- double-fisheye source (faithful float64 coordinates per pixel; an opt-in mode off the hot path) ----------------
// one eye's bilinear sample (0 where the nearest mode is black for that eye): eye image = columns [col0, col0 + we) of the
// frame, mirrored when `mirror` (the right eye, projection.py:430-431)
__device__ __forceinline__ unsigned pb_bilinear_eye(const PbParams& P, const uint8_t* __restrict__ s, double lat, double lon, int we, double cx,
                                                    int col0, bool mirror) {
    const int h = P.src.height, w = P.src.width;
    const double dist = pb_lens_forward(P.src.lens, lat, P.rect_max) * P.src.f_distance;
    double sl, cl;
    pb_expi_np(lon, &sl, &cl);  // np.exp(lon * 1j)
    const double f0 = ((sl * dist) * -1.0) + P.src_cy, f1 = (cl * dist) + cx;
    const bool live = f0 == f0 && f1 == f1 && fabs(f0) < 1.0e9 && fabs(f1) < 1.0e9 && f0 >= 0.0 && f0 < (double)h && f1 >= 0.0 && f1 < (double)we;
    if (!live) return 0u;
    const double sy = f0 - 0.5, sx = f1 - 0.5;
    const int by = (int)floor(sy), bx = (int)floor(sx);
    const float ty = (float)(sy - by), tx = (float)(sx - bx);
    const int r0 = min(max(by, 0), h - 1), r1 = min(max(by + 1, 0), h - 1);
    int c0 = min(max(bx, 0), we - 1), c1 = min(max(bx + 1, 0), we - 1);
    c0 = col0 + (mirror ? we - 1 - c0 : c0);
    c1 = col0 + (mirror ? we - 1 - c1 : c1);
    const unsigned p00 = pb_load_px(s, r0 * w + c0), p01 = pb_load_px(s, r0 * w + c1);
    const unsigned p10 = pb_load_px(s, r1 * w + c0), p11 = pb_load_px(s, r1 * w + c1);
    unsigned out = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float a = (float)((p00 >> (8 * ch)) & 0xFF), b = (float)((p01 >> (8 * ch)) & 0xFF);
        const float c = (float)((p10 >> (8 * ch)) & 0xFF), d = (float)((p11 >> (8 * ch)) & 0xFF);
        const float top = fmaf(tx, b - a, a), bot = fmaf(tx, d - c, c);
        const float v = fmaf(ty, bot - top, top);
        out |= ((unsigned)(int)rintf(fminf(fmaxf(v, 0.0f), 255.0f))) << (8 * ch);
    }
    return out;
}

// one output pixel of the double-fisheye bilinear mode from the float64 chain (the mode's definition on the device)
__device__ __forceinline__ unsigned pb_bilinear_double_px(const PbParams& P, int i, int j, const uint8_t* __restrict__ s) {
    PbCoord c = pb_dst_coord(P, i, j);
    c = pb_rotate_all(P, c);
    if (c.inv) return 0u;
    const double lat_r = (c.lat * -1.0) + PB_PI;  // projection.py:426-427
    const double fl = pb_merge_factor(P, c.lat), fr = pb_merge_factor(P, lat_r);
    const unsigned l = pb_bilinear_eye(P, s, c.lat, c.lon, P.src_eye_w, P.src_cx, 0, false);
    const unsigned r = pb_bilinear_eye(P, s, lat_r, c.lon, P.src_eye_w_right, P.src_cx_r, P.src_eye_w, true);
    return pb_blend_u8(l & 0xFF, r & 0xFF, fl, fr) | (pb_blend_u8((l >> 8) & 0xFF, (r >> 8) & 0xFF, fl, fr) << 8) |
           (pb_blend_u8((l >> 16) & 0xFF, (r >> 16) & 0xFF, fl, fr) << 16);
}

__global__ __launch_bounds__(PB_BLOCK) void pb_bilinear_double_kernel(const PbParams P, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                      int n_frames, unsigned long long src_stride, unsigned long long dst_stride) {
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= (unsigned)P.dst.height * (unsigned)P.dst.width) return;
    const int i = p / (unsigned)P.dst.width, j = p - (unsigned)i * (unsigned)P.dst.width;
    for (int f = 0; f < n_frames; ++f) {
        const unsigned v = pb_bilinear_double_px(P, i, j, src + (unsigned long long)f * src_stride);
        uint8_t* o = dst + (unsigned long long)f * dst_stride + 3ull * p;
        o[0] = (uint8_t)(v & 0xFF);
        o[1] = (uint8_t)((v >> 8) & 0xFF);
        o[2] = (uint8_t)((v >> 16) & 0xFF);
    }
}

// ---- double-fisheye source -------------------------------------------------------------------------------------------------
// Launched over the plan's PAIR-layout table (pb_kernels_tile.hpp; frames of a batch: a grid dimension), one wave per slot, every slot
// with ONE eye's entry in scalar registers and its own region of the workgroup's LDS pool - the register state of a single-source tile:
//   SOLO slot   a tile that sees one eye with weight exactly 1: that eye's bilinear sample, stored (the single-source tile code with the
//               eye's column range);
//   pair slots  a tile that samples both eyes: wave R (PB_TILE_PAIR_R) samples the right eye by its entry's path (table / black /
//               window / direct), parks its sixteen packed pixels per lane in ITS region (dead by then) with the numbers wave L needs
//               of its entry, and meets the workgroup's barrier; wave L (PB_TILE_TWO) samples the left eye, meets the barrier, reads
//               R's pixels and blends with the tile's weight class like the nearest mode - UNIT the integer sum, ROW the row table, LAT
//               the stored latitudes; a FAILED tile (both eyes from the coordinate table) with the faithful factors the nearest mode
//               stores for its pixels (PbDoubleFix) - stores, and redoes the fix pixels of either eye's list from their exact coordinates
//               and stored factors.
// Pair workgroups hold nothing but pair slots (and pads that only meet the barrier): every wave of one reaches the barrier exactly once.
// Round 5's one wave per two-eye tile sampled the eyes one after the other and held both results: 168 VGPRs, 3 waves per SIMD.
// What only SOME waves need - the blend's tables and constants (projection.py:414-418), the fix lists - lives in ONE plan-resident block
// behind a pointer, fetched by the waves that use it (a by-row / by-latitude / failed tile, a tile with fix pixels).  As kernel arguments
// (round 5: the whole 1216-byte parameter block by value, then six pointers and four doubles) they sat in scalar registers for the
// life of EVERY wave next to its 64-register tile entry: the kernel ran at the edge of the 102 SGPRs, and whether the compiler kept the
// entry there or moved it through VGPR lanes (1 400 v_readlane in the tile code, +40 % vector instructions per wave, c5 80 -> 91 us)
// changed with the spelling of an unrelated `if` (experiments/README.md round 6).  experiments/r6/isa_stats.py prints the lane traffic.
struct PbDblTables {
    double mrg_min, mrg_max_safe, mrg_max, mrg_range;
    const PbSepRow* rows;
    const double* lat_tab;
    const int32_t* fix_px;
    const PbBilCoord* fix_xy;
    const PbDoubleFix* tile_fix;
    const PbDoubleFix* px_fix;
};
#define PB_PAIR_HDR 1024  // dword offset of wave R's header in its region, behind its 64 x 16 packed pixels: fix_cnt, fix_off, aux_off
template <int WMODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, PB_BIL_WPE_DBL) void pb_bilinear_double_hot_kernel(const PbHot Hd, const int eye_w, const PbTileEntry* __restrict__ ltable,
                                                                                     const PbDblTables* __restrict__ X,
                                                                                     const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                                     const unsigned groups_per_frame, unsigned long long src_stride,
                                                                                     unsigned long long dst_stride, int windows,
                                                                                     const PbBilCoord* __restrict__ bil_xy) {
    asm volatile("" ::"s"(ltable), "s"(Hd.dst_w), "s"(Hd.dst_h), "s"(Hd.src_w), "s"(Hd.src_h), "s"(Hd.win_budget), "s"(groups_per_frame));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned wg = blockIdx.x;
    if (wg >= groups_per_frame) {
        const unsigned f = wg / groups_per_frame;
        wg -= f * groups_per_frame;
        src += (unsigned long long)f * src_stride;
        dst += (unsigned long long)f * dst_stride;
    }
    const unsigned vslot = (unsigned)__builtin_amdgcn_readfirstlane((int)pb_bil_slot_of<WAVES>(wg, (unsigned)wave));
    PbTileEntry entry;
    pb_load_entry(ltable + vslot, entry);
    const int flags = entry.flags;
    const bool pair = (flags & (PB_TILE_TWO | PB_TILE_PAIR_R)) != 0, right = (flags & (PB_TILE_PAIR_R | PB_TILE_EYE_R)) != 0;
    // (a slot that leaves early - a pad, a tile on the plan's float64 list, which pb_bilinear_double_fix_kernel repaints - still owes its
    // pair workgroup the barrier, and a wave R its header: wave L walks the fix lists it names)
    const bool leave = (flags & PB_TILE_SKIP) || (entry.bil_off >= 0 && !bil_xy);
    if (leave) {
        if (pair) {
            if (!(flags & PB_TILE_SKIP) && (flags & PB_TILE_PAIR_R) && lane == 0) {
                unsigned* hdr = pb_dyn_lds + (((unsigned)entry.win_r0 & 0xFFFFu) >> 2) + PB_PAIR_HDR;
                hdr[0] = (unsigned)entry.fix_cnt;
                hdr[1] = (unsigned)entry.fix_off;
                hdr[2] = (unsigned)entry.aux_off;
            }
            __syncthreads();
        }
        return;
    }
    const int tx = entry.tile_xy & 0xFFFF, ty = (int)((unsigned)entry.tile_xy >> 16);
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    const int xg = lane & 7, yb = lane >> 3;
    const int W = Hd.dst_w, H = Hd.dst_h;
    unsigned* win = pb_dyn_lds + (((unsigned)entry.win_r0 & 0xFFFFu) >> 2);  // (the wave's LDS region: its slot says where in the workgroup's pool)
    unsigned a[16];
    pb_bil_vals<false>(Hd, &entry, flags & ~(PB_TILE_TWO | PB_TILE_PAIR_R), lane, win, windows, src, bil_xy, right ? eye_w : 0, right ? Hd.src_w : eye_w, a);
    if (!pair) {
        pb_bil_store<false>(a, dst, X0, Y0, lane, W, H);
        return;
    }
    if (flags & PB_TILE_PAIR_R) {
        // (every path of pb_bil_vals ends behind a wave-wide LDS hand-off: no lane still reads the region)
        uint4* x = reinterpret_cast<uint4*>(win);
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q * 64 + lane] = make_uint4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
        if (lane == 0) {
            win[PB_PAIR_HDR] = (unsigned)entry.fix_cnt;
            win[PB_PAIR_HDR + 1] = (unsigned)entry.fix_off;
            win[PB_PAIR_HDR + 2] = (unsigned)entry.aux_off;
        }
        __syncthreads();
        return;
    }
    __syncthreads();
    const unsigned* xr = pb_dyn_lds + ((unsigned)entry.win_r0 >> 18);  // wave R's region: the high half of the slot's offset word (pb_bilinear_pool_kernel)
    unsigned ar[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 t = reinterpret_cast<const uint4*>(xr)[q * 64 + lane];
        ar[4 * q] = t.x; ar[4 * q + 1] = t.y; ar[4 * q + 2] = t.z; ar[4 * q + 3] = t.w;
    }
    const int nr = __builtin_amdgcn_readfirstlane((int)xr[PB_PAIR_HDR]), off_r = __builtin_amdgcn_readfirstlane((int)xr[PB_PAIR_HDR + 1]);
    const int nl = entry.fix_cnt, off_l = entry.fix_off;
    if (flags & PB_TILE_FAILED) {
        // the faithful factors of every pixel of a failed tile (slot = the right-eye entry's aux_off, pb_double_tables_kernel)
        const PbDoubleFix* __restrict__ slot = X->tile_fix + (size_t)__builtin_amdgcn_readfirstlane((int)xr[PB_PAIR_HDR + 2]) * (PB_TILE * PB_TILE);
#pragma unroll
        for (int jr = 0; jr < 4; ++jr)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const PbDoubleFix* __restrict__ t = slot + (yb + 8 * jr) * PB_TILE + 4 * xg + k;
                a[jr * 4 + k] = pb_sep_blend(a[jr * 4 + k], ar[jr * 4 + k], t->fl, t->fr);
            }
    } else {
        const bool by_row = WMODE == 1 && (flags & PB_TILE_W_ROW) != 0;
        const bool by_lat = WMODE == 2 && (flags & PB_TILE_W_LAT) != 0;
        if (!by_row && !by_lat) {  // UNIT: l * 1.0 + r * 1.0 is the exact integer l + r; astype(uint8) keeps its low 8 bits
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const unsigned l = a[n], r = ar[n];
                a[n] = (((l & 0x00FF00FFu) + (r & 0x00FF00FFu)) & 0x00FF00FFu) | (((l & 0x0000FF00u) + (r & 0x0000FF00u)) & 0x0000FF00u);
            }
        } else {
            const PbSepRow* __restrict__ rows = X->rows;
            const double* __restrict__ lat_tab = X->lat_tab;
            const double mrg_min = X->mrg_min, mrg_max_safe = X->mrg_max_safe, mrg_max = X->mrg_max, mrg_range = X->mrg_range;
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                double wl = 1.0, wr = 1.0;
                if (by_row) {
                    const PbSepRow R = rows[min(Y0 + yb + 8 * jr, H - 1)];
                    wl = R.f_l;
                    wr = R.f_r;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (by_lat) {
                        const double t = lat_tab[(size_t)entry.aux_off * PB_LAT_TILE_DOUBLES + (yb + 8 * jr) * PB_TILE + 4 * xg + k];
                        wl = pb_merge_factor_of(mrg_min, mrg_max_safe, mrg_max, mrg_range, t);
                        wr = pb_merge_factor_of(mrg_min, mrg_max_safe, mrg_max, mrg_range, (t * -1.0) + PB_PI);
                    }
                    a[jr * 4 + k] = pb_sep_blend(a[jr * 4 + k], ar[jr * 4 + k], wl, wr);
                }
            }
        }
    }
    pb_bil_store<false>(a, dst, X0, Y0, lane, W, H);
    if (nl + nr > 0 && bil_xy) {  // (the fix list's coordinates exist with the coordinate table)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's own stores have completed
        const unsigned frame_bytes = 3u * (unsigned)Hd.src_w * (unsigned)Hd.src_h;
        const int32_t* __restrict__ fix_px = X->fix_px;
        const PbBilCoord* __restrict__ fix_xy = X->fix_xy;
        const PbDoubleFix* __restrict__ px_fix = X->px_fix;
        for (int base = 0; base < nl + nr; base += 64) {
            const int n = base + lane;
            if (n < nl + nr) {
                const int item = n < nl ? off_l + n : off_r + (n - nl);
                const unsigned p = (unsigned)fix_px[item];
                const PbBilCoord ql = fix_xy[2 * item], qr = fix_xy[2 * item + 1];
                const unsigned l = pb_bil_table_px<false>(src, ql.y, ql.x, Hd.src_h, Hd.src_w, 0, eye_w, frame_bytes);
                const unsigned r = pb_bil_table_px<false>(src, qr.y, qr.x, Hd.src_h, Hd.src_w, eye_w, Hd.src_w, frame_bytes);
                const unsigned px = pb_sep_blend(l, r, px_fix[item].fl, px_fix[item].fr);
                uint8_t* o = dst + 3ull * p;
                o[0] = (uint8_t)(px & 0xFF);
                o[1] = (uint8_t)((px >> 8) & 0xFF);
                o[2] = (uint8_t)((px >> 16) & 0xFF);
            }
        }
    }
}

// ---- plan creation -------------------------------------------------------------------------------------------------------------
// Which tiles the models cannot serve in this mode, per source / eye: failed tiles, COARSE models (beyond 1/1024 px of the faithful
// coordinate somewhere: fine for the truncating sampler, whose exceptions are tabulated; too coarse to interpolate at), tiles that
// are not plain (generic: partly outside the image or the eye, wrapping; MASKED: invalid destination pixels inside), and - an eye -
// plain tiles whose taps' reach leaves the eye's own columns.  Each gets a slot of the coordinate table (PbTileEntry::bil_off);
// `list` names the tiles with a slot that are not on the fail list (what the float64 fallback pass recomputes besides the failed
// tiles).  counters: [0] tiles listed, [1] slots handed out.  The LDS budget only moves tiles between LEAN and DIRECT, never in or
// out of this set.
__device__ __forceinline__ bool pb_bil_needs_table(const PbTileEntry& e, bool eye, int cmin, int cmax) {
    const int f = e.flags;
    if (f & PB_TILE_FAILED) return true;
    if (f & PB_TILE_BLACK) return false;
    if (!(f & (PB_TILE_LEAN | PB_TILE_DIRECT))) return true;
    if (f & (PB_TILE_COARSE | PB_TILE_MASKED)) return true;
    return eye && !(e.win_c0 >= cmin && e.win_c0 + e.win_cols <= cmax);
}
__global__ void pb_bilinear_tile_list_kernel(PbTileEntry* __restrict__ table_l, PbTileEntry* __restrict__ table_r, unsigned n_tiles, int eye_w, int src_w,
                                             int32_t* __restrict__ list, unsigned* __restrict__ counters) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const bool need_l = pb_bil_needs_table(table_l[t], table_r != nullptr, 0, eye_w);
    const bool need_r = table_r && pb_bil_needs_table(table_r[t], true, eye_w, src_w);
    table_l[t].bil_off = need_l ? (int)atomicAdd(&counters[1], 1u) : -1;
    if (table_r) table_r[t].bil_off = need_r ? (int)atomicAdd(&counters[1], 1u) : -1;
    const bool failed = ((table_l[t].flags | (table_r ? table_r[t].flags : 0)) & PB_TILE_FAILED) != 0;
    if ((need_l || need_r) && !failed) list[atomicAdd(&counters[0], 1u)] = (int32_t)t;
}

// Plan creation: which direct-gather slots of the bilinear launch table can be served as two half windows (PB_TILE_HALVES).  One wave per
// slot: every pixel's tap coordinate with the hot path's own evaluation (same functions, same order, TD3 or not), the bounding box of the
// taps of the top and of the bottom half, both at most `budget` bytes of LDS and loadable like a window tile's.  counters[2] counts them.
// solo_only: a double-fisheye plan's one-eye slots only (the diagnostic build's PB_BIL_OFF bit 8: no half windows for pair slots).  Round
// 5's single wave per two-eye tile staged FOUR windows one after the other with half windows (c5 113 us against 106); a pair slot is a
// one-eye tile like any other.
__global__ __launch_bounds__(256) void pb_bilinear_halves_kernel(PbTileEntry* __restrict__ ltable, unsigned n_slots, int budget, int src_h, int src_w, int solo_only,
                                                                 unsigned* __restrict__ counters) {
    const unsigned v = blockIdx.x * 4u + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform: the entry is read with scalar loads)
    const int lane = threadIdx.x & 63;
    if (v >= n_slots) return;
    PbTileEntry* slot = ltable + v;
    PbTileEntry L;
    pb_load_entry(slot, L);
    const PbTileEntry* __restrict__ e = &L;
    const int f = e->flags;
    if ((f & PB_TILE_SKIP) || !(f & PB_TILE_DIRECT) || (f & PB_TILE_MASKED) || e->bil_off >= 0) return;
    if (solo_only && !(f & PB_TILE_SOLO)) return;
    const unsigned rowbytes = 3u * (unsigned)src_w, frame_bytes = rowbytes * (unsigned)src_h, safe_len = frame_bytes & ~15u;
    if (rowbytes & 15u) return;
    const int xg = lane & 7, yb = lane >> 3;
    const bool along_x = fabsf(e->c[1][0]) <= fabsf(e->c[5][0]);
    const bool td3 = (f & PB_TILE_TD3) != 0;
    const pb_f2 half = {0.5f, 0.5f};
    int ymin[2] = {0x7fffffff, 0x7fffffff}, ymax[2] = {-1, -1}, xmin[2] = {0x7fffffff, 0x7fffffff}, xmax[2] = {-1, -1};
    bool ok = true;
    for (int jr = 0; jr < 4; ++jr)
        for (int k = 0; k < 4; ++k) {
            pb_f2 c[5], sv;
            if (td3) {
                pb_bil_collapse<true>(e, along_x, along_x ? 4 * xg + k : yb + 8 * jr, c);
                c[0] = c[0] - half;
                sv = pb_bil_eval<true>(c, pb_tile_coord(along_x ? yb + 8 * jr : 4 * xg + k));
            } else {
                pb_bil_collapse<false>(e, along_x, along_x ? 4 * xg + k : yb + 8 * jr, c);
                c[0] = c[0] - half;
                sv = pb_bil_eval<false>(c, pb_tile_coord(along_x ? yb + 8 * jr : 4 * xg + k));
            }
            ok = ok && sv.x >= 0.0f && sv.y >= 0.0f && sv.x < 60000.0f && sv.y < 60000.0f;
            const int iy = (int)sv.x, ix = (int)sv.y, p = jr >> 1;
            ymin[p] = min(ymin[p], iy); ymax[p] = max(ymax[p], iy);
            xmin[p] = min(xmin[p], ix); xmax[p] = max(xmax[p], ix);
        }
    if (__builtin_amdgcn_ballot_w64(!ok) != 0) return;
    int packed[2];
    bool fits = true;
    for (int p = 0; p < 2; ++p) {
        const int y0 = pb_wave_min(ymin[p]), y1 = pb_wave_max(ymax[p]), x0 = pb_wave_min(xmin[p]), x1 = pb_wave_max(xmax[p]);
        const unsigned rows = (unsigned)(y1 - y0 + 2), cols = (unsigned)(x1 - x0 + 2);  // taps i and i + 1
        const unsigned lr0 = (unsigned)e->anchor_r + (unsigned)y0, lc0 = (unsigned)e->anchor_c + (unsigned)x0;
        const unsigned a0 = (3u * lc0) & 15u, n16 = (a0 + 3u * cols + 1u + 15u) >> 4;
        const unsigned last_chunk_end = ((lr0 + rows - 1u) * rowbytes + 3u * lc0 & ~15u) + 16u * n16;
        fits = fits && y0 >= 0 && y0 < 256 && x0 >= 0 && x0 < 1024 && rows < 128u && n16 <= 64u && rows * 16u * n16 <= (unsigned)budget &&
               lr0 + rows <= (unsigned)src_h && lc0 + cols <= (unsigned)src_w && last_chunk_end <= safe_len &&
               (lr0 + rows - 1u) * rowbytes + 3u * (lc0 + cols - 1u) + 4u <= frame_bytes;
        packed[p] = PB_HALF_PACK(y0, x0, rows, n16);
    }
    // (two half windows move more bytes than the tile's direct gathers do when they are large: beyond PB_BIL_HALVES_MAX bytes in sum the
    // direct path is the faster one - measured, experiments/README.md round 5)
    if (PB_HALF_ROWS(packed[0]) * 16u * PB_HALF_N16(packed[0]) + PB_HALF_ROWS(packed[1]) * 16u * PB_HALF_N16(packed[1]) > (unsigned)PB_BIL_HALVES_MAX) fits = false;
    if (!fits || lane != 0) return;
    slot->flags = f | PB_TILE_HALVES;
    slot->win_c0 = packed[0];
    slot->bil_off = (int)(0x80000000u | (unsigned)packed[1]);  // (negative like every tile without a coordinate-table slot)
    atomicAdd(&counters[2], 1u);
}

// The LDS POOL of a workgroup (round 5).  Round 4 gave each of a workgroup's four waves a region of the full window budget (12 KiB: three
// workgroups per CU), though most windows are far smaller and a direct-gather or table tile needs only its 4 KiB regrouping buffer.
// Here every slot of the bilinear launch table gets the byte offset of ITS region in the workgroup's pool (stored in the slot's win_r0,
// which plain tiles do not use): regions are packed by their real size, so that a smaller pool - four workgroups per CU, measured
// -4 % (c1) / -5 % (c3) / -11 % (c5) at fixed work - holds the same windows.  One thread per workgroup.  Where four regions exceed the
// pool the largest window is demoted to the direct-gather path (same pixels) until they fit; counters[0] counts the demoted tiles,
// counters[1] the workgroups that cannot be made to fit.  dry: count only.
// Pair slots of a double-fisheye plan (round 6) are slots like any other - one eye's entry, one region; a wave R's region also carries
// its hand-over to wave L (64 x 16 packed pixels + header: PB_PAIR_LDS_BYTES), and wave L's slot learns where it is (the high half of
// its win_r0: a pool is smaller than 64 KiB).
#define PB_PAIR_LDS_BYTES (4u * (PB_PAIR_HDR + 4u))
__device__ __forceinline__ unsigned pb_bil_region_bytes(const PbTileEntry& e, int flags) {
    const unsigned floor_r = (flags & PB_TILE_PAIR_R) ? PB_PAIR_LDS_BYTES : 0u;
    unsigned need;
    if (flags & PB_TILE_SKIP) return 0u;
    if (flags & PB_TILE_BLACK) {
        need = e.bil_off >= 0 ? (unsigned)PB_DIRECT_LDS_BYTES + 16u : 0u;
    } else if (e.bil_off < 0 && (flags & PB_TILE_HALVES)) {  // two half windows, one after the other (and the direct path's buffer for frames LDS-DMA cannot address)
        const unsigned hb = (unsigned)e.bil_off & 0x7FFFFFFFu;
        const unsigned a = PB_HALF_ROWS(e.win_c0) * 16u * PB_HALF_N16(e.win_c0), b = PB_HALF_ROWS(hb) * 16u * PB_HALF_N16(hb);
        const unsigned w = a > b ? a : b;
        need = (w > (unsigned)PB_DIRECT_LDS_BYTES ? w : (unsigned)PB_DIRECT_LDS_BYTES) + 16u;
    } else if (e.bil_off >= 0 || !(flags & PB_TILE_LEAN)) {
        need = (unsigned)PB_DIRECT_LDS_BYTES + 16u;
    } else {
        const unsigned w = (unsigned)(e.win_rows * 16 * e.win_n16);  // (frames LDS-DMA cannot address send a window tile down the direct path: its buffer too)
        need = (w > (unsigned)PB_DIRECT_LDS_BYTES ? w : (unsigned)PB_DIRECT_LDS_BYTES) + 16u;
    }
    return need > floor_r ? need : floor_r;
}
// Before the regions are packed: the four one-eye slots of a virtual workgroup are dealt to its two real workgroups so that their LDS needs
// balance - (largest, smallest) and the two in between - instead of in tile order; pair slots stay where the pair layout put them.  One wave
// per virtual workgroup (an entry is 64 dwords: one per lane).  Speed only: which wave of which workgroup serves a tile moves no bit.
__global__ __launch_bounds__(64) void pb_bilinear_balance_kernel(PbTileEntry* __restrict__ ltable, unsigned n_groups) {
    if (blockIdx.x >= n_groups) return;
    PbTileEntry* slots = ltable + 4u * (size_t)blockIdx.x;
    const int lane = threadIdx.x;
    unsigned need[4];
    bool pair = false;
    for (int w = 0; w < 4; ++w) {
        need[w] = pb_bil_region_bytes(slots[w], slots[w].flags);
        pair = pair || (slots[w].flags & (PB_TILE_TWO | PB_TILE_PAIR_R)) != 0;
    }
    if (pair) return;
    int order[4] = {0, 1, 2, 3};  // by need, descending (a stable insertion sort: every lane computes the same)
    for (int i = 1; i < 4; ++i)
        for (int j = i; j > 0 && need[order[j]] > need[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    const int dest[4] = {order[0], order[3], order[1], order[2]};  // slots 0-1: largest + smallest, slots 2-3: the middle two
    if (dest[0] == 0 && dest[1] == 1 && dest[2] == 2 && dest[3] == 3) return;
    int v[4];
    for (int w = 0; w < 4; ++w) v[w] = reinterpret_cast<const int*>(slots + dest[w])[lane];
    for (int w = 0; w < 4; ++w) reinterpret_cast<int*>(slots + w)[lane] = v[w];  // (every lane holds its dword of all four entries: no hazard)
}
__global__ void pb_bilinear_pool_kernel(PbTileEntry* __restrict__ ltable, unsigned n_groups, unsigned pool_bytes, int dry, unsigned* __restrict__ counters,
                                        int waves) {
    // one thread per REAL workgroup: `waves` (4 or 2) consecutive slots of the table (a pair's L and R are neighbours: never split)
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups * (4u / (unsigned)waves)) return;
    PbTileEntry* slots = ltable + (size_t)g * (unsigned)waves;
    unsigned need[4] = {0u, 0u, 0u, 0u}, total = 0;
    bool window[4] = {false, false, false, false};
    for (int w = 0; w < waves; ++w) {
        const int f = slots[w].flags;
        need[w] = pb_bil_region_bytes(slots[w], f);
        window[w] = !(f & PB_TILE_SKIP) && slots[w].bil_off < 0 && (f & (PB_TILE_LEAN | PB_TILE_HALVES)) != 0;
        total += need[w];
    }
    unsigned demoted = 0;
    while (total > pool_bytes) {
        // (half-window tiles first: they go back to the direct path they came from, and do not count against the pool)
        int big = -1;
        bool big_half = false;
        unsigned big_need = 0;
        for (int w = 0; w < waves; ++w) {
            if (!window[w] || need[w] <= (unsigned)PB_DIRECT_LDS_BYTES + 16u) continue;
            const bool half = (slots[w].flags & PB_TILE_HALVES) != 0;
            if (big < 0 || (half && !big_half) || (half == big_half && need[w] > big_need)) { big = w; big_half = half; big_need = need[w]; }
        }
        if (big < 0) {
            atomicAdd(&counters[1], 1u);
            return;  // (the host falls back to the pool that always fits)
        }
        for (int w = 0; w < waves; ++w)
            if (w == big) {
                total -= need[w] - ((unsigned)PB_DIRECT_LDS_BYTES + 16u);
                need[w] = (unsigned)PB_DIRECT_LDS_BYTES + 16u;
                window[w] = false;
                if (!dry) slots[w].flags = (slots[w].flags & ~(PB_TILE_LEAN | PB_TILE_HALVES)) | PB_TILE_DIRECT;
            }
        if (!big_half) ++demoted;
    }
    if (demoted) atomicAdd(&counters[0], demoted);
    if (dry) return;
    unsigned off = 0;
    for (int w = 0; w < waves; ++w) {
        slots[w].win_r0 = (int)off;
        if (w > 0 && (slots[w].flags & PB_TILE_PAIR_R)) slots[w - 1].win_r0 |= (int)(off << 16);  // wave L's slot: where wave R parks its pixels
        off += need[w];
    }
}

// diagnostics (pb_plan_bilinear_tile_mix): how the bilinear mode serves the entries its waves read - counters: [0] window, [1] direct,
// [2] exact coordinate table, [3] black, [4] plain tiles evaluated on their TD3 part, [5] entries seen, [6] window tiles staged as two half windows, [7] table tiles whose taps need no guards (PB_TILE_TAB_PLAIN)
__device__ __forceinline__ void pb_bil_mix_count(const PbTileEntry& e, int f, unsigned* __restrict__ counters) {
    atomicAdd(&counters[5], 1u);
    if (e.bil_off < 0 && (f & PB_TILE_HALVES)) atomicAdd(&counters[6], 1u);
    if (e.bil_off >= 0 && (f & PB_TILE_TAB_PLAIN)) atomicAdd(&counters[7], 1u);
    if (e.bil_off >= 0) atomicAdd(&counters[2], 1u);
    else if (f & (PB_TILE_LEAN | PB_TILE_HALVES)) atomicAdd(&counters[0], 1u);  // (window tiles, whole or in two halves)
    else if (f & PB_TILE_DIRECT) atomicAdd(&counters[1], 1u);
    else if (f & PB_TILE_BLACK) atomicAdd(&counters[3], 1u);
    if (e.bil_off < 0 && (f & (PB_TILE_LEAN | PB_TILE_DIRECT)) && (f & PB_TILE_TD3)) atomicAdd(&counters[4], 1u);
}
// every slot of the launch-order table holds the entry its wave works with (a double-fisheye plan's pair layout: one slot per eye)
__global__ void pb_bilinear_mix_kernel(const PbTileEntry* __restrict__ ltable, unsigned n_slots, unsigned* __restrict__ counters) {
    const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_slots) return;
    const PbTileEntry& e = ltable[v];
    if (e.flags & PB_TILE_SKIP) return;
    pb_bil_mix_count(e, e.flags & ~(PB_TILE_TWO | PB_TILE_PAIR_R), counters);
}

// fills the coordinate table: 4 blocks per tile, every tile with a slot (pixels beyond the image repeat the edge: never stored)
template <int SRC_KIND>
__global__ __launch_bounds__(PB_BLOCK) void pb_bilinear_coord_kernel(const PbParams P, const PbTileEntry* __restrict__ table, PbBilCoord* __restrict__ bil_xy) {
    const int t = blockIdx.x >> 2;
    const int slot = table[t].bil_off;
    if (slot < 0) return;
    const int ty = t / pb_tiles_x(P), tx = t - ty * pb_tiles_x(P);
    const int local = (blockIdx.x & 3) * 256 + threadIdx.x;
    const int i = min(ty * PB_TILE + (local >> 5), P.dst.height - 1), j = min(tx * PB_TILE + (local & 31), P.dst.width - 1);
    PbCoord c = pb_dst_coord(P, i, j);
    c = pb_rotate_all(P, c);
    bil_xy[(size_t)slot * (PB_TILE * PB_TILE) + local] = pb_bil_coord_of<SRC_KIND>(P, c);
}
// How a table slot is walked (pb_bil_vals, table path): one block per tile with a slot.  From the slot's own coordinates: gx / gy = the
// mean change of the source ROW per pixel step in x / in y (neighbour pairs that are both live and less than 64 rows apart: a
// discontinuity inside the tile - the rim of the image circle - is not a gradient).  The walk runs along the direction with the smaller
// change (along y: PB_TILE_TAB_Y) and is sheared by slope = -g_along / g_across so that a half-wave follows the line of constant
// source row; the slot is rewritten in walk order, the shear goes into bil_off.  Speed only: the pixels do not depend on the walk.
// It also says whether the slot is PLAIN (PB_TILE_TAB_PLAIN, pb_tile.hpp): every live pixel's taps in columns [cmin, cmax) of the frame
// (an eye's half) without wrap, its rows at most one beyond the image (clamped), an 8-byte load at each tap's two rows inside the buffer.
// off: 2 = flag no slot plain, 4 = no walk (every slot by columns, no shear) - the diagnostic build's PB_BIL_OFF knob; 0 in the product.
// saved (may be null): the plan's copy of the certified flags (pb_classify_under_budget re-derives the live flags from it): it takes the
// walk bits of THIS pass - a deserialized plan's copy comes from the blob, its slots are written here.
// The gradients are integer sums (1/4096 px per pixel step): the same on every run, whatever order the atomics land in.
__global__ __launch_bounds__(256) void pb_bilinear_orient_kernel(PbTileEntry* __restrict__ table, PbBilCoord* __restrict__ bil_xy, int h, int w, int cmin,
                                                                 int cmax, int off_bits, int32_t* __restrict__ saved) {
    __shared__ PbBilCoord tile[PB_TILE * PB_TILE];
    __shared__ int acc[4];
    PbTileEntry* e = table + blockIdx.x;
    const int off = e->bil_off;
    if (off < 0) return;
    const int slot = off & PB_BIL_SLOT_MASK;
    PbBilCoord* t = bil_xy + (size_t)slot * (PB_TILE * PB_TILE);
    if (threadIdx.x < 4) acc[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < PB_TILE * PB_TILE; i += 256) tile[i] = t[i];
    __syncthreads();
    int sx = 0, nx = 0, sy = 0, ny = 0;  // (|d| < 2^18 over at most 992 pairs: no overflow)
    const int far = 64 << PB_BIL_SHIFT;
    const unsigned frame_bytes = 3u * (unsigned)w * (unsigned)h;
    int plain = (3u * (unsigned)w + 8u <= frame_bytes) ? 1 : 0;  // (a dead pixel loads at offset 0 of both rows)
    for (int i = threadIdx.x; i < PB_TILE * PB_TILE; i += 256) {
        const int x = i & 31, y = i >> 5;
        const PbBilCoord a = tile[i];
        if (a.y == PB_BIL_DEAD) continue;
        {
            const int r0 = a.y >> PB_BIL_SHIFT, c0 = a.x >> PB_BIL_SHIFT;
            if (r0 < -1 || r0 > h - 1 || c0 < cmin || c0 + 1 > cmax - 1 || 3u * ((unsigned)min(r0 + 1, h - 1) * (unsigned)w + (unsigned)c0) + 8u > frame_bytes) plain = 0;
        }
        if (x + 1 < PB_TILE && tile[i + 1].y != PB_BIL_DEAD) {
            const int d = tile[i + 1].y - a.y;
            if (d > -far && d < far) { sx += d; nx += 1; }
        }
        if (y + 1 < PB_TILE && tile[i + PB_TILE].y != PB_BIL_DEAD) {
            const int d = tile[i + PB_TILE].y - a.y;
            if (d > -far && d < far) { sy += d; ny += 1; }
        }
    }
    atomicAdd(&acc[0], sx); atomicAdd(&acc[1], nx); atomicAdd(&acc[2], sy); atomicAdd(&acc[3], ny);
    plain = __syncthreads_and(plain);
    const float gx = acc[1] > 0 ? (float)acc[0] / (float)acc[1] : 0.f, gy = acc[3] > 0 ? (float)acc[2] / (float)acc[3] : 0.f;
    const bool by_rows = fabsf(gy) < fabsf(gx) && !(off_bits & 4);  // the source row changes less down a column: lanes along y
    const float along = by_rows ? gy : gx, across = by_rows ? gx : gy;
    float slope = across != 0.f ? -along / across : 0.f;
    slope = fminf(fmaxf(slope, -1.9f), 1.9f);
    const int q = (off_bits & 4) ? 0 : (int)rintf(slope * 64.0f);
    const int packed = slot | ((q & 0xFF) << 20);
    __syncthreads();
    // walk order: entry [a][p] = the coordinate of pixel (p, (a + shift(p)) & 31) (by_rows: of pixel ((a + shift(p)) & 31, p))
    for (int i = threadIdx.x; i < PB_TILE * PB_TILE; i += 256) {
        const int p = i & 31, a = i >> 5;
        const int b = (a + pb_bil_slot_shift(packed, p)) & 31;
        t[i] = by_rows ? tile[p * PB_TILE + b] : tile[b * PB_TILE + p];
    }
    if (threadIdx.x == 0) {
        int f = e->flags & ~(PB_TILE_TAB_Y | PB_TILE_TAB_PLAIN);
        if (by_rows) f |= PB_TILE_TAB_Y;
        if (plain && !(off_bits & 2)) f |= PB_TILE_TAB_PLAIN;
        e->flags = f;
        e->bil_off = packed;
        if (saved) saved[blockIdx.x] = (saved[blockIdx.x] & ~(PB_TILE_TAB_Y | PB_TILE_TAB_PLAIN)) | (f & (PB_TILE_TAB_Y | PB_TILE_TAB_PLAIN));
    }
}

// ... and the fix list's: out[item * stride + offset]
template <int SRC_KIND>
__global__ __launch_bounds__(PB_BLOCK) void pb_bilinear_fix_coord_kernel(const PbParams P, const int32_t* __restrict__ fix_px, int n_fix_px,
                                                                         PbBilCoord* __restrict__ out, int stride, int offset) {
    const unsigned item = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (item >= (unsigned)n_fix_px) return;
    const unsigned p = (unsigned)fix_px[item];
    const int i = (int)(p / (unsigned)P.dst.width), j = (int)(p - (unsigned)i * (unsigned)P.dst.width);
    PbCoord c = pb_dst_coord(P, i, j);
    c = pb_rotate_all(P, c);
    out[(size_t)item * stride + offset] = pb_bil_coord_of<SRC_KIND>(P, c);
}

// behind pb_bilinear_double_hot_kernel: the plan's failed tiles and the tiles of the list above (4 blocks each), then the fix
// pixels (either eye's list), from the float64 chain
__global__ __launch_bounds__(PB_BLOCK) void pb_bilinear_double_fix_kernel(const PbParams P, const int32_t* __restrict__ fail_tiles, int n_fail_only,
                                                                          const int32_t* __restrict__ more_tiles, int n_fail_tiles,
                                                                          const int32_t* __restrict__ fix_px, int n_fix_px,
                                                                          const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n_frames,
                                                                          unsigned long long src_stride, unsigned long long dst_stride) {
    int i, j;
    // n_fail_tiles: both lists together; the first n_fail_only entries come from fail_tiles
    if ((int)blockIdx.x < 4 * n_fail_tiles) {
        const int k = blockIdx.x >> 2;
        const int t = k < n_fail_only ? fail_tiles[k] : more_tiles[k - n_fail_only];
        const int ty = t / pb_tiles_x(P), tx = t - ty * pb_tiles_x(P);
        const int local = (blockIdx.x & 3) * 256 + threadIdx.x;
        i = ty * PB_TILE + (local >> 5);
        j = tx * PB_TILE + (local & 31);
        if (i >= P.dst.height || j >= P.dst.width) return;
    } else {
        const unsigned item = (blockIdx.x - 4u * (unsigned)n_fail_tiles) * PB_BLOCK + threadIdx.x;
        if (item >= (unsigned)n_fix_px) return;
        const unsigned p = (unsigned)fix_px[item];
        i = p / (unsigned)P.dst.width;
        j = p - (unsigned)i * (unsigned)P.dst.width;
    }
    const size_t p = (size_t)i * P.dst.width + j;
    for (int f = 0; f < n_frames; ++f) {
        const unsigned v = pb_bilinear_double_px(P, i, j, src + (unsigned long long)f * src_stride);
        uint8_t* o = dst + (unsigned long long)f * dst_stride + 3 * p;
        o[0] = (uint8_t)(v & 0xFF);
        o[1] = (uint8_t)((v >> 8) & 0xFF);
        o[2] = (uint8_t)((v >> 16) & 0xFF);
    }
}
