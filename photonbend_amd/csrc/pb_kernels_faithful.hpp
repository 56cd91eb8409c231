// pb_kernels_faithful.hpp - the per-pixel FAITHFUL kernels: every output pixel runs the
// float64 chain of pb_stages.hpp.  They are (a) the reference the plan builder certifies the
// tile models against, (b) the path of double-fisheye sources, (c) the materialised-map API
// (pb_coordmap_f64 / pb_rotate_f64 / pb_sample_map_u8) and (d) PB_MODE_FAITHFUL.
#pragma once
#include <hip/hip_runtime.h>

#include "pb_params.hpp"
#include "pb_stages.hpp"

#define PB_BLOCK 256
#define PB_PX 4  // output pixels per work-item: 12 contiguous bytes = 3 dword stores

// ----------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------
template <int ROT = PB_ROT_ANY>
__device__ __forceinline__ PbCoord pb_chain(const PbParams& P, int i, int j) {
    return pb_rotate_all<ROT>(P, pb_dst_coord(P, i, j));
}

__device__ __forceinline__ unsigned pb_load_px(const uint8_t* __restrict__ src, int idx) {
    if (idx < 0) return 0u;
    const uint8_t* p = src + 3ull * (unsigned)idx;
    return (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16);
}

// Packs 4 RGB pixels (24-bit each, in the low bits of a[0..3]) into 3 dwords and
// stores them; `full` = all four pixels exist and the address is 4-byte aligned.
__device__ __forceinline__ void pb_store_px4(uint8_t* __restrict__ out, unsigned long long p0, const unsigned a[PB_PX],
                                             int count, bool aligned) {
    uint8_t* o = out + 3ull * p0;
    if (count == PB_PX && aligned) {
        uint3 v;
        v.x = a[0] | (a[1] << 24);
        v.y = (a[1] >> 8) | (a[2] << 16);
        v.z = (a[2] >> 16) | (a[3] << 8);
        uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
        o32[0] = v.x;
        o32[1] = v.y;
        o32[2] = v.z;
    } else {
        for (int k = 0; k < count; ++k) {
            o[3 * k + 0] = (uint8_t)(a[k] & 0xFF);
            o[3 * k + 1] = (uint8_t)((a[k] >> 8) & 0xFF);
            o[3 * k + 2] = (uint8_t)((a[k] >> 16) & 0xFF);
        }
    }
}

// waves per SIMD the float64 remap kernel is compiled for (its register budget).  Measured on MI355X (experiments/r4/build_f64.sh):
// four waves (128 VGPRs, a few spilled) against the compiler's own choice of three - c2 187 -> 163 us, c3 530 -> 499; the two-eye chain
// is better left alone (434 against 454)
#ifndef PB_FAITHFUL_WPE
#define PB_FAITHFUL_WPE(kind) ((kind) == PB_KIND_DOUBLE ? 1 : 4)
#endif
#ifndef PB_FAITHFUL_UNROLL  // pixels of a work-item whose float64 chains the compiler may interleave
#define PB_FAITHFUL_UNROLL 4
#endif
#define PB_PRAGMA(x) _Pragma(#x)
#define PB_UNROLL(n) PB_PRAGMA(unroll n)
template <int SRC_KIND, int ROT>
__global__ __launch_bounds__(PB_BLOCK, PB_FAITHFUL_WPE(SRC_KIND)) void pb_remap_kernel(const PbParams P, const uint8_t* __restrict__ src,
                                                            uint8_t* __restrict__ dst, int n_frames,
                                                            unsigned long long src_stride,
                                                            unsigned long long dst_stride, int aligned) {
    const unsigned total = (unsigned)P.dst.height * (unsigned)P.dst.width;
    const unsigned g = blockIdx.x * PB_BLOCK + threadIdx.x;
    const unsigned p0 = g * PB_PX;
    if (p0 >= total) return;
    const int count = (total - p0 >= PB_PX) ? PB_PX : (int)(total - p0);
    const unsigned W = (unsigned)P.dst.width;
    unsigned i = p0 / W, j = p0 - i * W;

    int idx[PB_PX];
    int idx2[PB_PX];
    double fl[PB_PX], fr[PB_PX];
    bool inv[PB_PX];
    PB_UNROLL(PB_FAITHFUL_UNROLL)
    for (int k = 0; k < PB_PX; ++k) {
        idx[k] = -1;
        idx2[k] = -1;
        fl[k] = fr[k] = 1.0;
        inv[k] = true;
        if (k < count) {
            const PbCoord c = pb_chain<ROT>(P, (int)i, (int)j);
            if (SRC_KIND == PB_KIND_PANO) {
                idx[k] = pb_src_pano_index(P, c);
            } else if (SRC_KIND == PB_KIND_CAMERA) {
                idx[k] = pb_src_camera_index(P, c);
            } else {
                const PbDoubleTap t = pb_src_double_taps(P, c);
                idx[k] = t.il;
                idx2[k] = t.ir;
                fl[k] = t.fl;
                fr[k] = t.fr;
                inv[k] = c.inv;
            }
            if (++j == W) {
                j = 0;
                ++i;
            }
        }
    }
    for (int f = 0; f < n_frames; ++f) {
        const uint8_t* s = src + (unsigned long long)f * src_stride;
        uint8_t* d = dst + (unsigned long long)f * dst_stride;
        unsigned a[PB_PX];
#pragma unroll
        for (int k = 0; k < PB_PX; ++k) {
            if (SRC_KIND == PB_KIND_DOUBLE) {
                const unsigned l = pb_load_px(s, idx[k]);
                const unsigned r = pb_load_px(s, idx2[k]);
                unsigned v = 0;
                if (!inv[k]) {  // final_image[invalid_map] = 0, projection.py:460
                    v = pb_blend_u8(l & 0xFF, r & 0xFF, fl[k], fr[k]) |
                        (pb_blend_u8((l >> 8) & 0xFF, (r >> 8) & 0xFF, fl[k], fr[k]) << 8) |
                        (pb_blend_u8((l >> 16) & 0xFF, (r >> 16) & 0xFF, fl[k], fr[k]) << 16);
                }
                a[k] = v;
            } else {
                a[k] = pb_load_px(s, idx[k]);
            }
        }
        pb_store_px4(d, p0, a, count, aligned != 0);
    }
}

template <int SRC_KIND, int ROT>
__global__ __launch_bounds__(PB_BLOCK) void pb_index_kernel(const PbParams P, int32_t* __restrict__ out,
                                                            double* __restrict__ wout) {
    const unsigned total = (unsigned)P.dst.height * (unsigned)P.dst.width;
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    const unsigned W = (unsigned)P.dst.width;
    const unsigned i = p / W, j = p - i * W;
    const PbCoord c = pb_chain<ROT>(P, (int)i, (int)j);
    if (SRC_KIND == PB_KIND_PANO) {
        out[p] = pb_src_pano_index(P, c);
    } else if (SRC_KIND == PB_KIND_CAMERA) {
        out[p] = pb_src_camera_index(P, c);
    } else {
        const PbDoubleTap t = pb_src_double_taps(P, c);
        out[p] = t.il;
        out[(size_t)total + p] = t.ir;
        if (wout) {
            wout[p] = t.fl;
            wout[(size_t)total + p] = t.fr;
        }
    }
}

__global__ __launch_bounds__(PB_BLOCK) void pb_coordmap_kernel(const PbParams P, double* __restrict__ out) {
    const unsigned total = (unsigned)P.dst.height * (unsigned)P.dst.width;
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    const unsigned W = (unsigned)P.dst.width;
    const unsigned i = p / W, j = p - i * W;
    const PbCoord c = pb_dst_coord(P, (int)i, (int)j);
    double* o = out + 3ull * p;
    o[0] = c.lat;
    o[1] = c.lon;
    o[2] = c.inv ? 1.0 : 0.0;
}

struct PbMat {
    double m[9];
};

__global__ __launch_bounds__(PB_BLOCK) void pb_rotate_kernel(const PbMat R, double* __restrict__ in,
                                                             double* __restrict__ out, unsigned total) {
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    double* a = in + 3ull * p;
    PbCoord c;
    c.inv = a[2] != 0.0;  // NaN counts as invalid, rotation.py:118
    if (c.inv) {
        a[0] = 0.0;  // the reference zeroes the CALLER's map, rotation.py:119-125
        a[1] = 0.0;
    }
    c.lat = a[0];
    c.lon = a[1];
    c = pb_rotate(R.m, c);
    double* o = out + 3ull * p;
    o[0] = c.lat;
    o[1] = c.lon;
    o[2] = c.inv ? 1.0 : 0.0;
}

template <int SRC_KIND>
__global__ __launch_bounds__(PB_BLOCK) void pb_sample_map_kernel(const PbParams P, double* __restrict__ map,
                                                                 unsigned total, const uint8_t* __restrict__ src,
                                                                 uint8_t* __restrict__ dst) {
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    double* a = map + 3ull * p;
    PbCoord c;
    c.inv = a[2] != 0.0;
    if (SRC_KIND == PB_KIND_PANO && c.inv) {
        a[0] = 0.0;  // polar_map[invalid_map] = 0 writes through the view, projection.py:534-536
        a[1] = 0.0;
    }
    c.lat = a[0];
    c.lon = a[1];
    unsigned v;
    if (SRC_KIND == PB_KIND_PANO) {
        v = pb_load_px(src, pb_src_pano_index(P, c));
    } else if (SRC_KIND == PB_KIND_CAMERA) {
        v = pb_load_px(src, pb_src_camera_index(P, c));
    } else {
        const PbDoubleTap t = pb_src_double_taps(P, c);
        const unsigned l = pb_load_px(src, t.il), r = pb_load_px(src, t.ir);
        v = 0;
        if (!c.inv)
            v = pb_blend_u8(l & 0xFF, r & 0xFF, t.fl, t.fr) |
                (pb_blend_u8((l >> 8) & 0xFF, (r >> 8) & 0xFF, t.fl, t.fr) << 8) |
                (pb_blend_u8((l >> 16) & 0xFF, (r >> 16) & 0xFF, t.fl, t.fr) << 16);
    }
    uint8_t* o = dst + 3ull * p;
    o[0] = (uint8_t)(v & 0xFF);
    o[1] = (uint8_t)((v >> 8) & 0xFF);
    o[2] = (uint8_t)((v >> 16) & 0xFF);
}

// ---- generic images and user lenses (drop-in completeness; off the hot path) ------------------------------------
// map -> integer source index (+ blend weights for a double source): the sampling half of process_coordinate_map
// (projection.py:197-260, :408-462, :515-547) without the gather, so that images of any channel count / sample
// width can be gathered by index (pb_gather_px_kernel) - the reference fancy-indexes whatever array it is given.
// dist_l / dist_r (optional): the per-pixel fisheye radius forward_lens(latitude) * f_distance evaluated by the
// HOST for a Lens made of Python callables (lens.py:48-64); they replace the built-in forward lens.
__device__ __forceinline__ bool pb_src_camera_pos_d(double dist, double lon, int h, int w, double cy, double cx, int& py, int& px) {
    double sl, cl;
    pb_expi_np(lon, &sl, &cl);  // np.exp(lon * 1j)
    const double re = cl * dist, im = sl * dist;
    const long long y = pb_cvt_i64((im * -1.0) + cy);
    const long long x = pb_cvt_i64(re + cx);
    if (y >= h || y < 0 || x >= w || x < 0) return false;
    py = (int)y;
    px = (int)x;
    return true;
}

template <int SRC_KIND>
__global__ __launch_bounds__(PB_BLOCK) void pb_index_from_map_kernel(const PbParams P, double* __restrict__ map, unsigned total,
                                                                     const double* __restrict__ dist_l, const double* __restrict__ dist_r,
                                                                     int32_t* __restrict__ out, double* __restrict__ wout) {
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    double* a = map + 3ull * p;
    PbCoord c;
    c.inv = a[2] != 0.0;
    if (SRC_KIND == PB_KIND_PANO && c.inv) {
        a[0] = 0.0;  // polar_map[invalid_map] = 0 writes through the view, projection.py:534-536
        a[1] = 0.0;
    }
    c.lat = a[0];
    c.lon = a[1];
    if (SRC_KIND == PB_KIND_PANO) {
        out[p] = pb_src_pano_index(P, c);
    } else if (SRC_KIND == PB_KIND_CAMERA) {
        if (dist_l) {
            int py, px;
            const bool ok = pb_src_camera_pos_d(dist_l[p], c.lon, P.src.height, P.src.width, P.src_cy, P.src_cx, py, px);
            out[p] = (ok && !c.inv) ? py * P.src.width + px : -1;
        } else {
            out[p] = pb_src_camera_index(P, c);
        }
    } else {
        PbDoubleTap t;
        if (dist_l) {
            int py, px;
            bool ok = pb_src_camera_pos_d(dist_l[p], c.lon, P.src.height, P.src_eye_w, P.src_cy, P.src_cx, py, px);
            t.il = (ok && !c.inv) ? py * P.src.width + px : -1;
            ok = pb_src_camera_pos_d(dist_r[p], c.lon, P.src.height, P.src_eye_w_right, P.src_cy, P.src_cx_r, py, px);
            t.ir = (ok && !c.inv) ? py * P.src.width + (P.src_eye_w + (P.src_eye_w_right - 1 - px)) : -1;
            t.fl = pb_merge_factor(P, c.lat);
            t.fr = pb_merge_factor(P, (c.lat * -1.0) + PB_PI);
        } else {
            t = pb_src_double_taps(P, c);
        }
        out[p] = t.il;
        out[(size_t)total + p] = t.ir;
        if (wout) {
            wout[p] = t.fl;
            wout[(size_t)total + p] = t.fr;
        }
    }
}

// out[p] = idx[p] < 0 ? zeros : src[idx[p]], bpp bytes per pixel (new_image = image[positions]; new_image[bad] = 0,
// projection.py:234-243, :545-546) - any channel count, any sample width
__global__ __launch_bounds__(PB_BLOCK) void pb_gather_px_kernel(const int32_t* __restrict__ idx, const uint8_t* __restrict__ src,
                                                                uint8_t* __restrict__ dst, unsigned long long total, int bpp) {
    const unsigned long long p = (unsigned long long)blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    const int id = idx[p];
    uint8_t* o = dst + p * (unsigned)bpp;
    if (id < 0) {
        for (int b = 0; b < bpp; ++b) o[b] = 0;
    } else {
        const uint8_t* s = src + (unsigned long long)(unsigned)id * (unsigned)bpp;
        for (int b = 0; b < bpp; ++b) o[b] = s[b];
    }
}

// the double-fisheye blend for any channel count and 8 / 16-bit samples: (left * fl + right * fr).astype(np.uint8)
// per channel (projection.py:447-460) - the output is uint8 whatever the input width, like the reference's
template <typename SAMPLE>
__global__ __launch_bounds__(PB_BLOCK) void pb_gather_blend_kernel(const int32_t* __restrict__ idx2, const double* __restrict__ w2,
                                                                   const SAMPLE* __restrict__ src, uint8_t* __restrict__ dst,
                                                                   unsigned long long total, int channels) {
    const unsigned long long p = (unsigned long long)blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    const int il = idx2[p], ir = idx2[total + p];
    const double fl = w2[p], fr = w2[total + p];
    uint8_t* o = dst + p * (unsigned)channels;
    for (int c = 0; c < channels; ++c) {
        const double l = il < 0 ? 0.0 : (double)src[(unsigned long long)(unsigned)il * (unsigned)channels + c];
        const double r = ir < 0 ? 0.0 : (double)src[(unsigned long long)(unsigned)ir * (unsigned)channels + c];
        // final_image[invalid_map] = 0 (projection.py:460): an invalid destination pixel has both taps -1 -> 0 * f = 0
        // (NaN / inf factors give NaN, which the cast turns into 0 as well)
        o[c] = (uint8_t)pb_cvt_u8(l * fl + r * fr);
    }
}

__device__ __forceinline__ uint32_t pb_mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

__global__ __launch_bounds__(PB_BLOCK) void pb_synth_kernel(uint8_t* __restrict__ out, int height, int width,
                                                            uint32_t fkey, int circle_mask) {
    const unsigned total = (unsigned)height * (unsigned)width;
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    const unsigned r = p / (unsigned)width, c = p - r * (unsigned)width;
    unsigned keep = 1;
    if (circle_mask) {
        const long long ys = 2ll * r + 1 - height;
        long long xs, d;
        if (circle_mask == 1) {
            xs = 2ll * c + 1 - width;
            d = height < width ? height : width;
        } else {
            const int half = width / 2;
            xs = 2ll * (c % (unsigned)half) + 1 - half;
            d = height < half ? height : half;
        }
        keep = (ys * ys + xs * xs <= d * d) ? 1u : 0u;
    }
    const uint32_t base = (r * 0x85EBCA6Bu) ^ (c * 0xC2B2AE35u) ^ fkey;
    uint8_t* o = out + 3ull * p;
#pragma unroll
    for (uint32_t ch = 0; ch < 3; ++ch) o[ch] = (uint8_t)(keep * (pb_mix32(base ^ (ch * 0x27D4EB2Fu)) & 0xFFu));
}



// ---- map_projection (projection.py:550-599): coordinate map -> colour map ------------------------
// pass 1: min / max of the latitude over the valid pixels (invalid lat/lon zeroed in place, :563-567);
// doubles are reduced through their order-preserving uint64 image
__device__ __forceinline__ unsigned long long pb_f64_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double pb_key_f64(unsigned long long k) {
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

// ws[0] = min key, ws[1] = max key, ws[2] = number of valid pixels (caller zero-initialises as {~0, 0, 0})
__global__ __launch_bounds__(PB_BLOCK) void pb_mapproj_minmax_kernel(double* __restrict__ map, unsigned total,
                                                                     unsigned long long* __restrict__ ws) {
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    unsigned long long kmin = ~0ull, kmax = 0ull, cnt = 0;
    if (p < total) {
        double* a = map + 3ull * p;
        if (a[2] != 0.0) {
            a[0] = 0.0;
            a[1] = 0.0;
        } else {
            kmin = kmax = pb_f64_key(a[0]);   // NaN latitudes order above +inf / below -inf: np.min/np.max would
            cnt = 1;                          // return NaN; the facade rejects maps whose valid latitudes hold NaN
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o), c = __shfl_xor(cnt, o);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
        cnt += c;
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        atomicMin(&ws[0], kmin);
        atomicMax(&ws[1], kmax);
        atomicAdd(&ws[2], cnt);
    }
}

__global__ __launch_bounds__(PB_BLOCK) void pb_mapproj_colour_kernel(const double* __restrict__ map, unsigned total,
                                                                     const unsigned long long* __restrict__ ws,
                                                                     uint8_t* __restrict__ out) {
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    const double mn = pb_key_f64(ws[0]), mx = pb_key_f64(ws[1]);
    const double factor = 255.0 / (mx - mn);            // rgb_range / min_max_distance, :573-574
    const double* a = map + 3ull * p;
    const bool invalid = a[2] != 0.0;
    double d = a[0];
    if (!invalid) {
        d = d - mn;                                      // :576
        d = d * factor;                                  // :577
    }
    const double g = (255.0 / (PB_PI * 2)) * a[1];      // :583-584
    uint8_t* o = out + 3ull * p;
    o[0] = (uint8_t)pb_cvt_u8(rint(d));                  // np.round = half-to-even; astype(uint8) wraps
    o[1] = (uint8_t)pb_cvt_u8(rint(g));
    o[2] = invalid ? 255 : 0;
}
