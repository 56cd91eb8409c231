// photonbend_hip.hip - HIP kernels (gfx950 / CDNA4) and the C ABI declared in
// include/photonbend_hip.h.  One work-item owns PB_PX consecutive output pixels:
// inverse projection -> rotation(s) -> forward projection -> integer source
// index, once; then one gather + store per frame of the batch.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared ...
// (-ffp-contract=off is REQUIRED: the reference rounds every multiply and add
// separately; fused multiply-adds appear only where written as fma()).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <new>
#include <string>

#include "pb_params.hpp"
#include "pb_stages.hpp"
#include "pb_tile.hpp"

#define PB_BLOCK 256
#define PB_PX 4  // output pixels per work-item: 12 contiguous bytes = 3 dword stores

struct pb_plan {
    PbParams P;
    int mode;                     // PB_MODE_AUTO / PB_MODE_FAITHFUL / PB_MODE_FAST
    long long certify_mismatches; // -1 = not certified (no device at creation)
    long long exact_pixels = -1;  // pixels the fast path sent through the faithful chain (certification run)
    long long modelled_tiles = -1;
};

static thread_local std::string g_err;
static int pb_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define PB_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return pb_fail(PB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

// ----------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------
__device__ __forceinline__ PbCoord pb_chain(const PbParams& P, int i, int j) {
    PbCoord c = pb_dst_coord(P, i, j);
    for (int k = 0; k < P.n_rot; ++k) c = pb_rotate(P.R[k], c);
    return c;
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

template <int SRC_KIND>
__global__ __launch_bounds__(PB_BLOCK) void pb_remap_kernel(const PbParams P, const uint8_t* __restrict__ src,
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
#pragma unroll
    for (int k = 0; k < PB_PX; ++k) {
        idx[k] = -1;
        idx2[k] = -1;
        fl[k] = fr[k] = 1.0;
        inv[k] = true;
        if (k < count) {
            const PbCoord c = pb_chain(P, (int)i, (int)j);
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

template <int SRC_KIND>
__global__ __launch_bounds__(PB_BLOCK) void pb_index_kernel(const PbParams P, int32_t* __restrict__ out,
                                                            double* __restrict__ wout) {
    const unsigned total = (unsigned)P.dst.height * (unsigned)P.dst.width;
    const unsigned p = blockIdx.x * PB_BLOCK + threadIdx.x;
    if (p >= total) return;
    const unsigned W = (unsigned)P.dst.width;
    const unsigned i = p / W, j = p - i * W;
    const PbCoord c = pb_chain(P, (int)i, (int)j);
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


// ----------------------------------------------------------------------------------
// tile kernels (pano / camera sources): one wave per 32x32 tile, 4 tiles per block
// ----------------------------------------------------------------------------------
#define PB_TILE_WAVES 4
typedef unsigned pb_u32x3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ unsigned pb_load_px32(const uint8_t* __restrict__ src, int idx) {
    if (idx < 0) return 0u;
    unsigned v;
    __builtin_memcpy(&v, src + 3ull * (unsigned)idx, 4);  // unaligned dword; the source carries >= 1 byte of tail slack
    return v & 0xFFFFFFu;
}

// tile id -> (tile x, tile y); a block of 4 waves takes a 2x2 group of tiles (64x64 px)
__device__ __forceinline__ bool pb_tile_origin(const PbParams& P, int wave, int& X0, int& Y0) {
    const int gx = (P.dst.width + 2 * PB_TILE - 1) / (2 * PB_TILE);
    const int by = blockIdx.x / gx, bx = blockIdx.x - by * gx;
    X0 = (2 * bx + (wave & 1)) * PB_TILE;
    Y0 = (2 * by + (wave >> 1)) * PB_TILE;
    return X0 < P.dst.width && Y0 < P.dst.height;
}

// OUT 0: gather + store frames; OUT 1: write the int32 index map; OUT 2: certify (count pixels
// whose fast-path index differs from the faithful one)
template <int SRC_KIND, int OUT>
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_tile_kernel(const PbParams P, const uint8_t* __restrict__ src,
                                                                      uint8_t* __restrict__ dst, int n_frames,
                                                                      unsigned long long src_stride,
                                                                      unsigned long long dst_stride, int tail_slack,
                                                                      int32_t* __restrict__ idx_out,
                                                                      unsigned long long* __restrict__ counter) {
    __shared__ PbWaveLds lds[PB_TILE_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int X0, Y0;
    if (!pb_tile_origin(P, wave, X0, Y0)) return;  // wave-uniform; no workgroup barriers below
    PbWaveLds& L = lds[wave];
    unsigned n_exact = 0;
    const bool modelled = pb_tile_indices<SRC_KIND>(P, L, lane, X0, Y0, P.fast_tiles != 0, OUT == 2 ? &n_exact : nullptr);

    // gather-phase ownership: lane -> 4 consecutive px (x = 4*xg..) in rows yb + 8*jr
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
    if (OUT == 2) {
        unsigned bad = 0;
#pragma unroll 1
        for (int jr = 0; jr < 4; ++jr) {
            const int y = Y0 + yb + 8 * jr;
#pragma unroll 1
            for (int k = 0; k < 4; ++k)
                if (y < H && x + k < W) bad += (pb_exact_index<SRC_KIND>(P, y, x + k) != id[jr][k]);
        }
        if (bad) atomicAdd(counter, (unsigned long long)bad);
        if (n_exact) atomicAdd(counter + 1, (unsigned long long)n_exact);  // pixels that took the faithful chain
        if (lane == 0) atomicAdd(counter + 2, modelled ? 1ull : 0ull);      // tiles with an accepted model
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
                // the 4-byte read of the very last source pixel would touch one byte past the
                // frame: only allowed when the caller's buffer has slack
                a[k] = (!tail_slack && (unsigned)v == last_px) ? pb_load_px(s, v) : pb_load_px32(s, v);
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

// plan creation: bisection for the validity thresholds with the exact predicate
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

// ----------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------
static bool pb_end_ok(const pb_proj* p, std::string& why) {
    if (!p) {
        why = "null pb_proj";
        return false;
    }
    if (p->kind < PB_KIND_CAMERA || p->kind > PB_KIND_PANO) {
        why = "pb_proj.kind out of range";
        return false;
    }
    if (p->kind != PB_KIND_PANO && (p->lens < PB_LENS_EQUIDISTANT || p->lens > PB_LENS_THOBY)) {
        why = "pb_proj.lens out of range";
        return false;
    }
    if (p->height < 1 || p->width < 1 || (long long)p->height * p->width > 0x7FFFFFFFll / 4) {
        why = "pb_proj height/width out of range (need 1 <= h*w < 2^29)";
        return false;
    }
    if (p->kind == PB_KIND_DOUBLE && (p->width & 1)) {
        why = "a double-fisheye frame needs an even width";
        return false;
    }
    return true;
}

static PbEnd pb_to_end(const pb_proj* p) {
    PbEnd e;
    e.kind = p->kind;
    e.lens = (p->kind == PB_KIND_PANO) ? PB_LENS_EQUIDISTANT : p->lens;
    e.height = p->height;
    e.width = p->width;
    e.fov = p->fov;
    e.f_distance = p->f_distance;
    return e;
}

static inline unsigned pb_blocks(unsigned long long items) { return (unsigned)((items + PB_BLOCK - 1) / PB_BLOCK); }


static inline unsigned pb_tile_blocks(const PbParams& P) {
    const unsigned gx = (P.dst.width + 2 * PB_TILE - 1) / (2 * PB_TILE), gy = (P.dst.height + 2 * PB_TILE - 1) / (2 * PB_TILE);
    return gx * gy;
}

template <int OUT>
static void pb_launch_tiles(const PbParams& P, const uint8_t* src, uint8_t* dst, int n_frames, unsigned long long ss,
                            unsigned long long ds, int32_t* idx_out, unsigned long long* counter, hipStream_t st) {
    const dim3 grid(pb_tile_blocks(P)), block(64 * PB_TILE_WAVES);
    if (P.src.kind == PB_KIND_PANO)
        hipLaunchKernelGGL((pb_tile_kernel<PB_KIND_PANO, OUT>), grid, block, 0, st, P, src, dst, n_frames, ss, ds, 0, idx_out, counter);
    else
        hipLaunchKernelGGL((pb_tile_kernel<PB_KIND_CAMERA, OUT>), grid, block, 0, st, P, src, dst, n_frames, ss, ds, 0, idx_out, counter);
}

// Runs once per plan on the current device (synchronously, default stream):
//  1. validity thresholds of a camera / double destination by bisection with the exact predicate;
//  2. certification: the fast tile path's index map is compared with the faithful one for every
//     pixel; one differing pixel disables the fast path for this plan.
static int pb_plan_prepare_on_device(pb_plan* pl) {
    PbParams& P = pl->P;
    long long* scratch = nullptr;
    PB_HIP(hipMalloc((void**)&scratch, 8 * sizeof(long long)));
    int rc = PB_OK;
    do {
        if (P.dst.kind != PB_KIND_PANO) {
            hipLaunchKernelGGL(pb_threshold_kernel, dim3(1), dim3(2), 0, 0, P, scratch);
            long long thr[4];
            if (hipMemcpy(thr, scratch, sizeof(thr), hipMemcpyDeviceToHost) != hipSuccess) { rc = PB_ERR_HIP; break; }
            P.inv_lo[0] = thr[0]; P.inv_hi[0] = thr[1];
            P.inv_lo[1] = thr[2]; P.inv_hi[1] = thr[3];
        }
        P.thresholds_ready = 1;
        if (P.src.kind != PB_KIND_DOUBLE) {
            unsigned long long* counter = reinterpret_cast<unsigned long long*>(scratch + 4);
            if (hipMemset(counter, 0, 3 * sizeof(unsigned long long)) != hipSuccess) { rc = PB_ERR_HIP; break; }
            // the fast path needs 32-bit squares of the doubled pixel offsets and >= 14 fraction bits
            const int maxd = P.dst.width > P.dst.height ? P.dst.width : P.dst.height;
            if (maxd > 16384 || P.fx_shift < 14) break;
            P.fast_tiles = 1;
            pb_launch_tiles<2>(P, nullptr, nullptr, 0, 0, 0, nullptr, counter, 0);
            unsigned long long res[3] = {0, 0, 0};
            if (hipMemcpy(res, counter, sizeof(res), hipMemcpyDeviceToHost) != hipSuccess) { rc = PB_ERR_HIP; break; }
            pl->certify_mismatches = (long long)res[0];
            pl->exact_pixels = (long long)res[1];
            pl->modelled_tiles = (long long)res[2];
            if (res[0]) P.fast_tiles = 0;
        }
    } while (0);
    if (rc != PB_OK) g_err = std::string("plan preparation on device failed: ") + hipGetErrorString(hipGetLastError());
    (void)hipFree(scratch);
    return rc;
}

static PbParams pb_effective_params(const pb_plan* plan) {
    PbParams P = plan->P;
    if (plan->mode == PB_MODE_FAITHFUL) P.fast_tiles = 0;
    if (plan->mode == PB_MODE_FAST && P.src.kind != PB_KIND_DOUBLE && P.thresholds_ready && P.fx_shift >= 14 &&
        P.dst.width <= 16384 && P.dst.height <= 16384)
        P.fast_tiles = 1;
    return P;
}

extern "C" {

int pb_abi_version(void) { return PB_ABI_VERSION; }
const char* pb_last_error(void) { return g_err.c_str(); }

int pb_init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return pb_fail(PB_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return pb_fail(PB_ERR_INVALID, "device ordinal out of range");
    PB_HIP(hipSetDevice(device));
    return PB_OK;
}
int pb_shutdown(void) { return PB_OK; }

int pb_device_name(char* buf, size_t buflen) {
    if (!buf || !buflen) return pb_fail(PB_ERR_INVALID, "null buffer");
    int dev = 0;
    PB_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PB_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return PB_OK;
}

int pb_plan_create(const pb_proj* dst, const double* rot3x3, int n_rot, const pb_proj* src, pb_plan** out) {
    std::string why;
    if (!out) return pb_fail(PB_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!pb_end_ok(dst, why) || !pb_end_ok(src, why)) return pb_fail(PB_ERR_INVALID, why);
    if (n_rot < 0 || n_rot > PB_MAX_ROTATIONS) return pb_fail(PB_ERR_INVALID, "n_rot outside [0, PB_MAX_ROTATIONS]");
    if (n_rot > 0 && !rot3x3) return pb_fail(PB_ERR_INVALID, "null rotation matrices");
    pb_plan* pl = new (std::nothrow) pb_plan();
    if (!pl) return pb_fail(PB_ERR_INVALID, "out of host memory");
    memset(&pl->P, 0, sizeof(PbParams));
    pl->P.dst = pb_to_end(dst);
    pl->P.src = pb_to_end(src);
    pl->P.n_rot = n_rot;
    for (int k = 0; k < n_rot; ++k)
        for (int e = 0; e < 9; ++e) pl->P.R[k][e] = rot3x3[9 * k + e];
    pb_derive(pl->P);
    pl->mode = PB_MODE_AUTO;
    pl->certify_mismatches = -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    (void)hipGetLastError();
    if (ndev > 0) {
        const int rc = pb_plan_prepare_on_device(pl);
        if (rc != PB_OK) {
            delete pl;
            return rc;
        }
    }
    *out = pl;
    return PB_OK;
}

void pb_plan_destroy(pb_plan* plan) { delete plan; }

int pb_plan_dst_shape(const pb_plan* plan, int* height, int* width) {
    if (!plan || !height || !width) return pb_fail(PB_ERR_INVALID, "null argument");
    *height = plan->P.dst.height;
    *width = plan->P.dst.width;
    return PB_OK;
}
int pb_plan_src_shape(const pb_plan* plan, int* height, int* width) {
    if (!plan || !height || !width) return pb_fail(PB_ERR_INVALID, "null argument");
    *height = plan->P.src.height;
    *width = plan->P.src.width;
    return PB_OK;
}

int pb_remap_u8(const pb_plan* plan, const uint8_t* src_dev, uint8_t* dst_dev, int n_frames, size_t src_frame_stride,
                size_t dst_frame_stride, void* stream) {
    if (!plan || !src_dev || !dst_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (n_frames < 0) return pb_fail(PB_ERR_INVALID, "negative frame count");
    if (n_frames == 0) return PB_OK;
    const PbParams& P = plan->P;
    const unsigned long long npx = (unsigned long long)P.dst.height * P.dst.width;
    if (!src_frame_stride) src_frame_stride = 3ull * P.src.height * P.src.width;
    if (!dst_frame_stride) dst_frame_stride = 3ull * npx;
    if (dst_frame_stride < 3ull * npx) return pb_fail(PB_ERR_INVALID, "dst_frame_stride smaller than a frame");
    hipStream_t st = (hipStream_t)stream;
    if (P.src.kind != PB_KIND_DOUBLE) {
        pb_launch_tiles<0>(pb_effective_params(plan), src_dev, dst_dev, n_frames, src_frame_stride, dst_frame_stride, nullptr,
                           nullptr, st);
    } else {
        const int aligned = (((uintptr_t)dst_dev | dst_frame_stride) & 3u) == 0;
        const unsigned blocks = pb_blocks((npx + PB_PX - 1) / PB_PX);
        hipLaunchKernelGGL(pb_remap_kernel<PB_KIND_DOUBLE>, dim3(blocks), dim3(PB_BLOCK), 0, st, P, src_dev, dst_dev,
                           n_frames, src_frame_stride, dst_frame_stride, aligned);
    }
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_index_map_i32(const pb_plan* plan, int32_t* idx_dev, double* weights_dev, void* stream) {
    if (!plan || !idx_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    const PbParams& P = plan->P;
    hipStream_t st = (hipStream_t)stream;
    if (P.src.kind != PB_KIND_DOUBLE) {
        pb_launch_tiles<1>(pb_effective_params(plan), nullptr, nullptr, 0, 0, 0, idx_dev, nullptr, st);
    } else {
        const unsigned blocks = pb_blocks((unsigned long long)P.dst.height * P.dst.width);
        hipLaunchKernelGGL(pb_index_kernel<PB_KIND_DOUBLE>, dim3(blocks), dim3(PB_BLOCK), 0, st, P, idx_dev, weights_dev);
    }
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_plan_set_mode(pb_plan* plan, int mode) {
    if (!plan) return pb_fail(PB_ERR_INVALID, "null argument");
    if (mode < PB_MODE_AUTO || mode > PB_MODE_FAST) return pb_fail(PB_ERR_INVALID, "mode out of range");
    plan->mode = mode;
    return PB_OK;
}

int pb_plan_info(const pb_plan* plan, int* fast_path_enabled, long long* certify_mismatches, long long* thresholds4,
                 long long* tile_stats3) {
    if (!plan) return pb_fail(PB_ERR_INVALID, "null argument");
    if (fast_path_enabled) *fast_path_enabled = pb_effective_params(plan).fast_tiles;
    if (certify_mismatches) *certify_mismatches = plan->certify_mismatches;
    if (tile_stats3) {
        const PbParams& P = plan->P;
        tile_stats3[0] = (long long)((P.dst.width + PB_TILE - 1) / PB_TILE) * ((P.dst.height + PB_TILE - 1) / PB_TILE);
        tile_stats3[1] = plan->modelled_tiles;
        tile_stats3[2] = plan->exact_pixels;
    }
    if (thresholds4) {
        thresholds4[0] = plan->P.inv_lo[0];
        thresholds4[1] = plan->P.inv_hi[0];
        thresholds4[2] = plan->P.inv_lo[1];
        thresholds4[3] = plan->P.inv_hi[1];
    }
    return PB_OK;
}

int pb_coordmap_f64(const pb_proj* dst, double* map_dev, void* stream) {
    std::string why;
    if (!map_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (!pb_end_ok(dst, why)) return pb_fail(PB_ERR_INVALID, why);
    PbParams P;
    memset(&P, 0, sizeof(P));
    P.dst = pb_to_end(dst);
    P.src = P.dst;
    pb_derive(P);
    const unsigned blocks = pb_blocks((unsigned long long)P.dst.height * P.dst.width);
    hipLaunchKernelGGL(pb_coordmap_kernel, dim3(blocks), dim3(PB_BLOCK), 0, (hipStream_t)stream, P, map_dev);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_rotate_f64(const double* rot3x3, double* map_in_dev, double* map_out_dev, int height, int width, void* stream) {
    if (!rot3x3 || !map_in_dev || !map_out_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (map_in_dev == map_out_dev) return pb_fail(PB_ERR_INVALID, "map_out_dev must not alias map_in_dev");
    if (height < 1 || width < 1 || (long long)height * width > 0x7FFFFFFFll / 4)
        return pb_fail(PB_ERR_INVALID, "map size out of range");
    PbMat R;
    for (int e = 0; e < 9; ++e) R.m[e] = rot3x3[e];
    const unsigned total = (unsigned)height * (unsigned)width;
    hipLaunchKernelGGL(pb_rotate_kernel, dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, (hipStream_t)stream, R, map_in_dev,
                       map_out_dev, total);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_sample_map_u8(const pb_proj* src, double* map_dev, int height, int width, const uint8_t* src_dev,
                     uint8_t* dst_dev, void* stream) {
    std::string why;
    if (!map_dev || !src_dev || !dst_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (!pb_end_ok(src, why)) return pb_fail(PB_ERR_INVALID, why);
    if (height < 1 || width < 1 || (long long)height * width > 0x7FFFFFFFll / 4)
        return pb_fail(PB_ERR_INVALID, "map size out of range");
    PbParams P;
    memset(&P, 0, sizeof(P));
    P.src = pb_to_end(src);
    P.dst = P.src;
    P.dst.height = height;
    P.dst.width = width;
    pb_derive(P);
    const unsigned total = (unsigned)height * (unsigned)width;
    hipStream_t st = (hipStream_t)stream;
    switch (P.src.kind) {
        case PB_KIND_PANO:
            hipLaunchKernelGGL(pb_sample_map_kernel<PB_KIND_PANO>, dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, st, P,
                               map_dev, total, src_dev, dst_dev);
            break;
        case PB_KIND_CAMERA:
            hipLaunchKernelGGL(pb_sample_map_kernel<PB_KIND_CAMERA>, dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, st, P,
                               map_dev, total, src_dev, dst_dev);
            break;
        default:
            hipLaunchKernelGGL(pb_sample_map_kernel<PB_KIND_DOUBLE>, dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, st, P,
                               map_dev, total, src_dev, dst_dev);
    }
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_synth_frame_u8(uint8_t* frame_dev, int height, int width, uint32_t frame, uint32_t seed, int circle_mask,
                      void* stream) {
    if (!frame_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (height < 1 || width < 1 || (long long)height * width > 0x7FFFFFFFll / 4)
        return pb_fail(PB_ERR_INVALID, "frame size out of range");
    if (circle_mask < 0 || circle_mask > 2) return pb_fail(PB_ERR_INVALID, "circle_mask must be 0, 1 or 2");
    const uint32_t fkey = (frame * 0x9E3779B1u) ^ seed;
    const unsigned total = (unsigned)height * (unsigned)width;
    hipLaunchKernelGGL(pb_synth_kernel, dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, (hipStream_t)stream, frame_dev,
                       height, width, fkey, circle_mask);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// ---- plumbing ----------------------------------------------------------------------
int pb_malloc(void** dev_ptr, size_t bytes) {
    if (!dev_ptr) return pb_fail(PB_ERR_INVALID, "null argument");
    PB_HIP(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return PB_OK;
}
int pb_free(void* dev_ptr) {
    if (dev_ptr) PB_HIP(hipFree(dev_ptr));
    return PB_OK;
}
int pb_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
    PB_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return PB_OK;
}
int pb_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream) {
    PB_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PB_OK;
}
int pb_memset(void* dst_dev, int value, size_t bytes, void* stream) {
    PB_HIP(hipMemsetAsync(dst_dev, value, bytes, (hipStream_t)stream));
    return PB_OK;
}
int pb_stream_create(void** stream) {
    if (!stream) return pb_fail(PB_ERR_INVALID, "null argument");
    hipStream_t s;
    PB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return PB_OK;
}
int pb_stream_destroy(void* stream) {
    if (stream) PB_HIP(hipStreamDestroy((hipStream_t)stream));
    return PB_OK;
}
int pb_stream_sync(void* stream) {
    PB_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PB_OK;
}
int pb_event_create(void** event) {
    if (!event) return pb_fail(PB_ERR_INVALID, "null argument");
    hipEvent_t e;
    PB_HIP(hipEventCreate(&e));
    *event = (void*)e;
    return PB_OK;
}
int pb_event_destroy(void* event) {
    if (event) PB_HIP(hipEventDestroy((hipEvent_t)event));
    return PB_OK;
}
int pb_event_record(void* event, void* stream) {
    PB_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return PB_OK;
}
int pb_event_sync(void* event) {
    PB_HIP(hipEventSynchronize((hipEvent_t)event));
    return PB_OK;
}
int pb_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!ms) return pb_fail(PB_ERR_INVALID, "null argument");
    PB_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PB_OK;
}

}  // extern "C"
