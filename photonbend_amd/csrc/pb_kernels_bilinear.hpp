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
// reference's factors - pb_bilinear_double_hot_kernel from the per-eye tile models of the nearest mode's plan (round 3),
// pb_bilinear_double_kernel (float64 coordinates per pixel) for plans without tile tables.
//
//   pb_bilinear_hot_kernel   modelled tiles: float32 tile models give f (error ~1e-5 px, no fix list needed: there
//                            is no truncation to protect), exact integer validity thresholds; LEAN tiles read their
//                            four taps from the LDS window of the nearest mode, other tiles gather them directly
//   pb_bilinear_fix_kernel   failed tiles (seam, pole, centre, no model): float64 faithful chain per pixel
#pragma once
#include "pb_kernels_tile.hpp"

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

// one bilinear pixel from the wave's LDS window: the four taps around (sy, sx) = f - 0.5 in WINDOW coordinates
// (LEAN tiles carry one texel of margin on every side - exactly the taps' reach - and lie strictly inside the
// image, so nothing is clamped or wrapped)
__device__ __forceinline__ unsigned pb_bilinear_lds(const unsigned* win, float sy, float sx, unsigned pitch, unsigned a0) {
    const float fy0 = floorf(sy), fx0 = floorf(sx);
    const float ty = sy - fy0, tx = sx - fx0;
    const unsigned l00 = __umul24((unsigned)(int)fy0, pitch) + __umul24((unsigned)(int)fx0, 3u) + a0;
    const unsigned l10 = l00 + pitch;
    // two horizontally adjacent taps = 6 consecutive bytes: three aligned dwords cover them
    const unsigned w0 = win[l00 >> 2], w1 = win[(l00 >> 2) + 1], w2 = win[(l00 >> 2) + 2];
    const unsigned v0 = win[l10 >> 2], v1 = win[(l10 >> 2) + 1], v2 = win[(l10 >> 2) + 2];
    // the right-hand tap starts 3 bytes on: in the same dword pair only when the left tap is dword-aligned
    const bool c0 = (l00 & 3u) != 0, c1 = (l10 & 3u) != 0;
    const unsigned p00 = __builtin_amdgcn_alignbyte(w1, w0, l00);
    const unsigned p01 = __builtin_amdgcn_alignbyte(c0 ? w2 : w1, c0 ? w1 : w0, l00 + 3u);
    const unsigned p10 = __builtin_amdgcn_alignbyte(v1, v0, l10);
    const unsigned p11 = __builtin_amdgcn_alignbyte(c1 ? v2 : v1, c1 ? v1 : v0, l10 + 3u);
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

// The four channels' arithmetic of one pixel: taps p00 p01 / p10 p11 (low 3 bytes), weights (tx, ty) -> packed RGB
__device__ __forceinline__ unsigned pb_bilinear_mix(unsigned p00, unsigned p01, unsigned p10, unsigned p11, float tx, float ty) {
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

// One wave per tile, launched like pb_hot_win_kernel: `table` is the plan's LAUNCH-ORDER table (the entry says which
// tile it is), frames of a batch are a grid dimension.  LEAN tiles take their taps from the LDS window the nearest mode
// stages (same plan, same LDS-DMA loads); DIRECT tiles gather them from the frame, two horizontally adjacent taps (6
// consecutive bytes) per 8-byte load; other tiles tap by tap.
template <int SRC_KIND>
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_bilinear_hot_kernel(const PbParams P, const PbTileEntry* __restrict__ table,
                                                                              const uint8_t* __restrict__ src,
                                                                              uint8_t* __restrict__ dst, const unsigned groups_per_frame,
                                                                              unsigned long long src_stride,
                                                                              unsigned long long dst_stride, int windows) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned wg = blockIdx.x;
    if (wg >= groups_per_frame) {  // a batch: which frame
        const unsigned f = wg / groups_per_frame;
        wg -= f * groups_per_frame;
        src += (unsigned long long)f * src_stride;
        dst += (unsigned long long)f * dst_stride;
    }
    PbTileEntry entry;
    const unsigned vslot = (unsigned)__builtin_amdgcn_readfirstlane((int)(wg * (unsigned)PB_TILE_WAVES + (unsigned)wave));  // four waves per workgroup: the launch table's slot order
    pb_load_entry(table + vslot, entry);
    const PbTileEntry* __restrict__ e = &entry;
    const int flags = e->flags;
    if (flags & (PB_TILE_SKIP | PB_TILE_FAILED | PB_TILE_COARSE)) return;  // failed tiles, tiles whose model is too coarse to interpolate at: pb_bilinear_fix_kernel
    const int tx = e->tile_xy & 0xFFFF, ty = (int)((unsigned)e->tile_xy >> 16);
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    const int xg = lane & 7, yb = lane >> 3;
    const int W = P.dst.width, H = P.dst.height;
    const int h = P.src.height, w = P.src.width;
    unsigned* win = pb_wave_window(P, wave);
    const unsigned rowbytes = 3u * (unsigned)w;
    const unsigned frame_bytes = rowbytes * (unsigned)h;
    const unsigned safe_len = frame_bytes & ~15u;
    const uint8_t* s = src;
    uint8_t* d = dst;
    const int x = X0 + 4 * xg;
    if ((flags & PB_TILE_LEAN) && windows) {  // (windows == 0: frames LDS-DMA cannot address)
        const unsigned pitch = 16u * (unsigned)e->win_n16, a0 = (unsigned)e->win_a0;
        const unsigned gbase = (unsigned)e->anchor_r * rowbytes + 3u * (unsigned)e->anchor_c;
        // the model is evaluated in the order the direct-gather path below uses for this tile (column-first when the source row
        // changes least along x): the LDS budget moves a tile between the two paths and must not move a bit of its coordinates
        pb_f2 fv[4][4];
        if (fabsf(e->c[1][0]) <= fabsf(e->c[5][0])) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pb_f2 b[5];
                pb_collapse_col(e, 4 * xg + k, b);
#pragma unroll
                for (int jr = 0; jr < 4; ++jr) fv[jr][k] = pb_eval_row(b, pb_tile_coord(yb + 8 * jr));
            }
        } else {
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                pb_f2 a[5];
                pb_collapse_row(e, yb + 8 * jr, a);
#pragma unroll
                for (int k = 0; k < 4; ++k) fv[jr][k] = pb_eval_row(a, pb_tile_coord(4 * xg + k));
            }
        }
        asm volatile("" ::: "memory");
        pb_issue_window_loads(s, win, lane, gbase, rowbytes, e->win_rows, e->win_n16, safe_len);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pb_wave_sync();
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            unsigned a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = pb_bilinear_lds(win, fv[jr][k].x - 0.5f, fv[jr][k].y - 0.5f, pitch, a0);
            // LEAN tiles lie fully inside the image
            const unsigned long long off = 3ull * ((unsigned long long)(Y0 + yb + 8 * jr) * W + x);
            if ((((uintptr_t)d + off) & 3u) == 0) {
                pb_store3<SRC_KIND == PB_KIND_CAMERA>(pb_pack_px4(a[0], a[1], a[2], a[3]), d + off);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    d[off + 3 * k + 0] = (uint8_t)(a[k] & 0xFF);
                    d[off + 3 * k + 1] = (uint8_t)((a[k] >> 8) & 0xFF);
                    d[off + 3 * k + 2] = (uint8_t)((a[k] >> 16) & 0xFF);
                }
            }
        }
        return;
    }
    if ((flags & (PB_TILE_LEAN | PB_TILE_DIRECT)) && !(flags & PB_TILE_MASKED)) {  // (MASKED: invalid pixels inside - the guarded path below)
        // The four taps straight from the frame, unguarded (the tile's bounding box, margin texel included, lies inside the
        // frame with room for the last 4-byte read).  wide: an 8-byte load takes both taps of a row - allowed when even the
        // box's last tap has 8 bytes of frame behind it.  The gathers run like the nearest mode's (pb_win_tile, DIRECT): lane =
        // pixel column (or row), 16 pixels down the other direction, sheared along the line of constant source row, so that one
        // load instruction touches few lines whatever the tile's orientation in the source; the blended pixels are regrouped
        // for the 12-byte stores through the wave's LDS window.  Either evaluation order of the model is certified.
        // (round 3: c2 118 -> see experiments/README.md; the row-group order it replaces issued 4 dependent rounds of loads
        // whose 64 lanes touched up to 64 lines each.)
        const unsigned gbase = (unsigned)e->anchor_r * rowbytes + 3u * (unsigned)e->anchor_c;
        const bool wide = gbase + (unsigned)(e->win_rows - 1) * rowbytes + 3u * (unsigned)(e->win_cols - 1) + 8u <= frame_bytes;
        const bool along_x = fabsf(e->c[1][0]) <= fabsf(e->c[5][0]);  // |d row / du| <= |d row / dv|
        const int p = lane & 31, hh = lane >> 5;
        const float num = along_x ? e->c[1][0] : e->c[5][0], den = along_x ? e->c[5][0] : e->c[1][0];
        const float slope = (den != 0.0f) ? -num / den : 0.0f;
        const int shift = (int)rintf(slope * ((float)p - 15.5f));
        pb_f2 cf[5];
        if (along_x) pb_collapse_col(e, p, cf);
        else pb_collapse_row(e, p, cf);
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // 8 pixels' loads in flight together (16 x 8 bytes per lane)
            unsigned long long t8[8][2];
            unsigned t[8][4];
            float wy[8], wx[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int n = 8 * half + m, q = (2 * n + hh + shift) & 31;
                const pb_f2 fv = pb_eval_row(cf, pb_tile_coord(q));
                const float sy = fv.x - 0.5f, sx = fv.y - 0.5f;
                const float fy0 = floorf(sy), fx0 = floorf(sx);
                wy[m] = sy - fy0;
                wx[m] = sx - fx0;
                const unsigned g = gbase + (unsigned)(int)fy0 * rowbytes + __umul24((unsigned)(int)fx0, 3u);
                if (wide) {
                    __builtin_memcpy(&t8[m][0], s + g, 8);
                    __builtin_memcpy(&t8[m][1], s + g + rowbytes, 8);
                } else {
                    __builtin_memcpy(&t[m][0], s + g, 4);
                    __builtin_memcpy(&t[m][1], s + g + 3u, 4);
                    __builtin_memcpy(&t[m][2], s + g + rowbytes, 4);
                    __builtin_memcpy(&t[m][3], s + g + rowbytes + 3u, 4);
                }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int n = 8 * half + m, q = (2 * n + hh + shift) & 31;
                if (wide) {
                    t[m][0] = (unsigned)t8[m][0];
                    t[m][1] = (unsigned)(t8[m][0] >> 24);
                    t[m][2] = (unsigned)t8[m][1];
                    t[m][3] = (unsigned)(t8[m][1] >> 24);
                }
                // park as [y][x] with a 33-dword pitch (the lane's pixel is (p, q) or (q, p))
                win[along_x ? q * 33 + p : p * 33 + q] = pb_bilinear_mix(t[m][0], t[m][1], t[m][2], t[m][3], wx[m], wy[m]);
            }
        }
        pb_wave_sync();
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            const unsigned* r = win + (yb + 8 * jr) * 33 + 4 * xg;
            const unsigned long long off = 3ull * ((unsigned long long)(Y0 + yb + 8 * jr) * W + x);
            if ((((uintptr_t)d + off) & 3u) == 0) {
                pb_store3<SRC_KIND == PB_KIND_CAMERA>(pb_pack_px4(r[0], r[1], r[2], r[3]), d + off);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    d[off + 3 * k + 0] = (uint8_t)(r[k] & 0xFF);
                    d[off + 3 * k + 1] = (uint8_t)((r[k] >> 8) & 0xFF);
                    d[off + 3 * k + 2] = (uint8_t)((r[k] >> 16) & 0xFF);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        const int y = Y0 + yb + 8 * jr;
        PbRowModel R;
        pb_model_row(P, e, X0, Y0, yb + 8 * jr, 4 * xg, R);
        unsigned a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const pb_f2 fv = pb_eval_row(R.a, pb_tile_coord(4 * xg + k));
            unsigned px = 0;
            // black where the nearest mode is black: invalid destination pixel / camera position outside the image
            const float ay = (float)R.anchor_r + fv.x, ax = (float)R.anchor_c + fv.y;
            bool live = !(flags & PB_TILE_BLACK) && !pb_row_px_invalid(R, k) && y < H && x + k < W;
            if (SRC_KIND == PB_KIND_CAMERA) live = live && ay >= 0.0f && ay < (float)h && ax >= 0.0f && ax < (float)w;
            if (live) px = pb_bilinear_taps<SRC_KIND>(P, s, fv.x - 0.5f, fv.y - 0.5f, R.anchor_r, R.anchor_c);
            a[k] = px;
        }
        if (y < H) {
            const unsigned long long off = 3ull * ((unsigned long long)y * W + x);
            if (x + 3 < W && (((uintptr_t)d + off) & 3u) == 0) {
                pb_store3<SRC_KIND == PB_KIND_CAMERA>(pb_pack_px4(a[0], a[1], a[2], a[3]), d + off);
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
    for (int k = 0; k < P.n_rot; ++k) c = pb_rotate(P.R[k], c);
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
    pb_sincos_cr(lon, &sl, &cl);  // np.exp(lon * 1j)
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
    for (int k = 0; k < P.n_rot; ++k) c = pb_rotate(P.R[k], c);
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

// ---- double-fisheye source through the per-eye tile models (round 3) -----------------------------------------------------
// The bilinear taps of ONE eye for a lane's 16 pixels (4 consecutive pixels x 4 rows, as everywhere), from the eye's tile
// entry of the nearest mode's plan.  Coordinates are the model's, in FRAME space: the right eye's model already runs over
// the mirrored half (its column coordinate is w - x_eye), and bilinear interpolation commutes with the mirror, so the taps
// are the frame's texels floor(s), floor(s) + 1 of either eye - clamped to the eye's own columns [cmin, cmax) and to the
// frame's rows, as the definition clamps them to the eye's image.  Three paths, like the single-source kernel:
//   window   LEAN tile whose box (margin texel = the taps' reach included) lies inside the eye: the nearest mode's LDS window;
//   direct   plain tile inside the eye: unguarded 8-byte loads of both taps of a row;
//   clamped  a plain tile at the edge of its eye: taps clamped one by one.
// Tiles that are not plain for an eye that sees them (partly outside the eye's image, partly invalid) are NOT served here: the
// nearest mode's certification lets their models be sloppy where almost no pixel samples (the truncation absorbs it), which a
// bilinear tap does not forgive - such tiles go to the float64 pass whole (pb_bilinear_double_fix_kernel, the plan's list).
__device__ __forceinline__ unsigned pb_bilinear_taps_clamped(const uint8_t* __restrict__ s, int h, int w, int cmin, int cmax, float sy, float sx, int by, int bx) {
    const float fy0 = floorf(sy), fx0 = floorf(sx);
    const float ty = sy - fy0, tx = sx - fx0;
    int r0 = by + (int)fy0, c0 = bx + (int)fx0;
    int r1 = r0 + 1, c1 = c0 + 1;
    r0 = min(max(r0, 0), h - 1);
    r1 = min(max(r1, 0), h - 1);
    c0 = min(max(c0, cmin), cmax - 1);
    c1 = min(max(c1, cmin), cmax - 1);
    return pb_bilinear_mix(pb_load_px(s, r0 * w + c0), pb_load_px(s, r0 * w + c1), pb_load_px(s, r1 * w + c0), pb_load_px(s, r1 * w + c1), tx, ty);
}

template <int EYE>
__device__ __forceinline__ void pb_bilinear_eye_vals(const PbParams& P, const PbTileEntry* __restrict__ e, const int flags, const int X0, const int Y0,
                                                     const int lane, unsigned* win, const int windows, const uint8_t* __restrict__ s, unsigned v[16]) {
    const int xg = lane & 7, yb = lane >> 3;
    const int h = P.src.height, w = P.src.width;
    const unsigned rowbytes = 3u * (unsigned)w, frame_bytes = rowbytes * (unsigned)h, safe_len = frame_bytes & ~15u;
    int cmin, cmax;
    pb_src_col_range<EYE>(P, cmin, cmax);
    if (!(flags & (PB_TILE_LEAN | PB_TILE_DIRECT))) {  // BLACK (callers never pass anything else that is not plain)
#pragma unroll
        for (int n = 0; n < 16; ++n) v[n] = 0u;
        return;
    }
    const bool inside = e->win_c0 >= cmin && e->win_c0 + e->win_cols <= cmax;
    if (inside && (flags & PB_TILE_LEAN) && windows) {
        const unsigned pitch = 16u * (unsigned)e->win_n16, a0 = (unsigned)e->win_a0;
        const unsigned gbase = (unsigned)e->anchor_r * rowbytes + 3u * (unsigned)e->anchor_c;
        // the model is evaluated in the order the direct-gather path below uses for this tile (column-first when the source row
        // changes least along x): the LDS budget moves a tile between the two paths and must not move a bit of its coordinates
        pb_f2 fv[4][4];
        if (fabsf(e->c[1][0]) <= fabsf(e->c[5][0])) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pb_f2 b[5];
                pb_collapse_col(e, 4 * xg + k, b);
#pragma unroll
                for (int jr = 0; jr < 4; ++jr) fv[jr][k] = pb_eval_row(b, pb_tile_coord(yb + 8 * jr));
            }
        } else {
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                pb_f2 a[5];
                pb_collapse_row(e, yb + 8 * jr, a);
#pragma unroll
                for (int k = 0; k < 4; ++k) fv[jr][k] = pb_eval_row(a, pb_tile_coord(4 * xg + k));
            }
        }
        asm volatile("" ::: "memory");
        pb_issue_window_loads(s, win, lane, gbase, rowbytes, e->win_rows, e->win_n16, safe_len);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pb_wave_sync();
#pragma unroll
        for (int jr = 0; jr < 4; ++jr)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[jr * 4 + k] = pb_bilinear_lds(win, fv[jr][k].x - 0.5f, fv[jr][k].y - 0.5f, pitch, a0);
        pb_wave_sync();  // the window may be refilled (the other eye)
        return;
    }
    if (inside) {
        const unsigned gbase = (unsigned)e->anchor_r * rowbytes + 3u * (unsigned)e->anchor_c;
        const bool wide = gbase + (unsigned)(e->win_rows - 1) * rowbytes + 3u * (unsigned)(e->win_cols - 1) + 8u <= frame_bytes;
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            pb_f2 c[5];
            pb_collapse_row(e, yb + 8 * jr, c);
            unsigned long long t8[4][2];
            unsigned t[4][4];
            float wy[4], wx[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const pb_f2 fv = pb_eval_row(c, pb_tile_coord(4 * xg + k));
                const float sy = fv.x - 0.5f, sx = fv.y - 0.5f;
                const float fy0 = floorf(sy), fx0 = floorf(sx);
                wy[k] = sy - fy0;
                wx[k] = sx - fx0;
                const unsigned g = gbase + (unsigned)(int)fy0 * rowbytes + __umul24((unsigned)(int)fx0, 3u);
                if (wide) {
                    __builtin_memcpy(&t8[k][0], s + g, 8);
                    __builtin_memcpy(&t8[k][1], s + g + rowbytes, 8);
                } else {
                    __builtin_memcpy(&t[k][0], s + g, 4);
                    __builtin_memcpy(&t[k][1], s + g + 3u, 4);
                    __builtin_memcpy(&t[k][2], s + g + rowbytes, 4);
                    __builtin_memcpy(&t[k][3], s + g + rowbytes + 3u, 4);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (wide) {
                    t[k][0] = (unsigned)t8[k][0];
                    t[k][1] = (unsigned)(t8[k][0] >> 24);
                    t[k][2] = (unsigned)t8[k][1];
                    t[k][3] = (unsigned)(t8[k][1] >> 24);
                }
                v[jr * 4 + k] = pb_bilinear_mix(t[k][0], t[k][1], t[k][2], t[k][3], wx[k], wy[k]);
            }
        }
        return;
    }
    // a plain tile at the edge of its eye: every pixel is live (that is what plain means), the taps are clamped one by one
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        pb_f2 c[5];
        pb_collapse_row(e, yb + 8 * jr, c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const pb_f2 fv = pb_eval_row(c, pb_tile_coord(4 * xg + k));
            v[jr * 4 + k] = pb_bilinear_taps_clamped(s, h, w, cmin, cmax, fv.x - 0.5f, fv.y - 0.5f, e->anchor_r, e->anchor_c);
        }
    }
}

// One wave per tile, launched like pb_hot_double_kernel over the plan's launch-order table (frames of a batch: a grid
// dimension).  A tile that sees ONE eye with weight exactly 1 (PB_TILE_SOLO: its slot carries the live eye's entry) is that
// eye's bilinear sample; a two-eye tile samples the left eye, then the right eye (through the same LDS window), and blends
// with the tile's weight class like the nearest mode - UNIT: the integer sum, ROW: the row table, LAT: the stored latitudes.
// Failed tiles and the plan's fix pixels take the float64 chain (pb_bilinear_double_fix_kernel, behind this launch).
template <int WMODE>
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_bilinear_double_hot_kernel(const PbParams P, const PbTileEntry* __restrict__ table_l,
                                                                                     const PbTileEntry* __restrict__ table_r,
                                                                                     const PbTileEntry* __restrict__ ltable,
                                                                                     const PbSepRow* __restrict__ rows,
                                                                                     const double* __restrict__ lat_tab,
                                                                                     const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                                     const unsigned groups_per_frame, unsigned long long src_stride,
                                                                                     unsigned long long dst_stride, int windows) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned wg = blockIdx.x;
    if (wg >= groups_per_frame) {
        const unsigned f = wg / groups_per_frame;
        wg -= f * groups_per_frame;
        src += (unsigned long long)f * src_stride;
        dst += (unsigned long long)f * dst_stride;
    }
    const unsigned vslot = (unsigned)__builtin_amdgcn_readfirstlane((int)(wg * (unsigned)PB_TILE_WAVES + (unsigned)wave));
    PbTileEntry entry;
    pb_load_entry(ltable + vslot, entry);
    if (entry.flags & PB_TILE_SKIP) return;
    const int tx = entry.tile_xy & 0xFFFF, ty = (int)((unsigned)entry.tile_xy >> 16);
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    const int xg = lane & 7, yb = lane >> 3;
    const int W = P.dst.width, H = P.dst.height;
    unsigned* win = pb_wave_window(P, wave, 8);
    unsigned a[16];
    if (entry.flags & PB_TILE_SOLO) {
        if (entry.flags & PB_TILE_COARSE) return;  // (on the plan's float64 list)
        if (entry.flags & PB_TILE_EYE_R)
            pb_bilinear_eye_vals<PB_KIND_EYE_R>(P, &entry, entry.flags, X0, Y0, lane, win, windows, src, a);
        else
            pb_bilinear_eye_vals<PB_KIND_EYE_L>(P, &entry, entry.flags, X0, Y0, lane, win, windows, src, a);
    } else {
        const size_t tile = (size_t)ty * pb_tiles_x(P) + tx;
        pb_load_entry(table_l + tile, entry);
        const int fl0 = entry.flags, lat_slot = entry.aux_off;
        const int served = PB_TILE_LEAN | PB_TILE_DIRECT | PB_TILE_BLACK;
        if ((fl0 & (PB_TILE_FAILED | PB_TILE_COARSE)) || !(fl0 & served)) return;  // failed, coarse, or not plain for the left eye: the float64 pass's (pb_bilinear_tile_list_kernel)
        unsigned al[16];
        pb_bilinear_eye_vals<PB_KIND_EYE_L>(P, &entry, fl0, X0, Y0, lane, win, windows, src, al);
        pb_load_entry(table_r + tile, entry);
        if (!(entry.flags & served) || (entry.flags & PB_TILE_COARSE)) return;
        pb_bilinear_eye_vals<PB_KIND_EYE_R>(P, &entry, entry.flags, X0, Y0, lane, win, windows, src, a);
        const bool by_row = WMODE == 1 && (fl0 & PB_TILE_W_ROW) != 0;
        const bool by_lat = WMODE == 2 && (fl0 & PB_TILE_W_LAT) != 0;
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
                    const double t = lat_tab[(size_t)lat_slot * PB_LAT_TILE_DOUBLES + (yb + 8 * jr) * PB_TILE + 4 * xg + k];
                    wl = pb_merge_factor(P, t);
                    wr = pb_merge_factor(P, (t * -1.0) + PB_PI);
                }
                a[jr * 4 + k] = pb_sep_blend(al[jr * 4 + k], a[jr * 4 + k], wl, wr);
            }
        }
    }
    const int x = X0 + 4 * xg;
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        const int y = Y0 + yb + 8 * jr;
        if (y >= H) continue;
        const unsigned long long off = 3ull * ((unsigned long long)y * W + x);
        if (x + 3 < W && (((uintptr_t)dst + off) & 3u) == 0) {
            pb_store3<false>(pb_pack_px4(a[jr * 4], a[jr * 4 + 1], a[jr * 4 + 2], a[jr * 4 + 3]), dst + off);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x + k < W) {
                    dst[off + 3 * k + 0] = (uint8_t)(a[jr * 4 + k] & 0xFF);
                    dst[off + 3 * k + 1] = (uint8_t)((a[jr * 4 + k] >> 8) & 0xFF);
                    dst[off + 3 * k + 2] = (uint8_t)((a[jr * 4 + k] >> 16) & 0xFF);
                }
        }
    }
}

// Plan creation (double-fisheye plans): the tiles pb_bilinear_double_hot_kernel does not serve - an eye sees the tile but the
// tile is not plain for it - listed once (the budget only moves tiles between LEAN and DIRECT, never in or out of this list).
__global__ void pb_bilinear_tile_list_kernel(const PbTileEntry* __restrict__ table_l, const PbTileEntry* __restrict__ table_r, unsigned n_tiles,
                                             int32_t* __restrict__ list, unsigned* __restrict__ count) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    // (single-source plans: table_r == nullptr, and only COARSE tiles are listed - the single-source bilinear kernel serves
    // generic tiles itself)
    const int fl = table_l[t].flags, fr = table_r ? table_r[t].flags : 0;
    const int served = PB_TILE_LEAN | PB_TILE_DIRECT | PB_TILE_BLACK;
    if ((fl | fr) & PB_TILE_FAILED) return;  // on the fail list already
    const bool coarse = ((fl | fr) & PB_TILE_COARSE) != 0;
    if (coarse || (table_r && (!(fl & served) || !(fr & served)))) list[atomicAdd(count, 1u)] = (int32_t)t;
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
