// pb_kernels_double.hpp - fast path for DOUBLE-fisheye sources (DoubleCameraImage.process_coordinate_map,
// projection.py:408-462): the Gear-360 stitch (BASELINE c5) and every other chain that ends in a
// side-by-side double fisheye, rotated or not.
//
// A double source is two camera sources - the left eye, and the right eye looking backwards and mirrored -
// whose samples are blended per pixel: out = (l * fl + r * fr).astype(uint8), with fl = fr = 1.0 outside the
// merge band (projection.py:440-444).  The plan therefore carries TWO tile tables, one per eye, built and
// certified by the same kernels as a camera source's (pb_kernels_tile.hpp with SRC_KIND = PB_KIND_EYE_L /
// PB_KIND_EYE_R: per-tile float32 models, source windows, every pixel's index compared with the faithful
// one), plus a weight class per tile, also certified against the faithful chain for every pixel
// (pb_double_pair_kernel):
//   UNIT  every pixel of the tile has fl == fr == 1.0 exactly: the blend is the integer l + r (mod 256);
//   ROW   unrotated panorama destination: the factors depend on the output row only and come from the
//         separable path's row table (exact float64, pb_kernels_sep.hpp);
//   LAT   any other tile that touches the merge band (rotated chains, fisheye destinations): the plan stores the
//         faithful float64 latitude of each of its pixels (8 KiB per tile); the factors are the reference's own
//         function of that latitude, evaluated per pixel - exact;
//   (a tile beyond the latitude table's capacity is listed as failed and recomputed by the faithful chain).
// pb_hot_double_kernel: one wave per 32x32 tile; both eyes' windows are pulled into the wave's LDS by
// LDS-DMA while the two models are evaluated, pixels are gathered from LDS (or straight from the frame for
// sparse windows), blended and stored.  What the models miss is looked up, not recomputed: the plan stores the
// faithful taps and factors of every fix-list pixel and of every pixel of a failed tile (PbDoubleFix,
// pb_double_tables_kernel), and the hot waves copy through them.  One launch per frame; the union is
// bit-identical to the faithful kernel for every pixel, by construction.
#pragma once
#include "pb_kernels_sep.hpp"
#include "pb_kernels_tile.hpp"

static_assert(PB_TILE_W_UNIT_BIT == 64, "PB_TILE_W_UNIT");
#define PB_TILE_W_UNIT 64   // (left-eye entry) blend factors are exactly 1.0 for every pixel of the tile
#define PB_TILE_W_ROW 128   // (left-eye entry) blend factors come from the row table
#define PB_TILE_W_LAT 256   // (left-eye entry) blend factors from the stored per-pixel latitudes (slot aux_off)
#define PB_LAT_TILE_DOUBLES (PB_TILE * PB_TILE)

struct PbTileCtx {
    unsigned rowbytes, frame_bytes, safe_len;
    int lane, xg, yb, W, H, X0, Y0;
    float u[4];
};

struct PbDesc {  // the scalar part of a tile entry
    int flags, anchor_r, anchor_c, win_rows, win_r0, win_c0, win_cols, win_n16, win_a0;
};

__device__ __forceinline__ PbDesc pb_load_desc(const PbTileEntry* __restrict__ e) {
    PbDesc d;
    d.flags = e->flags; d.anchor_r = e->anchor_r; d.anchor_c = e->anchor_c; d.win_rows = e->win_rows;
    d.win_r0 = e->win_r0; d.win_c0 = e->win_c0; d.win_cols = e->win_cols; d.win_n16 = e->win_n16; d.win_a0 = e->win_a0;
    return d;
}

// A tile entry held as ONE vector register (lane i = dword i of the 256-byte entry, one coalesced load): fields and
// coefficients come out with v_readlane (constant lane -> scalar register).  The double kernel needs two entries per tile:
// 128 scalar registers do not exist, and scalar loads on demand put a memory round trip into the middle of the model math
// (measured on c5, experiments/diag_trace.py: 2.85 us of a wave's 8.3 us went there).
#define PB_E_DWORD(field) (offsetof(PbTileEntry, field) / 4)
__device__ __forceinline__ int pb_lane_i(unsigned v, int dword) { return __builtin_amdgcn_readlane((int)v, dword); }
__device__ __forceinline__ float pb_lane_f(unsigned v, int dword) { return __int_as_float(__builtin_amdgcn_readlane((int)v, dword)); }

__device__ __forceinline__ PbDesc pb_desc_of_lanes(unsigned v) {
    PbDesc d;
    d.flags = pb_lane_i(v, PB_E_DWORD(flags)); d.anchor_r = pb_lane_i(v, PB_E_DWORD(anchor_r)); d.anchor_c = pb_lane_i(v, PB_E_DWORD(anchor_c));
    d.win_rows = pb_lane_i(v, PB_E_DWORD(win_rows)); d.win_r0 = pb_lane_i(v, PB_E_DWORD(win_r0)); d.win_c0 = pb_lane_i(v, PB_E_DWORD(win_c0));
    d.win_cols = pb_lane_i(v, PB_E_DWORD(win_cols)); d.win_n16 = pb_lane_i(v, PB_E_DWORD(win_n16)); d.win_a0 = pb_lane_i(v, PB_E_DWORD(win_a0));
    return d;
}

// The 25 coefficient pairs of a lane-held entry as scalars: ONE v_readlane each.  (Reading them inside pb_collapse_row_lanes, once
// per row group, left 200 v_readlane per eye in the ISA - 70 % of the collapse's instructions; the two-eye tiles of c5 spent
// 5 us of their 13.5 us life in the model math: experiments/session_r3_ab.sh.)
struct PbCoefs {
    float c[50];
};
__device__ __forceinline__ PbCoefs pb_coefs_of_lanes(unsigned v) {
    PbCoefs K;
    const int c0 = PB_E_DWORD(c);
#pragma unroll
    for (int n = 0; n < 50; ++n) K.c[n] = pb_lane_f(v, c0 + n);
    return K;
}

// pb_collapse_row on a lane-held entry: the same packed FMAs on the same values, hence the same bits
__device__ __forceinline__ void pb_collapse_row_lanes(const PbCoefs& K, int y, pb_f2 a[5]) {
    const float t = pb_tile_coord(y);
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        pb_f2 s = {K.c[2 * (20 + n)], K.c[2 * (20 + n) + 1]};
#pragma unroll
        for (int m = 3; m >= 0; --m) {
            const pb_f2 cm = {K.c[2 * (m * 5 + n)], K.c[2 * (m * 5 + n) + 1]};
            s = pb_fma2(s, t, cm);
        }
        a[n] = s;
    }
}

#define PB_D_PLAIN(flags) ((flags) & (PB_TILE_LEAN | PB_TILE_DIRECT))
#define PB_D_GENERIC(flags) (!((flags) & (PB_TILE_LEAN | PB_TILE_DIRECT | PB_TILE_BLACK | PB_TILE_FAILED)))

__device__ __forceinline__ int pb_d_lean_bytes(const PbDesc& D) { return (D.flags & PB_TILE_LEAN) ? D.win_rows * 16 * D.win_n16 : 0; }

// The wave's LDS window is shared by the two eyes: LEAN windows take what they need (the plan made sure
// they fit together), generic windows share the rest.
__device__ __forceinline__ void pb_d_budgets(const PbParams& P, const PbDesc& L, const PbDesc& R, int& budget_l, int& budget_r) {
    const int need_l = pb_d_lean_bytes(L), need_r = pb_d_lean_bytes(R);
    const int rest = P.win_budget - need_l - need_r;
    const bool gen_l = PB_D_GENERIC(L.flags), gen_r = PB_D_GENERIC(R.flags);
    const int share = ((gen_l && gen_r) ? rest / 2 : rest) & ~15;
    budget_l = (L.flags & PB_TILE_LEAN) ? need_l : (gen_l ? share : 0);
    budget_r = (R.flags & PB_TILE_LEAN) ? need_r : (gen_r ? share : 0);
}

// window geometry of a generic tile under an LDS budget (the rule of pb_win_tile)
__device__ __forceinline__ void pb_d_generic_window(const PbDesc& D, const PbTileCtx& C, int budget, int& nrows, int& n16, unsigned& gbase) {
    nrows = D.win_rows;
    gbase = (unsigned)D.win_r0 * C.rowbytes + 3u * (unsigned)D.win_c0;
    n16 = 1;
    if (nrows > 0) {
        n16 = (3 * D.win_cols + 15 + 1 + 15) >> 4;
        if (n16 > 64) n16 = 64;
        const int cap = budget / (16 * n16);
        if (nrows > cap) nrows = cap;
    }
}

// issue the LDS-DMA loads of one eye's source window (nothing for BLACK / DIRECT tiles)
__device__ __forceinline__ void pb_d_issue(const PbDesc& D, const PbTileCtx& C, int budget, const uint8_t* __restrict__ s, unsigned* win) {
    if (D.flags & PB_TILE_LEAN) {
        const unsigned gbase = (unsigned)D.anchor_r * C.rowbytes + 3u * (unsigned)D.anchor_c;
        pb_issue_window_loads(s, win, C.lane, gbase, C.rowbytes, D.win_rows, D.win_n16, C.safe_len);
    } else if (PB_D_GENERIC(D.flags)) {
        int nrows, n16;
        unsigned gbase;
        pb_d_generic_window(D, C, budget, nrows, n16, gbase);
        if (nrows > 0) pb_issue_window_loads(s, win, C.lane, gbase, C.rowbytes, nrows, n16, C.safe_len);
    }
}

// one eye's model math -> per-pixel addresses q[jr * 4 + k]
//   LEAN: byte address in the eye's LDS window; DIRECT: byte offset in the frame; generic: (row << 16 | col) or -1
template <int SRC_KIND>
__device__ __forceinline__ void pb_d_math(const PbParams& P, const PbDesc& D, const PbTileCtx& C, const PbTileEntry* __restrict__ e,
                                          const unsigned ve, unsigned q[16]) {
    if (PB_D_PLAIN(D.flags)) {
        const bool lean = (D.flags & PB_TILE_LEAN) != 0;
        const unsigned pitch = lean ? 16u * (unsigned)D.win_n16 : C.rowbytes;
        const unsigned off = lean ? (unsigned)D.win_a0 : (unsigned)D.anchor_r * C.rowbytes + 3u * (unsigned)D.anchor_c;
        const PbCoefs K = pb_coefs_of_lanes(ve);
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            pb_f2 a[5];
            pb_collapse_row_lanes(K, C.yb + 8 * jr, a);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const pb_f2 fv = pb_eval_row(a, C.u[k]);
                const unsigned dr = (unsigned)(int)fv.x, dc = (unsigned)(int)fv.y;  // >= 0: truncation == floor
                q[jr * 4 + k] = (lean ? __umul24(dr, pitch) : dr * pitch) + (__umul24(dc, 3u) + off);
            }
        }
    } else if (PB_D_GENERIC(D.flags)) {
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            PbRowModel R;
            pb_model_row(P, e, C.X0, C.Y0, C.yb + 8 * jr, 4 * C.xg, R);
#pragma unroll
            for (int k = 0; k < 4; ++k) q[jr * 4 + k] = (unsigned)pb_model_px_rc<SRC_KIND>(P, R, 4 * C.xg, k);
        }
    }
}

// DIRECT tiles: 16 unaligned dword gathers per lane straight from the frame (land in a[])
__device__ __forceinline__ void pb_d_direct_loads(const unsigned q[16], const uint8_t* __restrict__ s, unsigned a[16]) {
#pragma unroll
    for (int n = 0; n < 16; ++n) __builtin_memcpy(&a[n], s + q[n], 4);
}

// one eye's 16 pixels per lane -> a[] (low 3 bytes valid; DIRECT tiles already hold them)
__device__ __forceinline__ void pb_d_gather(const PbDesc& D, const PbTileCtx& C, int budget, const unsigned q[16], const unsigned* win,
                                            const uint8_t* __restrict__ s, unsigned a[16]) {
    if (D.flags & PB_TILE_LEAN) {
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const unsigned l = q[n];
            a[n] = __builtin_amdgcn_alignbyte(win[(l >> 2) + 1], win[l >> 2], l);
        }
    } else if (PB_D_GENERIC(D.flags)) {
        int nrows, n16;
        unsigned gbase;
        pb_d_generic_window(D, C, budget, nrows, n16, gbase);
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
    } else if (!(D.flags & PB_TILE_DIRECT)) {
#pragma unroll
        for (int n = 0; n < 16; ++n) a[n] = 0u;  // BLACK: this eye contributes nothing to the tile
    }
}

// the faithful taps and blend factors of one output pixel, stored once per plan
struct PbDoubleFix {
    int32_t il, ir;  // indices into the side-by-side frame, or -1 (an invalid destination pixel has both -1: black)
    double fl, fr;
};

__device__ __forceinline__ unsigned pb_double_fix_px(const PbDoubleFix& t, const uint8_t* __restrict__ s) {
    const unsigned l = pb_load_px(s, t.il), r = pb_load_px(s, t.ir);
    return pb_blend_u8(l & 0xFF, r & 0xFF, t.fl, t.fr) | (pb_blend_u8((l >> 8) & 0xFF, (r >> 8) & 0xFF, t.fl, t.fr) << 8) |
           (pb_blend_u8((l >> 16) & 0xFF, (r >> 16) & 0xFF, t.fl, t.fr) << 16);
}

// WMODE 0: every tile is UNIT; 1: a row table exists (unrotated panorama destination); 2: a latitude table exists.
// tile_fix: faithful taps of failed tiles (1024 per tile, slot = right-eye entry's aux_off); px_fix: of the fix list.
// ONE: single-frame launch (the frame loop and everything that keeps its invariants alive disappear).
// VEC (pb_remap_u8v, with ONE): chunk c of the grid is the frame (vtab.src[c], vtab.dst[c]).
template <int WMODE, bool ONE, bool VEC = false>
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_hot_double_kernel(const PbParams P, const PbTileEntry* __restrict__ table_l,
                                                                            const PbTileEntry* __restrict__ table_r,
                                                                            const PbTileEntry* __restrict__ ltable,
                                                                            const PbSepRow* __restrict__ rows,
                                                                            const double* __restrict__ lat_tab,
                                                                            const int32_t* __restrict__ fix_px,
                                                                            const PbDoubleFix* __restrict__ px_fix,
                                                                            const PbDoubleFix* __restrict__ tile_fix,
                                                                            const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                            int n_frames, unsigned long long src_stride,
                                                                            unsigned long long dst_stride, const unsigned groups_per_frame,
                                                                            const int frames_per_wave, const typename PbFrameTabOf<VEC>::type vtab) {
    // A batch is cut into chunks of frames_per_wave frames; chunks are a grid dimension (chunk-major, a tile group keeps
    // its XCD residue): the launch ramp and drain are paid once per batch, and inside a chunk the wave reuses its two
    // entries, blend weights and per-pixel addresses across the frames (ONE: chunks of one frame, no frame loop).
    // the five numbers a one-eye tile needs besides its entry, fetched with the table pointer in the wave's FIRST scalar round
    // trip (the compiler would otherwise fetch them after the entry has arrived: one more dependent trip per wave)
    const PbHot Hd = pb_hot_of(P);
    asm volatile("" ::"s"(ltable), "s"(Hd.dst_w), "s"(Hd.dst_h), "s"(Hd.src_w), "s"(Hd.src_h), "s"(Hd.win_budget), "s"(groups_per_frame));
    unsigned group = blockIdx.x;
    int frames = ONE ? 1 : frames_per_wave;
    if (VEC) {
        const unsigned chunk = group / groups_per_frame;
        group -= chunk * groups_per_frame;
        src = pb_frame_src(vtab, chunk, src);
        dst = pb_frame_dst(vtab, chunk, dst);
    } else if (group >= groups_per_frame) {
        const unsigned chunk = group / groups_per_frame;
        group -= chunk * groups_per_frame;
        src += (unsigned long long)chunk * (unsigned)frames_per_wave * src_stride;
        dst += (unsigned long long)chunk * (unsigned)frames_per_wave * dst_stride;
        if (!ONE) frames = min(frames_per_wave, n_frames - (int)chunk * frames_per_wave);
    } else if (!ONE) {
        frames = min(frames_per_wave, n_frames);
    }
    PbTileCtx C;
    C.lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = C.lane;
    // the wave's slot of the plan's launch-order table (pb_launch_table_kernel): which tile, and - for a tile that sees ONE
    // eye with weight exactly 1, most tiles of a stitch (28 260 of c5's 32 768) - that eye's entry, already in scalar
    // registers: a plain camera-source tile, done by the single-source tile code with half the vector instructions of the
    // two-eye path (PMC: 968 per wave against 405-445)
    const unsigned vslot = (unsigned)__builtin_amdgcn_readfirstlane((int)(group * (unsigned)PB_TILE_WAVES + (unsigned)wave));
    int tx, ty;
    {
        PbTileEntry entry;
        pb_load_entry(ltable + vslot, entry);
        if (entry.flags & PB_TILE_SKIP) return;
        tx = entry.tile_xy & 0xFFFF;
        ty = (int)((unsigned)entry.tile_xy >> 16);
        PB_TR(0);
#ifdef PB_TRACE
        if (lane == 0 && blockIdx.x / pb_trace_wpf == pb_trace_frame)
            pb_trace[(size_t)(ty * pb_tiles_x(P) + tx) * 16 + 14] = (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u) | ((unsigned long long)blockIdx.x << 8);  // HW_REG_XCC_ID, workgroup
#endif
#ifdef PB_ABLATION  // timing experiments only: 256 = skip the two-eye tiles, 512 = skip the one-eye (SOLO) tiles
        if ((P.exp_flags & 256) && !(entry.flags & PB_TILE_SOLO)) return;
        if ((P.exp_flags & 512) && (entry.flags & PB_TILE_SOLO)) return;
#endif
        if (entry.flags & PB_TILE_SOLO) {
            PB_TR(1);
            pb_win_tile<PB_KIND_CAMERA, false>(P, Hd, &entry, entry.flags & (PB_TILE_LEAN | PB_TILE_DIRECT | PB_TILE_BLACK), tx, ty, lane,
                                               pb_dyn_lds + (size_t)wave * ((Hd.win_budget >> 2) + 8), src, dst, frames, src_stride, dst_stride);
#ifdef PB_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PB_TR(6);
            PB_TR(7);
#endif
            return;
        }
    }
    const size_t tile = (size_t)ty * pb_tiles_x(P) + tx;
    const PbTileEntry* __restrict__ el = table_l + tile;
    const PbTileEntry* __restrict__ er = table_r + tile;
    // both entries: one coalesced 256-byte vector load each, in flight together
    const unsigned vl = reinterpret_cast<const unsigned*>(el)[C.lane], vr = reinterpret_cast<const unsigned*>(er)[C.lane];
    const PbDesc DL = pb_desc_of_lanes(vl), DR = pb_desc_of_lanes(vr);
    PB_TR(1);
    if ((DL.flags | DR.flags) & PB_TILE_FAILED) {
        // failed tile: every pixel through its stored faithful taps and factors (lane = 4 consecutive pixels x 4 rows)
        const PbDoubleFix* __restrict__ slot = tile_fix + (size_t)pb_lane_i(vr, PB_E_DWORD(aux_off)) * (PB_TILE * PB_TILE);
        const int xg = C.lane & 7, yb = C.lane >> 3, W = P.dst.width, H = P.dst.height;
        for (int f = 0; f < frames; ++f) {
            const uint8_t* s = src + (unsigned long long)f * src_stride;
            uint8_t* d = dst + (unsigned long long)f * dst_stride;
            for (int jr = 0; jr < 4; ++jr) {
                const int y = ty * PB_TILE + yb + 8 * jr, x = tx * PB_TILE + 4 * xg;
                if (y >= H) continue;
                unsigned a[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = pb_double_fix_px(slot[(yb + 8 * jr) * PB_TILE + 4 * xg + k], s);
                const unsigned long long off = 3ull * ((unsigned long long)y * W + x);
                if (x + 3 < W && (((uintptr_t)d + off) & 3u) == 0) {
                    pb_store3<false>(pb_pack_px4(a[0], a[1], a[2], a[3]), d + off);  // double sources: plain stores (pb_store3)
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
        return;
    }
    const bool by_row = WMODE == 1 && (DL.flags & PB_TILE_W_ROW) != 0;
    const bool by_lat = WMODE == 2 && (DL.flags & PB_TILE_W_LAT) != 0;
    C.rowbytes = 3u * (unsigned)P.src.width;
    C.frame_bytes = C.rowbytes * (unsigned)P.src.height;  // < 2^31 (host check)
    C.safe_len = C.frame_bytes & ~15u;
    C.xg = C.lane & 7;
    C.yb = C.lane >> 3;
    C.W = P.dst.width;
    C.H = P.dst.height;
    C.X0 = tx * PB_TILE;
    C.Y0 = ty * PB_TILE;
#pragma unroll
    for (int k = 0; k < 4; ++k) C.u[k] = pb_tile_coord(4 * C.xg + k);
    int budget_l, budget_r;
    pb_d_budgets(P, DL, DR, budget_l, budget_r);
    unsigned* win_l = pb_wave_window(P, wave, 8);
    unsigned* win_r = win_l + (budget_l >> 2);
    unsigned ql[16], qr[16], al[16], ar[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) ql[n] = qr[n] = al[n] = ar[n] = 0u;
    pb_d_issue(DL, C, budget_l, src, win_l);
    pb_d_issue(DR, C, budget_r, src, win_r);
    PB_TR(2);
    pb_d_math<PB_KIND_EYE_L>(P, DL, C, el, vl, ql);
    if (DL.flags & PB_TILE_DIRECT) pb_d_direct_loads(ql, src, al);
    pb_d_math<PB_KIND_EYE_R>(P, DR, C, er, vr, qr);
    if (DR.flags & PB_TILE_DIRECT) pb_d_direct_loads(qr, src, ar);
    PB_TR(3);
    // blend factors: per row group (UNIT / ROW), or per pixel from the stored latitudes (LAT; the loads fly with
    // the window loads, the factors are evaluated at the blend)
    double fl[4], fr[4];
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) fl[jr] = fr[jr] = 1.0;
    if (by_row) {
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            const PbSepRow R = rows[min(C.Y0 + C.yb + 8 * jr, C.H - 1)];
            fl[jr] = R.f_l;
            fr[jr] = R.f_r;
        }
    }
    double lat[WMODE == 2 ? 16 : 1];
    if (by_lat) {
        const double* __restrict__ lt = lat_tab + (size_t)pb_lane_i(vl, PB_E_DWORD(aux_off)) * PB_LAT_TILE_DOUBLES;
#pragma unroll
        for (int jr = 0; jr < 4; ++jr)
#pragma unroll
            for (int k = 0; k < 4; ++k) lat[WMODE == 2 ? jr * 4 + k : 0] = lt[(C.yb + 8 * jr) * PB_TILE + 4 * C.xg + k];
    }
    const int x = C.X0 + 4 * C.xg;
    const bool inside = C.X0 + PB_TILE <= C.W && C.Y0 + PB_TILE <= C.H;
    for (int f = 0; f < frames; ++f) {
        const uint8_t* s = src + (unsigned long long)f * src_stride;
        uint8_t* d = dst + (unsigned long long)f * dst_stride;
        // the addresses are loop-invariant; keep the compiler from hoisting everything derived from them out of
        // the frame loop (hundreds of live registers for nothing)
#pragma unroll
        for (int n = 0; n < 16 && !ONE; ++n) {
            asm volatile("" : "+v"(ql[n]));
            asm volatile("" : "+v"(qr[n]));
            if (WMODE == 2) asm volatile("" : "+v"(lat[WMODE == 2 ? n : 0]));
        }
        if (f > 0) {
            pb_d_issue(DL, C, budget_l, s, win_l);
            pb_d_issue(DR, C, budget_r, s, win_r);
            if (DL.flags & PB_TILE_DIRECT) pb_d_direct_loads(ql, s, al);
            if (DR.flags & PB_TILE_DIRECT) pb_d_direct_loads(qr, s, ar);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // both windows / the direct gathers have landed
        pb_wave_sync();
        PB_TR(4);
        pb_d_gather(DL, C, budget_l, ql, win_l, s, al);
        pb_d_gather(DR, C, budget_r, qr, win_r, s, ar);
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            unsigned a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double wl = fl[jr], wr = fr[jr];
                if (WMODE == 2 && by_lat) {
                    const double t = lat[WMODE == 2 ? jr * 4 + k : 0];
                    wl = pb_merge_factor(P, t);
                    wr = pb_merge_factor(P, (t * -1.0) + PB_PI);  // the right eye's latitude, projection.py:426-427
                }
                if (!by_row && !(WMODE == 2 && by_lat)) {  // UNIT tile: the integer sum, no float64 compare per pixel
                    const unsigned l = al[jr * 4 + k], r = ar[jr * 4 + k];
                    a[k] = (((l & 0x00FF00FFu) + (r & 0x00FF00FFu)) & 0x00FF00FFu) | (((l & 0x0000FF00u) + (r & 0x0000FF00u)) & 0x0000FF00u);
                } else {
                    a[k] = pb_sep_blend(al[jr * 4 + k], ar[jr * 4 + k], wl, wr);
                }
            }
            const int y = C.Y0 + C.yb + 8 * jr;
            if (!inside && y >= C.H) continue;
            const unsigned long long off = 3ull * ((unsigned long long)y * C.W + x);
            if ((inside || x + 3 < C.W) && (((uintptr_t)d + off) & 3u) == 0) {
                pb_store3<false>(pb_pack_px4(a[0], a[1], a[2], a[3]), d + off);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (inside || x + k < C.W) {
                        d[off + 3 * k + 0] = (uint8_t)(a[k] & 0xFF);
                        d[off + 3 * k + 1] = (uint8_t)((a[k] >> 8) & 0xFF);
                        d[off + 3 * k + 2] = (uint8_t)((a[k] >> 16) & 0xFF);
                    }
            }
        }
        PB_TR(5);
        if (f + 1 < frames) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // every lane has read its samples: the windows may be refilled
            pb_wave_sync();
        }
    }
#ifdef PB_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PB_TR(6);
#endif
    // the tile's fix pixels (either eye's list; a pixel listed by both is written twice): through their stored taps,
    // after the wave's own stores have completed
    const int nl = pb_lane_i(vl, PB_E_DWORD(fix_cnt)), nr = pb_lane_i(vr, PB_E_DWORD(fix_cnt));  // <= PB_TILE_FAIL_LIMIT each
    if (nl + nr > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int base = 0; base < nl + nr; base += 64) {
            const int n = base + C.lane;
            if (n < nl + nr) {
                const int item = n < nl ? pb_lane_i(vl, PB_E_DWORD(fix_off)) + n : pb_lane_i(vr, PB_E_DWORD(fix_off)) + (n - nl);
                const unsigned p = (unsigned)fix_px[item];
                const PbDoubleFix t = px_fix[item];
                for (int f = 0; f < frames; ++f) {
                    const unsigned v = pb_double_fix_px(t, src + (unsigned long long)f * src_stride);
                    uint8_t* o = dst + (unsigned long long)f * dst_stride + 3ull * p;
                    o[0] = (uint8_t)(v & 0xFF);
                    o[1] = (uint8_t)((v >> 8) & 0xFF);
                    o[2] = (uint8_t)((v >> 16) & 0xFF);
                }
            }
        }
    }
#ifdef PB_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PB_TR(7);
#endif
}

// Plan creation: the faithful taps and factors of the failed tiles' pixels (blocks [0, 4 * n_fail_tiles), 256 px
// each; slot = position in fail_tiles, recorded in the right-eye entry) and of the fix list (remaining blocks).
__global__ __launch_bounds__(PB_BLOCK) void pb_double_tables_kernel(const PbParams P, PbTileEntry* __restrict__ table_r,
                                                                    const int32_t* __restrict__ fail_tiles, int n_fail_tiles,
                                                                    const int32_t* __restrict__ fix_px, int n_fix_px,
                                                                    PbDoubleFix* __restrict__ tile_fix, PbDoubleFix* __restrict__ px_fix) {
    int i, j;
    PbDoubleFix* out;
    if ((int)blockIdx.x < 4 * n_fail_tiles) {
        const int s = blockIdx.x >> 2, t = fail_tiles[s];
        const int ty = t / pb_tiles_x(P), tx = t - ty * pb_tiles_x(P);
        const int local = (blockIdx.x & 3) * 256 + threadIdx.x;
        i = min(ty * PB_TILE + (local >> 5), P.dst.height - 1);  // pixels beyond the image are never stored
        j = min(tx * PB_TILE + (local & 31), P.dst.width - 1);
        out = tile_fix + (size_t)s * (PB_TILE * PB_TILE) + local;
        if (local == 0) table_r[t].aux_off = s;
    } else {
        const unsigned item = (blockIdx.x - 4u * n_fail_tiles) * PB_BLOCK + threadIdx.x;
        if (item >= (unsigned)n_fix_px) return;
        const unsigned p = (unsigned)fix_px[item];
        i = (int)(p / (unsigned)P.dst.width);
        j = (int)(p - (unsigned)i * (unsigned)P.dst.width);
        out = px_fix + item;
    }
    PbCoord c = pb_dst_coord(P, i, j);
    c = pb_rotate_all(P, c);
    const PbDoubleTap t = pb_src_double_taps(P, c);
    PbDoubleFix r;
    r.il = t.il;  // both -1 for an invalid destination pixel: black (final_image[invalid_map] = 0, projection.py:460)
    r.ir = t.ir;
    r.fl = t.fl;
    r.fr = t.fr;
    *out = r;
}

// Plan creation, after both eyes' tables are built and certified.  One wave per tile:
//  * a tile failed for one eye is failed for both (the hot kernel skips it, the fix kernel owns it);
//  * two LEAN windows that do not fit the wave's LDS together: the larger one becomes DIRECT;
//  * the weight class: the faithful factors of EVERY pixel of the tile are compared with 1.0 (UNIT) and, when
//    a row table is given, with the table's entries (ROW); a tile that is neither gets a slot of the latitude
//    table (LAT, filled by pb_double_lat_kernel) or, beyond lat_capacity slots, goes to the fail list.
// counters: [1] failed tiles (shared with pb_certify_kernel), [7] ROW tiles, [8] LAT tiles (= slots handed out).
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_double_pair_kernel(const PbParams P, PbTileEntry* __restrict__ table_l,
                                                                             PbTileEntry* __restrict__ table_r,
                                                                             const PbSepRow* __restrict__ rows,
                                                                             int32_t* __restrict__ fail_tiles,
                                                                             unsigned* __restrict__ counters, unsigned lat_capacity) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tx, ty;
    if (!pb_tile_of_wave(P, wave, tx, ty)) return;
    const int tile = ty * pb_tiles_x(P) + tx;
    PbTileEntry* el = table_l + tile;
    PbTileEntry* er = table_r + tile;
    const int fl0 = el->flags, fr0 = er->flags;
    if ((fl0 | fr0) & PB_TILE_FAILED) {  // already on the fail list (by the eye that failed)
        if (lane == 0) {
            el->flags = PB_TILE_FAILED;
            er->flags = PB_TILE_FAILED;
        }
        return;
    }
    const int X0 = tx * PB_TILE, Y0 = ty * PB_TILE;
    const int y = lane & 31, xh = (lane >> 5) * 16;
    const int i = Y0 + y;
    bool unit = true, row_ok = rows != nullptr;
    for (int k = 0; k < 16; ++k) {
        const int j = X0 + xh + k;
        if (i >= P.dst.height || j >= P.dst.width) continue;
        PbCoord c = pb_dst_coord(P, i, j);
        c = pb_rotate_all(P, c);
        // the blend factors are functions of the latitude alone (projection.py:439-457): no taps, no sine / cosine here (round 4: this
        // kernel used to evaluate both eyes' whole source stage per pixel for the sake of two numbers)
        const double t_fl = pb_merge_factor(P, c.lat), t_fr = pb_merge_factor(P, (c.lat * -1.0) + PB_PI);
        unit = unit && t_fl == 1.0 && t_fr == 1.0;
        if (row_ok) row_ok = rows[i].f_l == t_fl && rows[i].f_r == t_fr && !c.inv;
    }
    const bool all_unit = __builtin_amdgcn_ballot_w64(!unit) == 0;
    const bool all_row = rows != nullptr && __builtin_amdgcn_ballot_w64(!row_ok) == 0;
    if (lane != 0) return;
    int wclass = all_unit ? PB_TILE_W_UNIT : PB_TILE_W_ROW;
    if (!all_unit && !all_row) {
        const unsigned slot = atomicAdd(&counters[8], 1u);
        if (slot >= lat_capacity) {
            el->flags = PB_TILE_FAILED;
            er->flags = PB_TILE_FAILED;
            fail_tiles[atomicAdd(&counters[1], 1u)] = tile;
            return;
        }
        el->aux_off = (int)slot;
        wclass = PB_TILE_W_LAT;
    }
    int nl = fl0, nr = fr0;
    const int need_l = (nl & PB_TILE_LEAN) ? el->win_rows * 16 * el->win_n16 : 0;
    const int need_r = (nr & PB_TILE_LEAN) ? er->win_rows * 16 * er->win_n16 : 0;
    if (need_l + need_r > PB_WINLDS_BYTES) {
        if (need_l >= need_r) nl = (nl & ~PB_TILE_LEAN) | PB_TILE_DIRECT;
        else nr = (nr & ~PB_TILE_LEAN) | PB_TILE_DIRECT;
    }
    nl |= wclass;
    if (wclass == PB_TILE_W_ROW) atomicAdd(&counters[7], 1u);
    el->flags = nl;
    er->flags = nr;
}

// Fills the latitude table: the faithful (rotated) latitude of every pixel of every LAT tile, tile-major.
__global__ __launch_bounds__(64 * PB_TILE_WAVES) void pb_double_lat_kernel(const PbParams P, const PbTileEntry* __restrict__ table_l,
                                                                            double* __restrict__ lat_tab) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tx, ty;
    if (!pb_tile_of_wave(P, wave, tx, ty)) return;
    const PbTileEntry* el = table_l + ((size_t)ty * pb_tiles_x(P) + tx);
    if ((el->flags & PB_TILE_FAILED) || !(el->flags & PB_TILE_W_LAT)) return;
    double* __restrict__ lt = lat_tab + (size_t)el->aux_off * PB_LAT_TILE_DOUBLES;
    const int y = lane & 31, xh = (lane >> 5) * 16;
    for (int k = 0; k < 16; ++k) {
        const int i = min(ty * PB_TILE + y, P.dst.height - 1), j = min(tx * PB_TILE + xh + k, P.dst.width - 1);
        PbCoord c = pb_dst_coord(P, i, j);
        c = pb_rotate_all(P, c);
        lt[y * PB_TILE + xh + k] = c.lat;
    }
}

// Plan creation: applies an LDS budget to a double plan (see pb_budget_kernel): an eye's LEAN window larger than the
// budget goes direct; two LEAN windows that do not fit together: the larger one goes direct.
__global__ void pb_budget_double_kernel(PbTileEntry* __restrict__ table_l, PbTileEntry* __restrict__ table_r,
                                        const int32_t* __restrict__ saved_l, const int32_t* __restrict__ saved_r, unsigned n_tiles,
                                        int budget, unsigned* __restrict__ counters) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    int fl = saved_l[t], fr = saved_r[t];
    int need_l = (fl & PB_TILE_LEAN) ? table_l[t].win_rows * 16 * table_l[t].win_n16 : 0;
    int need_r = (fr & PB_TILE_LEAN) ? table_r[t].win_rows * 16 * table_r[t].win_n16 : 0;
    if (need_l > budget) { fl = (fl & ~PB_TILE_LEAN) | PB_TILE_DIRECT; need_l = 0; }
    if (need_r > budget) { fr = (fr & ~PB_TILE_LEAN) | PB_TILE_DIRECT; need_r = 0; }
    if (need_l + need_r > budget) {
        if (need_l >= need_r) fl = (fl & ~PB_TILE_LEAN) | PB_TILE_DIRECT;
        else fr = (fr & ~PB_TILE_LEAN) | PB_TILE_DIRECT;
    }
    table_l[t].flags = fl;
    table_r[t].flags = fr;
    atomicAdd(&counters[0], (unsigned)((fl & PB_TILE_LEAN) != 0) + (unsigned)((fr & PB_TILE_LEAN) != 0));
    atomicAdd(&counters[1], (unsigned)((fl & PB_TILE_DIRECT) != 0) + (unsigned)((fr & PB_TILE_DIRECT) != 0));
}
