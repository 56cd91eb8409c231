// EXPERIMENT, NOT THE PRODUCT (round 5; compiled only with -DPB_BIL_PIPE_EXPERIMENT, experiments/r5/build_var.sh): a pipelined launch of
// the opt-in bilinear mode.  Measured on MI355X (c3): 69 us against 44 - REJECTED.  The idea was that pb_bilinear_hot_kernel's waves all
// wait for their window together and then all compute together, so a wave that keeps the NEXT tile's LDS-DMA loads in flight while it
// computes (two LDS regions per wave, half as many waves per CU for the same LDS) would keep the vector units busy.  The SQ counters
// say otherwise (gpurun_out/r5_sq/c3_bilinear_pipe71_*): with two waves per SIMD instead of four the vector units are busy 51 % of the
// launch instead of 62 % - what idles them is not the load phase but the LDS-read and dependency stalls INSIDE the tile arithmetic,
// which only more resident waves hide - and the loop costs +30 % vector instructions per tile (the 64-SGPR entry is spilled to VGPR
// lanes around the next entry's geometry).  Included at the end of csrc/pb_kernels_bilinear.hpp.
#pragma once

// ---- the PIPELINED launch (round 5) -----------------------------------------------------------------------------------------
// pb_bilinear_hot_kernel's waves all start together, all wait for their entry and their window (~6 000 cycles in which the CU's
// vector units idle), then all compute together: a launch is two to four such rounds, the vector units busy 59-68 % of it
// (profiles/r05_*_bilinear_final_sq.txt) - and the tile arithmetic IS the kernel's floor.  Here a wave takes PB_BIL_PIPE tiles one
// after the other (its workgroup: that many consecutive groups of its XCD's share of the launch-order table) and the LDS-DMA loads of
// tile t + 1's window are in flight while tile t is computed: the slots of a wave alternate between two LDS regions of the
// workgroup's pool (pb_bilinear_pipe_pool_kernel: each region the larger of the slots that share it), half as many waves per CU hold
// the same LDS.  Per tile: [entry t, window geometry t + 1: one scalar round trip] -> issue the loads of t + 1 -> compute t from its
// region (landed: waited for at the end of the previous step) -> s_waitcnt vmcnt(0) (t + 1's window; long done) -> the tile's stores,
// which nothing waits for until the end of the next step.  Tiles that do not take a whole window (direct gathers, coordinate table,
// half windows, black) run as in pb_bilinear_hot_kernel, in their own region; their waits also cover the loads in flight.
#ifndef PB_BIL_PIPE
#define PB_BIL_PIPE 4
#endif
struct PbBilGeom {  // what issuing a tile's window loads needs of its entry (dwords 0-3, 54-55, 56-63)
    int anchor_r, anchor_c, flags, win_rows, win_r0, win_c0, win_cols, win_n16, win_a0, fix_off, fix_cnt, aux_off, tile_xy, bil_off;
};
typedef int pb_i32x4 __attribute__((ext_vector_type(4)));
typedef int pb_i32x2 __attribute__((ext_vector_type(2)));
typedef int pb_i32x8 __attribute__((ext_vector_type(8)));
// entry `e` whole and the geometry of entry `n`, in one scalar round trip
__device__ __forceinline__ void pb_load_entry_and_geom(const PbTileEntry* __restrict__ e, PbTileEntry& L, const PbTileEntry* __restrict__ n, PbBilGeom& G) {
    pb_i32x16 q0, q1, q2, q3;
    pb_i32x4 g0;
    pb_i32x2 g1;
    pb_i32x8 g2;
    asm volatile("s_load_dwordx16 %0, %7, 0x0\n\ts_load_dwordx16 %1, %7, 0x40\n\ts_load_dwordx16 %2, %7, 0x80\n\t"
                 "s_load_dwordx16 %3, %7, 0xc0\n\ts_load_dwordx4 %4, %8, 0x0\n\ts_load_dwordx2 %5, %8, 0xd8\n\t"
                 "s_load_dwordx8 %6, %8, 0xe0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3), "=&s"(g0), "=&s"(g1), "=&s"(g2)
                 : "s"(e), "s"(n)
                 : "memory");
    int* w = reinterpret_cast<int*>(&L);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        w[i] = q0[i];
        w[16 + i] = q1[i];
        w[32 + i] = q2[i];
        w[48 + i] = q3[i];
    }
    G.anchor_r = g0[0]; G.anchor_c = g0[1]; G.flags = g0[2]; G.win_rows = g0[3];
    G.win_r0 = g1[0]; G.win_c0 = g1[1];
    G.win_cols = g2[0]; G.win_n16 = g2[1]; G.win_a0 = g2[2]; G.fix_off = g2[3]; G.fix_cnt = g2[4]; G.aux_off = g2[5]; G.tile_xy = g2[6]; G.bil_off = g2[7];
}
static_assert(offsetof(PbTileEntry, win_r0) == 0xd8 && offsetof(PbTileEntry, win_cols) == 0xe0 && offsetof(PbTileEntry, bil_off) == 0xfc, "PbBilGeom follows PbTileEntry's layout");
// pb_issue_window_loads with the LDS-DMA instruction written out: the compiler does not know of these loads, so it does not make the
// LDS reads of the tile being computed wait for them (it orders every LDS read behind every pending LDS-DMA load it knows of, whatever
// the addresses).  The kernel waits itself: s_waitcnt vmcnt(0) + pb_wave_sync() before a region is read.  lds_addr: byte address in LDS.
__device__ __forceinline__ void pb_issue_window_loads_ahead(const uint8_t* __restrict__ s, unsigned lds_addr, int lane, unsigned gbase, unsigned rowbytes,
                                                            int nrows, int n16, unsigned safe_len) {
    const unsigned pitch = 16u * (unsigned)n16;
    const unsigned inv = (65536u + n16 - 1) / n16;  // lane / n16 for lane < 64
    const unsigned lrow = ((unsigned)lane * inv) >> 16, chunk = (unsigned)lane - lrow * n16;
    const unsigned rpp = 64u / n16;
    const bool lane_on = lrow < rpp;
    for (unsigned rowb = 0; rowb < (unsigned)nrows; rowb += rpp) {
        const unsigned row = rowb + lrow;
        const unsigned ga = ((gbase + row * rowbytes) & ~15u) + 16u * chunk;
        const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_addr + rowb * pitch));
        if (lane_on && row < (unsigned)nrows && ga + 16u <= safe_len) {
            unsigned keep;  // (M0 - the instruction's LDS address - is the compiler's: handed back as found)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "s"(la), "v"(ga), "s"(s)
                         : "memory");
        }
    }
}
// a tile whose whole window can be in flight ahead of its turn
__device__ __forceinline__ bool pb_bil_prefetchable(int flags, int bil_off, int windows) {
    return windows && bil_off < 0 && (flags & PB_TILE_LEAN) && !(flags & (PB_TILE_SKIP | PB_TILE_BLACK)) && (PB_BIL_PATHS & 1) && !(PB_BIL_ABL & (4 | 64));
}
template <int SRC_KIND>
__global__ __launch_bounds__(64 * PB_TILE_WAVES, 2) void pb_bilinear_pipe_kernel(const PbHot Hd, const PbTileEntry* __restrict__ table,
                                                                             const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                             const unsigned groups_per_frame, const unsigned wgs_per_frame,
                                                                             unsigned long long src_stride, unsigned long long dst_stride, int windows,
                                                                             const PbBilCoord* __restrict__ bil_xy, const int32_t* __restrict__ fix_px,
                                                                             const PbBilCoord* __restrict__ fix_xy) {
    asm volatile("" ::"s"(table), "s"(Hd.dst_w), "s"(Hd.dst_h), "s"(Hd.src_w), "s"(Hd.src_h), "s"(groups_per_frame));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned wg = blockIdx.x;
    if (wg >= wgs_per_frame) {  // a batch: which frame
        const unsigned f = wg / wgs_per_frame;
        wg -= f * wgs_per_frame;
        src += (unsigned long long)f * src_stride;
        dst += (unsigned long long)f * dst_stride;
    }
    // the workgroup's groups: PB_BIL_PIPE consecutive groups of ITS XCD's stream (workgroup i runs on XCD i % 8: group g of the
    // launch-order table was laid out for XCD g % 8)
    const unsigned g0 = (wg & 7u) + 8u * ((wg >> 3) * (unsigned)PB_BIL_PIPE);
    const unsigned rowbytes = 3u * (unsigned)Hd.src_w, safe_len = (rowbytes * (unsigned)Hd.src_h) & ~15u;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)pb_dyn_lds;  // (LDS addresses are 32-bit offsets)
    int state = 0;  // of the CURRENT tile's window (pb_bil_model_vals)
    for (int t = 0; t < PB_BIL_PIPE; ++t) {
        const unsigned g = g0 + 8u * (unsigned)t;
        if (g >= groups_per_frame) break;
        const bool has_next = t + 1 < PB_BIL_PIPE && g + 8u < groups_per_frame;
        const unsigned vslot = (unsigned)__builtin_amdgcn_readfirstlane((int)(g * (unsigned)PB_TILE_WAVES + (unsigned)wave));
        PbTileEntry entry;
        PbBilGeom nx;
        pb_load_entry_and_geom(table + vslot, entry, table + (has_next ? vslot + 8u * (unsigned)PB_TILE_WAVES : vslot), nx);
        const PbTileEntry* __restrict__ e = &entry;
        const int flags = e->flags;
        if (t == 0 && pb_bil_prefetchable(flags, e->bil_off, windows)) {  // the first tile's own window
            pb_issue_window_loads_ahead(src, lds0 + (unsigned)e->win_r0, lane, (unsigned)e->anchor_r * rowbytes + 3u * (unsigned)e->anchor_c, rowbytes, e->win_rows,
                                        e->win_n16, safe_len);
            state = 1;
        }
        const bool next_ahead = has_next && pb_bil_prefetchable(nx.flags, nx.bil_off, windows);
        if (next_ahead)
            pb_issue_window_loads_ahead(src, lds0 + (unsigned)nx.win_r0, lane, (unsigned)nx.anchor_r * rowbytes + 3u * (unsigned)nx.anchor_c, rowbytes, nx.win_rows,
                                        nx.win_n16, safe_len);
        const bool live = !(flags & PB_TILE_SKIP) && !(e->bil_off >= 0 && !bil_xy);  // (no coordinate table: the float64 pass owns the tile)
        unsigned v[16];
        if (live) pb_bil_vals<SRC_KIND == PB_KIND_PANO>(Hd, e, flags, lane, pb_dyn_lds + ((unsigned)e->win_r0 >> 2), windows, src, bil_xy, 0, Hd.src_w, v, state);
        // the next tile's window has had this tile's arithmetic to arrive (and the previous tile's stores to complete)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        state = next_ahead ? 2 : 0;
        if (!live) continue;
        const int tx = e->tile_xy & 0xFFFF, ty = (int)((unsigned)e->tile_xy >> 16);
#ifdef PB_BIL_PLAIN_STORES  // A/B builds only
        pb_bil_store<false>(v, dst, tx * PB_TILE, ty * PB_TILE, lane, Hd.dst_w, Hd.dst_h);
#else
        pb_bil_store<SRC_KIND == PB_KIND_CAMERA>(v, dst, tx * PB_TILE, ty * PB_TILE, lane, Hd.dst_w, Hd.dst_h);
#endif
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
}


// The pool of a PIPELINED workgroup (pb_bilinear_pipe_kernel; single sources): wave w takes the slots (t, w) of the workgroup's
// PB_BIL_PIPE groups one after the other, the window of slot t + 1 in flight while slot t is computed - so its slots alternate between
// TWO regions of its own, each as large as the largest slot that uses it, and no wave ever touches another wave's region.  One thread per
// workgroup; demotion and counters as in pb_bilinear_pool_kernel.
__global__ void pb_bilinear_pipe_pool_kernel(PbTileEntry* __restrict__ ltable, unsigned n_groups, unsigned n_wgs, unsigned pool_bytes, int dry,
                                             unsigned* __restrict__ counters) {
    const unsigned W = blockIdx.x * blockDim.x + threadIdx.x;
    if (W >= n_wgs) return;
    const unsigned g0 = (W & 7u) + 8u * ((W >> 3) * (unsigned)PB_BIL_PIPE);
    unsigned need[PB_BIL_PIPE][4];
    bool lean[PB_BIL_PIPE][4];
    for (int t = 0; t < PB_BIL_PIPE; ++t)
        for (int w = 0; w < 4; ++w) {
            const unsigned g = g0 + 8u * (unsigned)t;
            need[t][w] = 0u;
            lean[t][w] = false;
            if (g >= n_groups) continue;
            const PbTileEntry& e = ltable[4u * g + w];
            const int f = e.flags;
            if (f & PB_TILE_SKIP) continue;
            need[t][w] = pb_bil_region_bytes(e, f);
            lean[t][w] = e.bil_off < 0 && (f & (PB_TILE_LEAN | PB_TILE_HALVES)) != 0;
        }
    unsigned demoted = 0;
    unsigned R[4][2];
    for (;;) {
        unsigned total = 0;
        for (int w = 0; w < 4; ++w)
            for (int p = 0; p < 2; ++p) {
                unsigned m = 0;
                for (int t = p; t < PB_BIL_PIPE; t += 2) m = need[t][w] > m ? need[t][w] : m;
                R[w][p] = m;
                total += m;
            }
        if (total <= pool_bytes) break;
        int bt = -1, bw = -1;
        for (int t = 0; t < PB_BIL_PIPE; ++t)
            for (int w = 0; w < 4; ++w)
                if (lean[t][w] && need[t][w] > (unsigned)PB_DIRECT_LDS_BYTES + 16u && (bt < 0 || need[t][w] > need[bt][bw])) { bt = t; bw = w; }
        if (bt < 0) {
            atomicAdd(&counters[1], 1u);
            return;
        }
        need[bt][bw] = (unsigned)PB_DIRECT_LDS_BYTES + 16u;
        lean[bt][bw] = false;
        if (!dry) {
            PbTileEntry& e = ltable[4u * (g0 + 8u * (unsigned)bt) + bw];
            e.flags = (e.flags & ~(PB_TILE_LEAN | PB_TILE_HALVES)) | PB_TILE_DIRECT;
        }
        ++demoted;
    }
    if (demoted) atomicAdd(&counters[0], demoted);
    if (dry) return;
    unsigned off = 0, base[4][2];
    for (int w = 0; w < 4; ++w)
        for (int p = 0; p < 2; ++p) {
            base[w][p] = off;
            off += R[w][p];
        }
    for (int t = 0; t < PB_BIL_PIPE; ++t)
        for (int w = 0; w < 4; ++w) {
            const unsigned g = g0 + 8u * (unsigned)t;
            if (g < n_groups) ltable[4u * g + w].win_r0 = (int)base[w][t & 1];
        }
}

