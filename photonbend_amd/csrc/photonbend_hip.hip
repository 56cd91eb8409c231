// photonbend_hip.hip - the C ABI declared in include/photonbend_hip.h over the HIP kernels
// (gfx950 / CDNA4):
//   pb_kernels_faithful.hpp  per-pixel float64 chain (the reference semantics on the device; map API; PB_MODE_FAITHFUL)
//   pb_kernels_tile.hpp      hot kernel for pano / camera sources (per-tile float32 models, LDS windows, exact
//                            lookup tables), plan builders
//   pb_kernels_double.hpp    hot kernel for double-fisheye sources (one tile table per eye, weight classes)
//   pb_kernels_sep.hpp       separable tables for the unrotated stitch (row factors; unaligned-frame fallback)
//   pb_kernels_bilinear.hpp  opt-in bilinear sampling
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared ...
// (-ffp-contract=off is REQUIRED: the reference rounds every multiply and add
// separately; fused multiply-adds appear only where written as fma()/fmaf()).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <algorithm>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "pb_params.hpp"
#include "pb_stages.hpp"
#include "pb_kernels_faithful.hpp"
#include "pb_tile.hpp"
#include "pb_kernels_tile.hpp"
#include "pb_kernels_sep.hpp"
#include "pb_kernels_double.hpp"
#include "pb_kernels_bilinear.hpp"

#define PB_DOUBLE_FRAMES_PER_WAVE 1  // frames a double-source wave loops over (the rest of a batch is a grid dimension)
#define PB_WAVES_PER_WG 4  // waves per workgroup of the hot kernel (LDS is released per workgroup)
struct pb_plan {
    PbParams P;
    int mode = PB_MODE_AUTO;     // PB_MODE_AUTO / PB_MODE_FAITHFUL / PB_MODE_FAST
    int device = -1;             // device that owns the tables below
    int fast_ready = 0;          // models + fix list built and certified on `device`
    PbTileEntry* table = nullptr;
    int32_t* fail_tiles = nullptr;
    int32_t* fix_px = nullptr;
    int32_t* idx_tab = nullptr;  // exact source indices of the pixels of failed tiles (4 KiB per tile) ...
    int32_t* fix_idx = nullptr;  // ... and of the fix list's pixels: looked up per frame instead of recomputed
    unsigned n_tiles = 0, n_fail_tiles = 0, n_fix_px = 0, n_lean_tiles = 0, n_black_tiles = 0, n_direct_tiles = 0;
    long long diff_pixels = -1;  // pixels (outside failed tiles) where model and faithful index differed
    // separable path (double source, unrotated pano destination): row / column tables.  sep_ready: the tables exist (their row weights
    // are exact by construction and serve the tile kernels); sep_checked: the exhaustive check of their TAPS against the faithful ones -
    // 0 not started, 2 in flight, 1 passed, -1 failed.  The check (one float64 chain per pixel: 0.27 ms of c5's 1.3 ms preparation) is only
    // needed by pb_sep_double_kernel, the fallback for frames the windowed two-eye kernel cannot take; the first launch that wants the
    // fallback ENQUEUES it (pb_sep_usable: no allocation, no synchronisation - the launch functions' contract) and takes the float64 kernel,
    // later launches take the fallback once the check has been seen to pass.
    int sep_ready = 0;
    mutable int sep_checked = 0;
    int sep_slot = -1;               // this plan's word of the process's pinned result page (pb_sep_setup)
    hipEvent_t sep_event = nullptr;  // recorded behind the check
    PbSepRow* sep_rows = nullptr;
    PbSepCol* sep_cols = nullptr;
    // double-fisheye source: one certified tile table per eye (pb_kernels_double.hpp); `table` is the left eye's
    int dbl_ready = 0;
    PbTileEntry* table_r = nullptr;
    PbDoubleFix* dbl_tile_fix = nullptr;  // faithful taps + factors of failed tiles' pixels / of the fix list's pixels
    PbDoubleFix* dbl_px_fix = nullptr;
    double* lat_tab = nullptr;   // faithful latitudes of the pixels of merge-band tiles (PB_TILE_W_LAT), 8 KiB per tile
    unsigned n_row_weight_tiles = 0, n_lat_tiles = 0;
    // the tile flags as classified and certified with the largest window budget: pb_apply_budget derives the flags
    // in use from them, so the budget can be changed at any time without touching a pixel
    int32_t* saved_l = nullptr;
    int32_t* saved_r = nullptr;
    // the hot kernel's launch-order copy of `table` (rebuilt with every budget change; derived, never serialized)
    PbTileEntry* ltable = nullptr;
    int32_t* bil_tiles = nullptr;  // tiles the bilinear tile kernels leave to the float64 pass: PB_TILE_COARSE models; double-fisheye plans also tiles an eye sees but that are not plain for it
    unsigned n_bil_tiles = 0;
    // the bilinear mode's exact coordinate tables (pb_kernels_bilinear.hpp): 8 KiB per slot (PbTileEntry::bil_off), and the fix list's
    // coordinates (two per pixel for a double-fisheye source); nullptr = the tables would not fit, those tiles take the float64 pass
    PbBilCoord* bil_xy = nullptr;
    PbBilCoord* bil_fix_xy = nullptr;
    unsigned n_bil_slots = 0;
    // ... and its own launch-order table, classified under ITS window budget (PB_BIL_WIN_BUDGET: a direct-gather tile costs the
    // bilinear mode two 8-byte loads per pixel, the nearest mode one dword - its best budget is larger) and ordered by its own costs;
    // built once per plan (pb_build_bilinear_launch), untouched by pb_plan_set_window_budget
    PbTileEntry* ltable_bil = nullptr;
    unsigned launch_groups_bil = 0;
    int bil_budget = 0;
    PbDblTables* bil_dbl_tables = nullptr;  // double-fisheye plans: what only some waves of the bilinear launch need, behind one pointer (pb_kernels_bilinear.hpp)
    int bil_wanted = 1;           // 0: created with PB_PLAN_NO_BILINEAR - the opt-in mode's tables are built when pb_plan_prepare(PB_PLAN_BILINEAR) asks for them
    int bil_waves = 4;            // waves per REAL workgroup of a bilinear launch (4 or 2: chosen with the pool, pb_build_bilinear_launch)
    unsigned bil_pool_bytes = 0;  // dynamic LDS of a bilinear launch's workgroup: the slots' regions are packed into it (pb_bilinear_pool_kernel)
    PbParams* P_dev = nullptr;   // device copy of P as the hot launches see it (refreshed with every budget change)
    unsigned launch_groups = 0;  // virtual workgroups (of four waves) per frame, a multiple of 8
    int walk = 0;                // launch-order rule: 0 = by policy (pb_build_launch_table), 1 = plain, 2 = rows from the heaviest outwards,
                                 // 3 = super-tiles heaviest first; PB_PLAN_TUNE may pick 1-3 by timing, a serialized plan remembers it
    double prepare_ms = 0.0, tune_ms = 0.0;  // host wall time of the device preparation / of the optional budget tuning
};

// Experiment knobs (environment variables) exist in the -DPB_ABLATION diagnostic build only (experiments/); the product
// library reads no environment.
#ifdef PB_ABLATION
static int pb_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
static inline int pb_knob(const char*, int dflt) { return dflt; }
#endif

// -DPB_ABLATION only: PB_PLAN_STAGES=1 prints where a plan preparation's host time goes (experiments/r6/plan_stages.py): PB_STAGE("name")
// stamps the host clock when the code reaches it, pb_stage_report prints the intervals.  Nothing in the product build.
#ifdef PB_ABLATION
#include <vector>
static std::vector<std::pair<const char*, double>> g_stages;
static void pb_stage_mark(const char* name) {
    static const int on = pb_knob("PB_PLAN_STAGES", 0);
    if (!on) return;
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    g_stages.emplace_back(name, ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6);
}
static void pb_stage_report() {
    if (g_stages.size() > 1) {
        fprintf(stderr, "[plan stages]");
        for (size_t k = 1; k < g_stages.size(); ++k) fprintf(stderr, " %s %.0f", g_stages[k].first, 1e3 * (g_stages[k].second - g_stages[k - 1].second));
        fprintf(stderr, " | total %.0f us\n", 1e3 * (g_stages.back().second - g_stages.front().second));
    }
    g_stages.clear();
}
#define PB_STAGE(name) pb_stage_mark(name)
#else
#define PB_STAGE(name) ((void)0)
static inline void pb_stage_report() {}
#endif

// -DPB_ABLATION only: PB_FAIL_LTABLE_ALLOC=n makes the n-th launch-table allocation of the process fail (tests of the error path)
static bool pb_test_alloc_fails() {
#ifdef PB_ABLATION
    static int countdown = pb_knob("PB_FAIL_LTABLE_ALLOC", 0);
    if (countdown > 0 && --countdown == 0) return true;
#endif
    return false;
}

// -DPB_ABLATION only: PB_BIL_OFF=<bits> switches SPEED-ONLY features of the opt-in bilinear mode off at plan creation (tests: the pixels
// must not depend on them) - 1 half windows, 2 unguarded table tiles, 4 the table tiles' walk, 8 half windows in pair slots, 16 the small LDS pool
static inline bool pb_bil_off(int bit) { return (pb_knob("PB_BIL_OFF", 0) & bit) != 0; }

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

// Small device buffers that live for a few launches of plan preparation (counters, per-unit costs, a column table): a hipFree costs ~35 us
// on this stack (it synchronises the device) and a plan used to make nine - a third of a millisecond of a 1.3 ms preparation
// (experiments/r4/plan_api.sh).  They come from a per-device cache instead: a freed block is kept (up to 96 MiB in blocks of up to 16 MiB: the tables of a 33-Mpx plan)
// and handed to the next request it fits.  Safe because every user works on the NULL stream, which orders a block's next kernel behind its
// last one device-wide, and because each site releases its block only after a synchronising copy or hipDeviceSynchronize.
// Round 6: a plan's OWN tables (tile tables, fix lists, launch tables, parameter block) come from the same cache - a warm c2 plan made
// eight hipMallocs (~40 us of its 0.45 ms) and its destruction up to ten hipFrees (~350 us).  Launches read those tables on the caller's
// streams, which the NULL stream does not order: pb_plan_destroy and the budget change wait for the device ONCE before they hand blocks
// back (what every one of their hipFrees did before).
namespace {
struct PbTmpCache {
    struct Block { int device; size_t bytes; void* ptr; };
    std::mutex lock;
    std::vector<Block> idle;
    std::unordered_map<void*, std::pair<int, size_t>> live;
    size_t held = 0;
} g_tmp;
}  // namespace
static hipError_t pb_tmp_alloc(void** out, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> g(g_tmp.lock);
        size_t best = g_tmp.idle.size();
        for (size_t k = 0; k < g_tmp.idle.size(); ++k) {
            const PbTmpCache::Block& b = g_tmp.idle[k];
            if (b.device == device && b.bytes >= bytes && b.bytes <= 4 * bytes + 4096 && (best == g_tmp.idle.size() || b.bytes < g_tmp.idle[best].bytes)) best = k;
        }
        if (best != g_tmp.idle.size()) {
            const PbTmpCache::Block b = g_tmp.idle[best];
            g_tmp.idle.erase(g_tmp.idle.begin() + (long)best);
            g_tmp.held -= b.bytes;
            g_tmp.live[b.ptr] = {b.device, b.bytes};
            *out = b.ptr;
            return hipSuccess;
        }
    }
    e = hipMalloc(out, bytes);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> g(g_tmp.lock);
        g_tmp.live[*out] = {device, bytes};
    }
    return e;
}
static void pb_tmp_free(void* ptr) {
    if (!ptr) return;
    {
        std::lock_guard<std::mutex> g(g_tmp.lock);
        const auto it = g_tmp.live.find(ptr);
        if (it != g_tmp.live.end()) {
            const std::pair<int, size_t> info = it->second;
            g_tmp.live.erase(it);
            if (info.second <= ((size_t)16 << 20) && g_tmp.held + info.second <= ((size_t)96 << 20)) {
                g_tmp.idle.push_back({info.first, info.second, ptr});
                g_tmp.held += info.second;
                return;
            }
        }
    }
    (void)hipFree(ptr);
}
// Waits for `device` (< 0: the current one): a plan's launches may still be reading its tables on the caller's streams when a block goes
// back to the cache - what the hipFree of rounds 1-5 did by itself, once per table.
static void pb_wait_device(int device) {
    int cur = -1;
    if (device < 0 || hipGetDevice(&cur) != hipSuccess || cur == device) {
        (void)hipDeviceSynchronize();
        return;
    }
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    (void)hipSetDevice(cur);
}
// one table of a plan back to the cache (nullptr: nothing, and no wait)
static void pb_table_release(int device, void* ptr) {
    if (!ptr) return;
    pb_wait_device(device);
    pb_tmp_free(ptr);
}

static inline unsigned pb_blocks(unsigned long long items);
// The separable tables' taps against the faithful chain, every pixel, once per plan: enqueued by the first launch that wants
// pb_sep_double_kernel, its verdict read - without waiting - by the launches after it.  The verdict lands in a word of a page of pinned,
// device-mapped host memory the process allocates once (the check kernel counts mismatches straight into it).
namespace {
struct PbSepPage {
    std::mutex lock;
    unsigned* host = nullptr;  // 1024 words
    unsigned* dev = nullptr;
    std::vector<int> idle;
} g_sep;
}  // namespace
// plan preparation / deserialisation: a result word and an event for the plan (failure = the fallback stays unverified, i.e. unused)
static void pb_sep_setup(pb_plan* pl) {
    if (!pl->sep_ready || pl->sep_slot >= 0) return;
    std::lock_guard<std::mutex> g(g_sep.lock);
    if (!g_sep.host) {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 1024 * sizeof(unsigned), hipHostMallocMapped) != hipSuccess) return;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            (void)hipHostFree(h);
            return;
        }
        g_sep.host = (unsigned*)h;
        g_sep.dev = (unsigned*)d;
        for (int k = 1023; k >= 0; --k) g_sep.idle.push_back(k);
    }
    if (g_sep.idle.empty()) return;
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return;
    pl->sep_slot = g_sep.idle.back();
    g_sep.idle.pop_back();
    pl->sep_event = ev;
    pl->sep_checked = 0;
}
static void pb_sep_release(pb_plan* pl) {
    std::lock_guard<std::mutex> g(g_sep.lock);
    if (pl->sep_event) {
        (void)hipEventSynchronize(pl->sep_event);  // (a check still in flight writes its word: the slot must not be handed on before)
        (void)hipEventDestroy(pl->sep_event);
    }
    if (pl->sep_slot >= 0) g_sep.idle.push_back(pl->sep_slot);
    pl->sep_event = nullptr;
    pl->sep_slot = -1;
}
// launch path: may pb_sep_double_kernel run?  Never waits, never allocates.
static bool pb_sep_usable(const pb_plan* plan, hipStream_t st) {
    if (!plan->sep_ready || plan->sep_slot < 0 || !plan->sep_event || !plan->sep_rows || !plan->sep_cols) return false;
    std::lock_guard<std::mutex> g(g_sep.lock);
    if (plan->sep_checked == 1) return true;
    if (plan->sep_checked == -1) return false;
    if (plan->sep_checked == 2) {
        if (hipEventQuery(plan->sep_event) != hipSuccess) return false;  // still running (or the query failed): not yet
        plan->sep_checked = g_sep.host[plan->sep_slot] == 0u ? 1 : -1;
        return plan->sep_checked == 1;
    }
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;  // (an event recorded inside a capture could not be queried afterwards)
    if (st && hipStreamIsCapturing(st, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) return false;
    const PbParams& P = plan->P;
    g_sep.host[plan->sep_slot] = 0u;
    hipLaunchKernelGGL(pb_sep_check_kernel, dim3(pb_blocks((unsigned long long)P.dst.height * P.dst.width)), dim3(PB_BLOCK), 0, st, P, plan->sep_rows,
                       plan->sep_cols, g_sep.dev + plan->sep_slot);
    if (hipEventRecord(plan->sep_event, st) == hipSuccess) plan->sep_checked = 2;
    return false;
}

// ----------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------
enum { PB_ROLE_DST = 1, PB_ROLE_SRC = 2, PB_ROLE_CUSTOM_OK = 4 };
static bool pb_end_ok(const pb_proj* p, std::string& why, int role = PB_ROLE_DST | PB_ROLE_SRC) {
    if (!p) {
        why = "null pb_proj";
        return false;
    }
    if (p->kind < PB_KIND_CAMERA || p->kind > PB_KIND_PANO) {
        why = "pb_proj.kind out of range";
        return false;
    }
    if (p->kind != PB_KIND_PANO && (p->lens < PB_LENS_EQUIDISTANT || p->lens > PB_LENS_THOBY) &&
        !((role & PB_ROLE_CUSTOM_OK) && p->lens == PB_LENS_CUSTOM)) {
        why = p->lens == PB_LENS_CUSTOM ? "PB_LENS_CUSTOM is valid only where the host supplies the lens values (pb_index_from_map_i32 with distance planes)"
                                        : "pb_proj.lens out of range";
        return false;
    }
    if (p->height < 1 || p->width < 1 || (long long)p->height * p->width > 0x7FFFFFFFll / 4) {
        why = "pb_proj height/width out of range (need 1 <= h*w < 2^29)";
        return false;
    }
    if (p->kind == PB_KIND_DOUBLE && (p->width & 1) && (role & PB_ROLE_DST)) {
        // the reference's map of an odd-width double DESTINATION is 2 * (W // 2) wide (projection.py:389-397): pass that width;
        // an odd-width double SOURCE is fine (eyes of W // 2 and W - W // 2 columns, projection.py:429-431)
        why = "a double-fisheye destination needs an even width (the reference's map is 2 * (W // 2) wide)";
        return false;
    }
    if (p->kind == PB_KIND_DOUBLE && p->width < 2) {
        why = "a double-fisheye frame needs at least two columns";
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


static inline unsigned pb_hot_blocks(const PbParams& P) {
    const unsigned tx = (P.dst.width + PB_TILE - 1) / PB_TILE, ty = (P.dst.height + PB_TILE - 1) / PB_TILE;
    return ((tx + 1) / 2) * ((ty + 1) / 2);
}
static inline unsigned pb_num_tiles(const PbParams& P) {
    return ((P.dst.width + PB_TILE - 1) / PB_TILE) * ((P.dst.height + PB_TILE - 1) / PB_TILE);
}

static_assert(16384 / PB_TILE < 65536, "PbTileEntry::tile_xy packs tile coordinates in 16 bits each");
static bool pb_fast_possible_dims(const PbParams& P) {
    // the tile models need 32-bit squares of the doubled pixel offsets and u24 index arithmetic
    return P.dst.width <= 16384 && P.dst.height <= 16384 && P.src.width < 32768 && P.src.height < 32768;
}
static bool pb_fast_possible(const PbParams& P) {
    return P.src.kind != PB_KIND_DOUBLE && P.dst.width <= 16384 && P.dst.height <= 16384 && P.src.width < (1 << 24) &&
           P.src.height < (1 << 24);
}

// the bilinear mode's plan state (pb_kernels_bilinear.hpp): which tiles the models cannot serve, their slots in the exact coordinate
// table, the table itself and the fix list's coordinates - all from the faithful float64 chain, once per plan; derived state, not
// serialized (rebuilt from the tile tables).  Synchronous.
// A double-fisheye source whose field of view is within one degree of 180 (not 180 itself): the reference keeps blending for half a degree past the merge
// band's end with the band's own slope (projection.py:416-418, :440-444), so the factor there reaches -0.5 deg / (fov - 180 deg) - minus
// 45 at 180.011 degrees - and multiplies whatever an eye's sample is off by.  The tile kernels' eye samples may be the neighbouring integer
// (1 LSB); with a band of a degree or more the factor stays within [-0.5, 1] and the sum within 2 LSB; below that the mode
// runs its per-pixel float64 kernels, whose eye samples are the definition's.  (Exactly 180 degrees: the factor is infinite or NaN and
// the cast gives 0 whatever the sample - no restriction.)
static bool pb_bilinear_tiles_allowed(const PbParams& P) {
    // (below 180 degrees the band turns inside out; it exists down to 179.5 degrees, with factors from 1 up to 0.5 deg / (180 deg - fov))
    return !(P.src.kind == PB_KIND_DOUBLE && P.mrg_range != 0.0 && fabs(P.mrg_range) < 0.999 * (PB_PI / 180.0));  // (181 / 179 degrees themselves: tiles)
}
static int pb_build_bilinear_list(pb_plan* pl) {
    const PbParams& P = pl->P;
    unsigned* cnt = nullptr;
    pb_table_release(pl->device, pl->bil_tiles);
    pb_table_release(pl->device, pl->bil_xy);
    pb_table_release(pl->device, pl->bil_fix_xy);
    pl->bil_tiles = nullptr;
    pl->bil_xy = pl->bil_fix_xy = nullptr;
    pl->n_bil_tiles = pl->n_bil_slots = 0;
    if (!pb_bilinear_tiles_allowed(P)) return PB_OK;  // (no tables: pb_remap_bilinear_u8 takes the float64 kernels)
    PB_HIP(pb_tmp_alloc((void**)&cnt, 2 * sizeof(unsigned)));
    hipError_t e = hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned), 0);
    if (e == hipSuccess) e = pb_tmp_alloc((void**)&pl->bil_tiles, (size_t)(pl->n_tiles ? pl->n_tiles : 1) * sizeof(int32_t));
    unsigned res[2] = {0u, 0u};
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pb_bilinear_tile_list_kernel, dim3((pl->n_tiles + 255) / 256), dim3(256), 0, 0, pl->table, pl->table_r, pl->n_tiles, P.src_eye_w,
                           P.src.width, pl->bil_tiles, cnt);
        e = hipMemcpy(res, cnt, sizeof(res), hipMemcpyDeviceToHost);
    } else {
        (void)hipDeviceSynchronize();
    }
    pb_tmp_free(cnt);
    const bool dbl = pl->table_r != nullptr;
    // the tables: sources the 1/4096-px fixed point holds, and at most 1 GiB of coordinates (a geometry the models mostly cannot
    // follow keeps the float64 pass)
    const bool tables = e == hipSuccess && P.src.width < PB_BIL_MAX_DIM && P.src.height < PB_BIL_MAX_DIM && (size_t)P.src.width * P.src.height >= 3 &&
                        (size_t)res[1] * PB_TILE * PB_TILE * sizeof(PbBilCoord) <= ((size_t)1 << 30);
    if (tables) {
        const unsigned np = pl->n_fix_px, stride = dbl ? 2u : 1u;
        e = pb_tmp_alloc((void**)&pl->bil_xy, (size_t)(res[1] ? res[1] : 1u) * PB_TILE * PB_TILE * sizeof(PbBilCoord));
        if (e == hipSuccess) e = pb_tmp_alloc((void**)&pl->bil_fix_xy, (size_t)(np ? np : 1u) * stride * sizeof(PbBilCoord));
        if (e == hipSuccess) {
            const dim3 grid(4u * pl->n_tiles), block(PB_BLOCK), fgrid((np + PB_BLOCK - 1) / PB_BLOCK);
            if (dbl) {
                hipLaunchKernelGGL(pb_bilinear_coord_kernel<PB_KIND_EYE_L>, grid, block, 0, 0, P, pl->table, pl->bil_xy);
                hipLaunchKernelGGL(pb_bilinear_coord_kernel<PB_KIND_EYE_R>, grid, block, 0, 0, P, pl->table_r, pl->bil_xy);
                if (np) {
                    hipLaunchKernelGGL(pb_bilinear_fix_coord_kernel<PB_KIND_EYE_L>, fgrid, block, 0, 0, P, pl->fix_px, (int)np, pl->bil_fix_xy, 2, 0);
                    hipLaunchKernelGGL(pb_bilinear_fix_coord_kernel<PB_KIND_EYE_R>, fgrid, block, 0, 0, P, pl->fix_px, (int)np, pl->bil_fix_xy, 2, 1);
                }
            } else if (P.src.kind == PB_KIND_PANO) {
                hipLaunchKernelGGL(pb_bilinear_coord_kernel<PB_KIND_PANO>, grid, block, 0, 0, P, pl->table, pl->bil_xy);
                if (np) hipLaunchKernelGGL(pb_bilinear_fix_coord_kernel<PB_KIND_PANO>, fgrid, block, 0, 0, P, pl->fix_px, (int)np, pl->bil_fix_xy, 1, 0);
            } else {
                hipLaunchKernelGGL(pb_bilinear_coord_kernel<PB_KIND_CAMERA>, grid, block, 0, 0, P, pl->table, pl->bil_xy);
                if (np) hipLaunchKernelGGL(pb_bilinear_fix_coord_kernel<PB_KIND_CAMERA>, fgrid, block, 0, 0, P, pl->fix_px, (int)np, pl->bil_fix_xy, 1, 0);
            }
            // which way each slot is walked (lanes along the direction the source position moves least), slots walked by rows transposed
            // ... and whether its taps need the guards at all (PB_TILE_TAB_PLAIN); an eye's taps stay in its half of the frame
            hipLaunchKernelGGL(pb_bilinear_orient_kernel, dim3(pl->n_tiles), dim3(256), 0, 0, pl->table, pl->bil_xy, P.src.height, P.src.width, 0,
                               dbl ? P.src_eye_w : P.src.width, pb_knob("PB_BIL_OFF", 0), pl->saved_l);
            if (dbl)
                hipLaunchKernelGGL(pb_bilinear_orient_kernel, dim3(pl->n_tiles), dim3(256), 0, 0, pl->table_r, pl->bil_xy, P.src.height, P.src.width, P.src_eye_w,
                                   P.src.width, pb_knob("PB_BIL_OFF", 0), pl->saved_r);
            e = hipGetLastError();  // (not waited for: the launch builder's read-backs follow on the same stream and report a failure)
        }
    }
    if (e != hipSuccess) {
        pb_table_release(pl->device, pl->bil_tiles);
        pb_table_release(pl->device, pl->bil_xy);
        pb_table_release(pl->device, pl->bil_fix_xy);
        pl->bil_tiles = nullptr;
        pl->bil_xy = pl->bil_fix_xy = nullptr;
        return pb_fail(PB_ERR_HIP, std::string("bilinear tables: ") + hipGetErrorString(e));
    }
    pl->n_bil_tiles = res[0];
    pl->n_bil_slots = res[1];
    return PB_OK;
}

// Runs once per plan on the current device (synchronously, default stream):
//  1. validity thresholds of a camera / double destination by bisection with the exact predicate;
//  2. per-tile polynomial models (pb_model_kernel);
//  3. certification: the hot path's index is compared with the faithful one for EVERY pixel; differing
//     pixels / tiles become the plan's fix list (pb_certify_kernel).
static int pb_plan_prepare_on_device(pb_plan* pl) {
    PbParams& P = pl->P;
    long long* scratch = nullptr;
    PB_HIP(hipGetDevice(&pl->device));
    PB_HIP(pb_tmp_alloc((void**)&scratch, 16 * sizeof(long long)));
    int rc = PB_OK;
    do {
        if (P.dst.kind != PB_KIND_PANO) {
            hipLaunchKernelGGL(pb_threshold_kernel, dim3(1), dim3(128), 0, 0, P, scratch);
            // (a chain of ~10 dependent square roots and arcsines, and the kernel's launch: a single source's tile tables are allocated meanwhile)
            if (P.src.kind != PB_KIND_DOUBLE && pb_fast_possible(P)) {
                const unsigned nt0 = pb_num_tiles(P);
                if (pb_tmp_alloc((void**)&pl->table, (size_t)nt0 * sizeof(PbTileEntry)) != hipSuccess ||
                    pb_tmp_alloc((void**)&pl->fail_tiles, (size_t)nt0 * sizeof(int32_t)) != hipSuccess ||
                    pb_tmp_alloc((void**)&pl->fix_px, (size_t)nt0 * PB_TILE_FAIL_LIMIT * sizeof(int32_t)) != hipSuccess) { rc = PB_ERR_HIP; break; }
            }
            long long thr[4];
            if (hipMemcpy(thr, scratch, sizeof(thr), hipMemcpyDeviceToHost) != hipSuccess) { rc = PB_ERR_HIP; break; }
            P.inv_lo[0] = thr[0]; P.inv_hi[0] = thr[1];
            P.inv_lo[1] = thr[2]; P.inv_hi[1] = thr[3];
        }
        P.thresholds_ready = 1;
        if (P.src.kind == PB_KIND_DOUBLE && P.dst.kind == PB_KIND_PANO && P.n_rot == 0) {
            // separable path: the row / column tables (their taps are checked against the faithful ones when first needed: pb_sep_verified)
            if (pb_tmp_alloc((void**)&pl->sep_rows, (size_t)P.dst.height * sizeof(PbSepRow)) != hipSuccess ||
                pb_tmp_alloc((void**)&pl->sep_cols, (size_t)P.dst.width * sizeof(PbSepCol)) != hipSuccess) { rc = PB_ERR_HIP; break; }
            hipLaunchKernelGGL(pb_sep_tables_kernel, dim3(pb_blocks((unsigned long long)P.dst.height + P.dst.width)), dim3(PB_BLOCK), 0, 0,
                               P, pl->sep_rows, pl->sep_cols);
            pl->sep_ready = 1;
            pl->sep_checked = 0;
            pb_sep_setup(pl);
        }
        if (P.src.kind == PB_KIND_DOUBLE) {
            // two certified tile tables (one per eye) + the weight class of every tile
            if (!pb_fast_possible_dims(P)) break;
            const unsigned ntiles = pb_num_tiles(P);
            const unsigned cap = 2u * ntiles * PB_TILE_FAIL_LIMIT;
            if (pb_tmp_alloc((void**)&pl->table, (size_t)ntiles * sizeof(PbTileEntry)) != hipSuccess ||
                pb_tmp_alloc((void**)&pl->table_r, (size_t)ntiles * sizeof(PbTileEntry)) != hipSuccess ||
                pb_tmp_alloc((void**)&pl->fail_tiles, (size_t)2 * ntiles * sizeof(int32_t)) != hipSuccess ||
                pb_tmp_alloc((void**)&pl->fix_px, (size_t)cap * sizeof(int32_t)) != hipSuccess) { rc = PB_ERR_HIP; break; }
            unsigned* counters = reinterpret_cast<unsigned*>(scratch + 4);
            if (hipMemsetAsync(counters, 0, 12 * sizeof(unsigned), 0) != hipSuccess) { rc = PB_ERR_HIP; break; }
            const dim3 grid(pb_hot_blocks(P)), block(64 * PB_TILE_WAVES);
            // (unrotated panorama destination: the separable path's column table holds the sine / cosine of every column's longitude)
            const double* col_sc = (P.dst.kind == PB_KIND_PANO && P.n_rot == 0 && pl->sep_cols) ? reinterpret_cast<const double*>(pl->sep_cols) : nullptr;
            hipLaunchKernelGGL(pb_model_kernel<PB_KIND_EYE_L>, grid, block, 0, 0, P, pl->table);
            hipLaunchKernelGGL(pb_window_kernel<PB_KIND_EYE_L>, grid, block, 0, 0, P, pl->table);
            PB_LAUNCH_BY_ROT(P.n_rot, pb_certify_kernel, PB_KIND_EYE_L, grid, block, 0, 0, P, pl->table, pl->fail_tiles, pl->fix_px, cap, counters, col_sc);
            hipLaunchKernelGGL(pb_model_kernel<PB_KIND_EYE_R>, grid, block, 0, 0, P, pl->table_r);
            hipLaunchKernelGGL(pb_window_kernel<PB_KIND_EYE_R>, grid, block, 0, 0, P, pl->table_r);
            PB_LAUNCH_BY_ROT(P.n_rot, pb_certify_kernel, PB_KIND_EYE_R, grid, block, 0, 0, P, pl->table_r, pl->fail_tiles, pl->fix_px, cap, counters, col_sc);
            hipLaunchKernelGGL(pb_count_flags_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, 0, pl->table, ntiles, counters);
            hipLaunchKernelGGL(pb_count_flags_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, 0, pl->table_r, ntiles, counters);
            const unsigned lat_capacity = ntiles < 65536u ? ntiles : 65536u;  // <= 512 MiB of latitudes
            hipLaunchKernelGGL(pb_double_pair_kernel, grid, block, 0, 0, P, pl->table, pl->table_r, pl->sep_ready ? pl->sep_rows : nullptr,
                               pl->fail_tiles, counters, lat_capacity);
            unsigned res[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (hipMemcpy(res, counters, sizeof(res), hipMemcpyDeviceToHost) != hipSuccess) { rc = PB_ERR_HIP; break; }
            pl->n_lat_tiles = res[8] < lat_capacity ? res[8] : lat_capacity;
            if (pl->n_lat_tiles) {
                if (pb_tmp_alloc((void**)&pl->lat_tab, (size_t)pl->n_lat_tiles * PB_LAT_TILE_DOUBLES * sizeof(double)) != hipSuccess) { rc = PB_ERR_HIP; break; }
                hipLaunchKernelGGL(pb_double_lat_kernel, grid, block, 0, 0, P, pl->table, pl->lat_tab);
                if (hipGetLastError() != hipSuccess) { rc = PB_ERR_HIP; break; }  // (not waited for: see the fix tables below)
            }
            {   // faithful taps of failed tiles and fix pixels, looked up per frame
                const unsigned nf = res[1], np = res[0] > cap ? cap : res[0];
                // a geometry the models mostly cannot follow is not worth gigabytes of stored taps: faithful kernel
                if ((size_t)nf * PB_TILE * PB_TILE * sizeof(PbDoubleFix) > ((size_t)1 << 30)) break;
                if (pb_tmp_alloc((void**)&pl->dbl_tile_fix, (size_t)(nf ? nf : 1) * PB_TILE * PB_TILE * sizeof(PbDoubleFix)) != hipSuccess ||
                    pb_tmp_alloc((void**)&pl->dbl_px_fix, (size_t)(np ? np : 1) * sizeof(PbDoubleFix)) != hipSuccess) { rc = PB_ERR_HIP; break; }
                const unsigned blocks = 4u * nf + (np + PB_BLOCK - 1) / PB_BLOCK;
                if (blocks) {
                    hipLaunchKernelGGL(pb_double_tables_kernel, dim3(blocks), dim3(PB_BLOCK), 0, 0, P, pl->table_r, pl->fail_tiles, (int)nf, pl->fix_px,
                                       (int)np, pl->dbl_tile_fix, pl->dbl_px_fix);
                    // (not waited for: the budget pass and the launch-order pass queue up behind it, and their readback reports a failure)
                    if (hipGetLastError() != hipSuccess) { rc = PB_ERR_HIP; break; }
                }
            }
            pl->n_tiles = ntiles;
            pl->n_lean_tiles = res[4];
            pl->n_black_tiles = res[5];
            pl->n_direct_tiles = res[6];
            pl->n_row_weight_tiles = res[7];
            pl->n_fix_px = res[0] > cap ? cap : res[0];
            pl->n_fail_tiles = res[1];
            pl->diff_pixels = res[2];
            if (pl->bil_wanted && pb_build_bilinear_list(pl) != PB_OK) { rc = PB_ERR_HIP; break; }
            pl->dbl_ready = 1;
            break;
        }
        PB_STAGE("thresholds");
        if (!pb_fast_possible(P)) break;
        const unsigned ntiles = pb_num_tiles(P);
        const unsigned cap = ntiles * PB_TILE_FAIL_LIMIT;
        if (!pl->table &&  // (not allocated beside the threshold kernel above)
            (pb_tmp_alloc((void**)&pl->table, (size_t)ntiles * sizeof(PbTileEntry)) != hipSuccess ||
             pb_tmp_alloc((void**)&pl->fail_tiles, (size_t)ntiles * sizeof(int32_t)) != hipSuccess ||
             pb_tmp_alloc((void**)&pl->fix_px, (size_t)cap * sizeof(int32_t)) != hipSuccess)) { rc = PB_ERR_HIP; break; }
        PB_STAGE("malloc3");
        unsigned* counters = reinterpret_cast<unsigned*>(scratch + 4);
        if (hipMemsetAsync(counters, 0, 8 * sizeof(unsigned), 0) != hipSuccess) { rc = PB_ERR_HIP; break; }
        const dim3 grid(pb_hot_blocks(P)), block(64 * PB_TILE_WAVES);
        if (P.src.kind == PB_KIND_PANO) {
            hipLaunchKernelGGL(pb_model_kernel<PB_KIND_PANO>, grid, block, 0, 0, P, pl->table);
            hipLaunchKernelGGL(pb_window_kernel<PB_KIND_PANO>, grid, block, 0, 0, P, pl->table);
            PB_LAUNCH_BY_ROT(P.n_rot, pb_certify_kernel, PB_KIND_PANO, grid, block, 0, 0, P, pl->table, pl->fail_tiles, pl->fix_px, cap, counters);
        } else {
            double* col_sc = nullptr;  // (unrotated panorama destination: one sine / cosine per column instead of one per pixel)
            if (P.dst.kind == PB_KIND_PANO && P.n_rot == 0 && pb_tmp_alloc((void**)&col_sc, (size_t)P.dst.width * 2 * sizeof(double)) == hipSuccess)
                hipLaunchKernelGGL(pb_col_sincos_kernel, dim3((P.dst.width + 255) / 256), dim3(256), 0, 0, P, col_sc);
            hipLaunchKernelGGL(pb_model_kernel<PB_KIND_CAMERA>, grid, block, 0, 0, P, pl->table);
            hipLaunchKernelGGL(pb_window_kernel<PB_KIND_CAMERA>, grid, block, 0, 0, P, pl->table);
            PB_LAUNCH_BY_ROT(P.n_rot, pb_certify_kernel, PB_KIND_CAMERA, grid, block, 0, 0, P, pl->table, pl->fail_tiles, pl->fix_px, cap, counters, (const double*)col_sc);
            if (col_sc) {
                (void)hipDeviceSynchronize();
                pb_tmp_free(col_sc);
            }
        }
        PB_STAGE("enqueue+colsync");
        hipLaunchKernelGGL(pb_count_flags_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, 0, pl->table, ntiles, counters);
        unsigned res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpy(res, counters, sizeof(res), hipMemcpyDeviceToHost) != hipSuccess) { rc = PB_ERR_HIP; break; }
        PB_STAGE("certify-readback");
        pl->n_lean_tiles = res[4];
        pl->n_black_tiles = res[5];
        pl->n_direct_tiles = res[6];
        pl->n_fix_px = res[0] > cap ? cap : res[0];
        pl->n_fail_tiles = res[1];
        pl->diff_pixels = res[2];
        pl->n_tiles = ntiles;
        {   // exact-index tables for the windowed hot kernel
            const unsigned nf = pl->n_fail_tiles, np = pl->n_fix_px;
            if (pb_tmp_alloc((void**)&pl->idx_tab, (size_t)(nf ? nf : 1) * PB_TILE * PB_TILE * sizeof(int32_t)) != hipSuccess ||
                pb_tmp_alloc((void**)&pl->fix_idx, (size_t)(np ? np : 1) * sizeof(int32_t)) != hipSuccess) { rc = PB_ERR_HIP; break; }
            const unsigned blocks = 4u * nf + (np + PB_BLOCK - 1) / PB_BLOCK;
            if (blocks) {
                if (P.src.kind == PB_KIND_PANO)
                    hipLaunchKernelGGL(pb_fix_tables_kernel<PB_KIND_PANO>, dim3(blocks), dim3(PB_BLOCK), 0, 0, P, pl->table, pl->fail_tiles, (int)nf,
                                       pl->fix_px, (int)np, pl->idx_tab, pl->fix_idx);
                else
                    hipLaunchKernelGGL(pb_fix_tables_kernel<PB_KIND_CAMERA>, dim3(blocks), dim3(PB_BLOCK), 0, 0, P, pl->table, pl->fail_tiles, (int)nf,
                                       pl->fix_px, (int)np, pl->idx_tab, pl->fix_idx);
                // (not waited for: the budget pass and the launch-order pass queue up behind it, and their readback reports a failure)
                if (hipGetLastError() != hipSuccess) { rc = PB_ERR_HIP; break; }
            }
        }
        PB_STAGE("fix-tables");
        if (pl->bil_wanted && pb_build_bilinear_list(pl) != PB_OK) { rc = PB_ERR_HIP; break; }
        pl->fast_ready = 1;
    } while (0);
    if (rc != PB_OK) {
        g_err = std::string("plan preparation on device failed: ") + hipGetErrorString(hipGetLastError());
        (void)hipDeviceSynchronize();
        pb_tmp_free(pl->table); pb_tmp_free(pl->fail_tiles); pb_tmp_free(pl->fix_px); pb_tmp_free(pl->idx_tab); pb_tmp_free(pl->fix_idx);
        pl->idx_tab = nullptr; pl->fix_idx = nullptr;
        pb_tmp_free(pl->sep_rows); pb_tmp_free(pl->sep_cols); pb_tmp_free(pl->table_r); pb_tmp_free(pl->lat_tab);
        pb_tmp_free(pl->dbl_tile_fix); pb_tmp_free(pl->dbl_px_fix); pb_table_release(pl->device, pl->bil_tiles); pb_table_release(pl->device, pl->bil_xy); pb_table_release(pl->device, pl->bil_fix_xy);
        pl->dbl_tile_fix = nullptr; pl->dbl_px_fix = nullptr; pl->bil_tiles = nullptr; pl->n_bil_tiles = 0;
        pl->bil_xy = pl->bil_fix_xy = nullptr; pl->n_bil_slots = 0;
        pb_sep_release(pl);
        pl->table_r = nullptr; pl->lat_tab = nullptr; pl->sep_ready = 0; pl->dbl_ready = 0;
        pl->table = nullptr; pl->fail_tiles = nullptr; pl->fix_px = nullptr; pl->sep_rows = nullptr; pl->sep_cols = nullptr;
    }
    PB_STAGE("bil-list");
    // (the scratch block's next user is ordered behind these kernels by the NULL stream; pb_plan_prepare_full's budget pass ends the
    //  preparation with the wait - a failed one waits here, its tables have just been freed)
    if (rc != PB_OK || !(pl->fast_ready || pl->dbl_ready)) (void)hipDeviceSynchronize();
    pb_tmp_free(scratch);
    PB_STAGE("final-sync");
    return rc;
}

static bool pb_use_fast(const pb_plan* plan) { return plan->fast_ready && plan->mode != PB_MODE_FAITHFUL; }

// hot kernel + fix kernel on `st`; OUT 0 = frames, OUT 1 = int32 index map
template <int OUT>
static void pb_launch_fast(const pb_plan* pl, const uint8_t* src, uint8_t* dst, int n_frames, unsigned long long ss,
                           unsigned long long ds, int32_t* idx_out, hipStream_t st) {
    const PbParams& P = pl->P;
    const dim3 grid(pb_hot_blocks(P)), block(64 * PB_TILE_WAVES);
    const unsigned fix_blocks = 4u * pl->n_fail_tiles + (pl->n_fix_px + PB_BLOCK - 1) / PB_BLOCK;
    // the windowed kernel needs 16-byte aligned frames (LDS-DMA row segments); PB_MODE_FAST_DIRECT and
    // unaligned frames take the direct-gather hot kernel + the fix kernel
    const bool windowed = OUT == 0 && pl->ltable && pl->P_dev && pl->mode != PB_MODE_FAST_DIRECT && P.src.width < 32768 && P.src.height < 32768 &&
                          ((((uintptr_t)src) | ss) & 15u) == 0;
    if (windowed) {
        // one launch per frame: failed tiles and fix pixels are looked up in the plan's exact-index tables by the
        // hot waves themselves (pb_kernels_tile.hpp)
        // frames are a grid dimension, frame-major; a frame's share of the grid is a multiple of 8 workgroups so that a
        // tile group keeps its XCD residue in every frame
        const unsigned gpf = pl->launch_groups;
        static const unsigned wpw = [] { const int v = pb_knob("PB_WPW", PB_WAVES_PER_WG); return (v == 1 || v == 2) ? (unsigned)v : 4u; }();
        const dim3 wblock(64u * wpw);
        const size_t lds = pb_window_lds_bytes(P) / PB_TILE_WAVES * wpw + (size_t)pb_knob("PB_LDS_PAD", 0);  // (the pad: occupancy experiments, -DPB_ABLATION only)
        const unsigned wpf = gpf * (4u / wpw);
        const int per_launch = (int)(0x7FFFFFFFu / wpf);  // grid limit: absurdly long batches go in several launches
        for (int f0 = 0; f0 < n_frames; f0 += per_launch) {
            const int nf = n_frames - f0 < per_launch ? n_frames - f0 : per_launch;
            const dim3 bgrid(wpf * (unsigned)nf);
            const uint8_t* sf = src + (unsigned long long)f0 * ss;
            uint8_t* df = dst + (unsigned long long)f0 * ds;
#define PB_LAUNCH_WIN(KIND)                                                                                                   \
    hipLaunchKernelGGL((pb_hot_win_kernel<KIND>), bgrid, wblock, lds, st, (const PbParams*)pl->P_dev, pb_hot_of_host(P), pl->ltable, sf, df, gpf, ss, ds, pl->idx_tab, \
                       pl->fix_px, pl->fix_idx, (unsigned)nf, PbNoFrameTab{0})
            if (P.src.kind == PB_KIND_PANO) PB_LAUNCH_WIN(PB_KIND_PANO);
            else PB_LAUNCH_WIN(PB_KIND_CAMERA);
        }
#undef PB_LAUNCH_WIN
        return;
    }
    // the direct-gather hot kernel + the fix kernel (index maps, PB_MODE_FAST_DIRECT, frames LDS-DMA cannot address)
    if (P.src.kind == PB_KIND_PANO)
        hipLaunchKernelGGL((pb_hot_kernel<PB_KIND_PANO, OUT>), grid, block, 0, st, P, pl->table, src, dst, n_frames, ss, ds, idx_out);
    else
        hipLaunchKernelGGL((pb_hot_kernel<PB_KIND_CAMERA, OUT>), grid, block, 0, st, P, pl->table, src, dst, n_frames, ss, ds, idx_out);
    if (fix_blocks)
        hipLaunchKernelGGL((pb_fix_kernel<OUT>), dim3(fix_blocks), dim3(PB_BLOCK), 0, st, P, pl->fail_tiles, (int)pl->n_fail_tiles,
                           pl->fix_px, (int)pl->n_fix_px, pl->idx_tab, pl->fix_idx, src, dst, n_frames, ss, ds, idx_out);
}

template <int KIND>
static void pb_launch_faithful_remap(const PbParams& P, const uint8_t* src, uint8_t* dst, int n_frames, unsigned long long ss,
                                     unsigned long long ds, hipStream_t st) {
    const unsigned long long npx = (unsigned long long)P.dst.height * P.dst.width;
    const int aligned = (((uintptr_t)dst | ds) & 3u) == 0;
    PB_LAUNCH_BY_ROT(P.n_rot, pb_remap_kernel, KIND, dim3(pb_blocks((npx + PB_PX - 1) / PB_PX)), dim3(PB_BLOCK), 0, st, P, src, dst,
                       n_frames, ss, ds, aligned);
}

static int pb_remap_launch(const pb_plan* plan, const uint8_t* src_dev, uint8_t* dst_dev, int n_frames, size_t src_frame_stride,
                           size_t dst_frame_stride, hipStream_t st);

// The LDS window budget of a plan (bytes per wave).  Smaller windows let more workgroups share a CU (the hot
// kernels are latency x concurrency bound) but push tiles with larger windows onto the direct-gather path; which
// side wins depends on the geometry.  The classification only decides the PATH a tile takes, never its pixels
// (same model, same anchors, same tables), so a budget change cannot change a byte.
#define PB_DEFAULT_EXP 0
#define PB_DEFAULT_WIN_BUDGET 7168  // 5 workgroups per CU: fastest or within noise of the fastest on every BASELINE geometry (round-2 sweep r2o / r2p:
                                    // c1 14.3 us vs 15.1-16 elsewhere, c3 38.4, c5 86.4, c2 flat 44.1-44.6 from 7 to 12 KiB, batches 33.0-33.5 vs 34.9 at 12 KiB)
static int pb_clamp_budget(int budget) {
    budget &= ~15;
    if (budget < PB_DIRECT_LDS_BYTES) budget = PB_DIRECT_LDS_BYTES;
    if (budget > PB_WINLDS_MAX) budget = PB_WINLDS_MAX;
    return budget;
}

// The hot kernel's launch order: the tile entries copied into the order the waves of a launch take them, so that a
// wave's entry is found by its slot (no tile arithmetic) and the order is the plan's to choose.  Workgroup ids keep
// their XCD residue (round-robin dispatch); where the grid divides into 256x256-px super-tiles (16 workgroups), XCD x
// takes the super-tiles x, x + 8, ... of a ROW of super-tiles, and the chip walks through the rows together (every
// order that gave each XCD its own compact region - row bands, column strips, a Hilbert curve cut into eight - measured
// 5-15 % slower, in batches too, and so did every order that broke a column's vertical neighbours apart).  Which row comes
// next is the plan's choice: a launch that ENDS on its cheapest rows drains faster.  Where the rows differ (a fisheye
// output: black corners above and below, the dense centre between), the walk starts at the heaviest row and moves
// outwards on two fronts, heavier neighbour first: c2 44.1 -> 41.8 us (ending on the centre rows instead: 46.7).  Rows
// of even cost keep the plain top-to-bottom order (two fronts cost c3 10 %, c1 2 %) unless their SUPER-TILES differ (clusters of
// failed and direct-gather tiles): then the super-tiles go heaviest first, dealt round-robin (c3 37.9 -> 35.1 us).  Double-fisheye
// plans walk top to bottom and let columns of super-tiles change XCD when one XCD runs ahead (the policy is spelled out where it
// is applied, below).  -DPB_ABLATION builds: PB_ORDER=1 forces top-to-bottom.  Synchronous.
// class_dev / class_out: the budget pass's counters still on the device (pb_classify_under_budget keep_dev) and where their first two
// words go - fetched with the cost pass's readback when there is one, by themselves otherwise.
static int pb_build_launch_table(pb_plan* pl, const bool bil = false, const unsigned* class_dev = nullptr, unsigned* class_out = nullptr) {
    PbParams& P = pl->P;
    PbTileEntry*& out_table = bil ? pl->ltable_bil : pl->ltable;
    unsigned& out_groups = bil ? pl->launch_groups_bil : pl->launch_groups;
    if (!pl->fast_ready && !pl->dbl_ready) return PB_OK;  // (a double-fisheye plan: the left eye's table + PB_TILE_SOLO entries)
    const unsigned tiles_x = (P.dst.width + PB_TILE - 1) / PB_TILE, tiles_y = (P.dst.height + PB_TILE - 1) / PB_TILE;
    const unsigned gx = (tiles_x + 1) / 2, gy = (tiles_y + 1) / 2;
    static const unsigned U = [] { const int v = pb_knob("PB_UNIT", 4); return (v == 2 || v == 8 || v == 16) ? (unsigned)v : 4u; }();  // workgroups per unit side
    static const unsigned UYk = [] { const int v = pb_knob("PB_UNIT_Y", 0); return (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ? (unsigned)v : 0u; }();  // experiments: unit height in workgroups (0: square)
    const unsigned UY = UYk ? UYk : U;
    const bool units = gx % U == 0 && gy % UY == 0 && (gx / U) * (gy / UY) >= 16u;
    static const int order_mode = pb_knob("PB_ORDER", 0);
    std::vector<int> unit_of;
    int units_per_xcd = 0;
    unsigned n_groups = (gx * gy + 7u) & ~7u;
    if (units) {
        const unsigned sgx = gx / U, sgy = gy / UY, ns = sgx * sgy;
        units_per_xcd = (int)((ns + 7u) / 8u);
        n_groups = 8u * (unsigned)units_per_xcd * U * UY;
        // the walk: rows of super-tiles top to bottom, XCD = position in the walk mod 8 (plain), unless the plan's tiles say
        // that work is unevenly spread - then the launch starts on its heaviest part and ENDS on its cheapest, which drains fast:
        //   rows differ (max / min > 1.45; a fisheye output's black-cornered edges against its dense centre, a panorama's
        //   pole rows against its rim rows): from the heaviest row outwards on two fronts, heavier neighbour first - the chip
        //   still walks through whole rows together (c2 44.1 -> 41.8 us, c1 14.1 -> 13.0);
        //   rows alike but super-tiles differ (max / mean > 1.7; clusters of failed and direct-gather tiles): super-tiles
        //   heaviest first, dealt round-robin (c3 37.9 -> 35.1 us, batches 32.0 -> 30.7).
        // Cost of a tile from its class and window: a wave's measured life (experiments/diag_trace.py) is 3 us on a black tile,
        // 4.2 + 0.4 per KiB of window on a window tile, 9.5 + 0.03 per source column on a direct-gather tile.
        std::vector<unsigned> seq(ns);  // the walk: super-tile ids in launch order
        for (unsigned S = 0; S < ns; ++S) seq[S] = S;
        std::vector<float> unit_cost;
        bool row_walk = true;  // seq is a sequence of whole rows
        const int walk = (order_mode >= 1 && order_mode <= 3) ? order_mode : pl->walk;  // (PB_ORDER=1 / 2 / 3, diagnostic build: force the plain walk / rows outwards / heaviest first)
        if (walk != 1 && sgy >= 4 && (ns >= 128u || walk != 0)) {  // (fewer than 16 super-tiles per XCD: too coarse to reorder - a 3072x2048 output measured 3-4 % slower)
            std::vector<unsigned> fixed(ns + 4u, 0u);
            unsigned* cost_dev = nullptr;
            if (pb_tmp_alloc((void**)&cost_dev, (ns + 4u) * sizeof(unsigned)) != hipSuccess) {
                pb_table_release(pl->device, out_table);  // (the old table may be classified under another budget)
                out_table = nullptr;
                out_groups = 0;
                return pb_fail(PB_ERR_HIP, "launch table: out of device memory");
            }
            (void)hipMemsetAsync(cost_dev, 0, ns * sizeof(unsigned), 0);
            hipLaunchKernelGGL(pb_unit_cost_kernel, dim3((pl->n_tiles + 255) / 256), dim3(256), 0, 0, pl->table, pl->n_tiles, tiles_x, 2u * U, sgx, cost_dev,
                               pl->dbl_ready ? pl->table_r : nullptr, 2u * UY, bil ? 1 : 0, class_dev, ns);
            PB_STAGE("lt-host0");
            const hipError_t ce = hipMemcpy(fixed.data(), cost_dev, (ns + (class_dev ? 4u : 0u)) * sizeof(unsigned), hipMemcpyDeviceToHost);
            if (class_dev && class_out && ce == hipSuccess) {
                class_out[0] = fixed[ns];
                class_out[1] = fixed[ns + 1];
                class_dev = nullptr;  // (delivered)
            }
            PB_STAGE("lt-cost-readback");
            if (ce != hipSuccess) (void)hipDeviceSynchronize();
            pb_tmp_free(cost_dev);
            if (ce != hipSuccess) {
                pb_table_release(pl->device, out_table);
                out_table = nullptr;
                out_groups = 0;
                return pb_fail(PB_ERR_HIP, std::string("launch table: ") + hipGetErrorString(ce));
            }
            std::vector<float> row_cost(sgy, 0.f);
            unit_cost.assign(ns, 0.f);
            for (unsigned S = 0; S < ns; ++S) {
                unit_cost[S] = (float)fixed[S] / 1024.0f;
                row_cost[S / sgx] += unit_cost[S];
            }
            const float rmax = *std::max_element(row_cost.begin(), row_cost.end()), rmin = *std::min_element(row_cost.begin(), row_cost.end());
            const float umax = *std::max_element(unit_cost.begin(), unit_cost.end());
            float usum = 0.f;
            for (float c : unit_cost) usum += c;
            if (walk == 0 && pl->dbl_ready) {
                // double-fisheye sources keep the plain walk (two fronts measured 1.5 % slower on c5); their problem is another one, below
            } else if (walk == 2 || (walk == 0 && rmax > 1.45f * rmin)) {
                const unsigned top = (unsigned)(std::max_element(row_cost.begin(), row_cost.end()) - row_cost.begin());
                int up = (int)top - 1;
                unsigned down = top + 1, k = 0;
                std::vector<unsigned> row_seq(sgy);
                row_seq[k++] = top;
                while (k < sgy) {
                    const bool take_down = down < sgy && (up < 0 || row_cost[down] >= row_cost[(unsigned)up]);
                    row_seq[k++] = take_down ? down++ : (unsigned)up--;
                }
                for (unsigned k2 = 0; k2 < sgy; ++k2)
                    for (unsigned i = 0; i < sgx; ++i) seq[k2 * sgx + i] = row_seq[k2] * sgx + i;
            } else if (walk == 3 || (walk == 0 && umax * (float)ns > 1.7f * usum)) {
                row_walk = false;
                std::stable_sort(seq.begin(), seq.end(), [&](unsigned a, unsigned b) { return unit_cost[a] > unit_cost[b]; });
            }
        }
        unit_of.assign((size_t)8 * units_per_xcd, -1);
        std::vector<int> filled(8, 0);
        float imbalance = 1.f;  // busiest / idlest XCD if every column of super-tiles kept "XCD = column mod 8"
        if (!unit_cost.empty() && sgx % 8 == 0) {
            float tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (unsigned S = 0; S < ns; ++S) tot[(S % sgx) & 7u] += unit_cost[S];
            imbalance = *std::max_element(tot, tot + 8) / std::max(1e-6f, *std::min_element(tot, tot + 8));
        }
        if (walk == 0 && row_walk && sgx % 8 == 0 && !unit_cost.empty() && (pl->dbl_ready || imbalance > 1.15f)) {
            // A stitch's expensive tiles - the seams where both eyes contribute, the eyes' rims - stand in COLUMNS of the
            // output, and "XCD = column mod 8" hands whole seams to the same XCDs row after row: the chip waits for them
            // (c5 76.5 us; super-tiles dealt at random 66.6).  Rows are still walked together and a column keeps its XCD from
            // row to row (vertical neighbours share an L2), but after every row the busiest XCD hands a column to the idlest
            // when it is more than half a super-tile ahead: c5 76.5 -> 64.6 us, c5shard 68.9 -> 54.7.  Single sources get the
            // exchange only when their columns are that uneven too (busiest / idlest XCD > 1.15 under "column mod 8": a panorama
            // from a 200-degree fisheye, black columns beyond the field of view, 26.7 -> 25.8 us); on even columns it measured
            // +-0 (c2, c3) or worse (c1 +6 % under the plain walk).
            std::vector<int> owner(sgx);
            for (unsigned i = 0; i < sgx; ++i) owner[i] = (int)(i & 7u);
            float total[8] = {0, 0, 0, 0, 0, 0, 0, 0}, all = 0.f;
            for (float c : unit_cost) all += c;
            const float thresh = 0.5f * all / (float)ns;
            for (unsigned r = 0; r < sgy; ++r) {
                for (unsigned i = 0; i < sgx; ++i) {
                    const int x = owner[i];
                    const unsigned S = (seq[r * sgx] / sgx) * sgx + i;  // row r of the walk, column i
                    unit_of[(size_t)x * units_per_xcd + filled[x]++] = (int)S;
                    total[x] += unit_cost[S];
                }
                if (r + 1 == sgy) break;
                float ahead[8];  // totals as they will stand after the exchanges decided so far
                for (unsigned x = 0; x < 8; ++x) ahead[x] = total[x];
                for (int pass = 0; pass < 4; ++pass) {
                    const int a = (int)(std::max_element(ahead, ahead + 8) - ahead), b = (int)(std::min_element(ahead, ahead + 8) - ahead);
                    if (ahead[a] - ahead[b] <= thresh) break;
                    float best = 0.f;
                    int bi = -1, bj = -1;
                    for (unsigned i = 0; i < sgx; ++i)
                        for (unsigned j = 0; j < sgx; ++j)
                            if (owner[i] == a && owner[j] == b) {
                                const unsigned nr = seq[(r + 1) * sgx] / sgx;
                                const float gain = unit_cost[nr * sgx + i] - unit_cost[nr * sgx + j];  // what `a` sheds next row
                                if (gain > best) { best = gain; bi = (int)i; bj = (int)j; }
                            }
                    if (bi < 0) break;
                    std::swap(owner[bi], owner[bj]);
                    ahead[a] -= best;
                    ahead[b] += best;
                }
            }
        } else {
            for (unsigned p = 0; p < ns; ++p) unit_of[(size_t)(p & 7u) * units_per_xcd + filled[p & 7u]++] = (int)seq[p];  // XCD = position in the walk, mod 8
        }
    }
    // the new table is built aside and replaces the plan's only when it is complete: a failure leaves the plan WITHOUT a
    // launch table (ltable == nullptr, launch_groups == 0 - its launches then take the direct-gather kernels), never with a
    // half-written one or one classified under another budget
    pb_table_release(pl->device, out_table);
    out_table = nullptr;
    out_groups = 0;
    PbTileEntry* fresh = nullptr;
    int* unit_dev = nullptr;
    hipError_t e = hipSuccess;
    if (class_dev && class_out) {  // (no cost pass has brought the budget pass's class counters along)
        unsigned r4[4] = {0, 0, 0, 0};
        e = hipMemcpy(r4, class_dev, sizeof(r4), hipMemcpyDeviceToHost);
        class_out[0] = r4[0];
        class_out[1] = r4[1];
    }
    if (units && e == hipSuccess) {
        e = pb_tmp_alloc((void**)&unit_dev, unit_of.size() * sizeof(int));
        PB_STAGE("lt-host-order");
        if (e == hipSuccess) {
            if (unit_of.size() <= 1024) {  // (up to four launches that nothing waits for)
                for (size_t o = 0; o < unit_of.size(); o += 256) {
                    PbWordChunk c;
                    const size_t n = std::min<size_t>(256, unit_of.size() - o);
                    memcpy(c.v, unit_of.data() + o, n * sizeof(int));
                    hipLaunchKernelGGL(pb_store_words_kernel, dim3(1), dim3(256), 0, 0, c, unit_dev + o, (int)n);
                }
                e = hipGetLastError();
            } else {
                e = hipMemcpy(unit_dev, unit_of.data(), unit_of.size() * sizeof(int), hipMemcpyHostToDevice);
            }
        }
        PB_STAGE("lt-unit-upload");
    }
    if (bil && pl->dbl_ready) {
        // the PAIR layout (pb_kernels_tile.hpp): a two-eye tile takes two slots of a pair workgroup, so the XCDs' lists grow by what
        // their tile groups need - counted, scanned per XCD and written in place of the plain layout's one workgroup per group
        unsigned* scan = nullptr;  // n_wgs[n_groups], start[n_groups], totals[8]
        unsigned totals[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (e == hipSuccess) e = pb_tmp_alloc((void**)&scan, (2 * (size_t)n_groups + 8) * sizeof(unsigned));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pb_pair_count_kernel, dim3((n_groups + 255) / 256), dim3(256), 0, 0, P, pl->table, pl->table_r, unit_dev, units_per_xcd, n_groups, (int)U, (int)UY, scan);
            hipLaunchKernelGGL(pb_pair_scan_kernel, dim3(1), dim3(512), 0, 0, scan, n_groups, scan + n_groups, scan + 2 * (size_t)n_groups);
            e = hipMemcpy(totals, scan + 2 * (size_t)n_groups, sizeof(totals), hipMemcpyDeviceToHost);
        }
        const unsigned per_xcd = *std::max_element(totals, totals + 8), pair_groups = 8u * per_xcd;
        if (e == hipSuccess && pair_groups == 0) e = hipErrorInvalidValue;  // (a plan without tiles does not get here)
        if (e == hipSuccess) e = pb_tmp_alloc((void**)&fresh, (size_t)pair_groups * 4u * sizeof(PbTileEntry));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pb_skip_fill_kernel, dim3(pair_groups), dim3(256), 0, 0, fresh, 4u * pair_groups);
            hipLaunchKernelGGL(pb_pair_table_kernel, dim3(n_groups), dim3(64), 0, 0, P, pl->table, pl->table_r, fresh, unit_dev, units_per_xcd, n_groups, (int)U, (int)UY,
                               scan + n_groups);
            e = hipDeviceSynchronize();
        }
        pb_tmp_free(scan);
        n_groups = pair_groups;
    } else {
        const unsigned n_slots = 4u * n_groups;
        if (e == hipSuccess) e = (!bil && pb_test_alloc_fails()) ? hipErrorOutOfMemory : pb_tmp_alloc((void**)&fresh, (size_t)n_slots * sizeof(PbTileEntry));  // (the test hook counts the nearest mode's tables)
        if (e == hipSuccess) {
            PB_STAGE("lt-malloc");
            hipLaunchKernelGGL(pb_launch_table_kernel, dim3(n_groups), dim3(256), 0, 0, P, pl->table, fresh, unit_dev, units_per_xcd, n_slots, (int)U,
                               pl->dbl_ready ? pl->table_r : nullptr, (int)UY);
            e = hipDeviceSynchronize();
            PB_STAGE("lt-kernel-sync");
        }
    }
    pb_tmp_free(unit_dev);
    if (e != hipSuccess) {
        pb_table_release(pl->device, fresh);
        (void)hipGetLastError();
        return pb_fail(PB_ERR_HIP, std::string("launch table: ") + hipGetErrorString(e));
    }
    out_table = fresh;
    out_groups = n_groups;
    return PB_OK;
}

// keeps the certified flags (once) and writes the flags under `budget` into the tile tables; counts = {LEAN, DIRECT} tiles (nullptr: not
// wanted - then the call does not wait for the device).
// keep_dev: the class counters stay on the device in that block (4 words, zeroed here) for the caller to fetch with something else.
static int pb_classify_under_budget(pb_plan* pl, int budget, unsigned counts[2], unsigned* keep_dev = nullptr) {
    const unsigned nt = pl->n_tiles;
    const dim3 g((nt + 255) / 256), b(256);
    if (!pl->saved_l) {
        PB_HIP(pb_tmp_alloc((void**)&pl->saved_l, (size_t)nt * sizeof(int32_t)));
        hipLaunchKernelGGL(pb_save_flags_kernel, g, b, 0, 0, pl->table, pl->saved_l, nt);
        if (pl->dbl_ready) {
            PB_HIP(pb_tmp_alloc((void**)&pl->saved_r, (size_t)nt * sizeof(int32_t)));
            hipLaunchKernelGGL(pb_save_flags_kernel, g, b, 0, 0, pl->table_r, pl->saved_r, nt);
        }
    }
    unsigned* counters = keep_dev;
    if (!counters) PB_HIP(pb_tmp_alloc((void**)&counters, 4 * sizeof(unsigned)));
    (void)hipMemsetAsync(counters, 0, 4 * sizeof(unsigned), 0);
    if (pl->dbl_ready)
        hipLaunchKernelGGL(pb_budget_double_kernel, g, b, 0, 0, pl->table, pl->table_r, pl->saved_l, pl->saved_r, nt, budget, counters);
    else
        hipLaunchKernelGGL(pb_budget_kernel, g, b, 0, 0, pl->table, pl->saved_l, nt, budget, counters);
    if (keep_dev) {
        PB_HIP(hipGetLastError());
        return PB_OK;
    }
    if (!counts) {  // (nobody reads the class counts of this pass: no round trip - the block's next user is ordered behind the kernel by the stream)
        pb_tmp_free(counters);
        PB_HIP(hipGetLastError());
        return PB_OK;
    }
    unsigned res[4] = {0, 0, 0, 0};
    const hipError_t e = hipMemcpy(res, counters, sizeof(res), hipMemcpyDeviceToHost);
    if (e != hipSuccess) (void)hipDeviceSynchronize();
    pb_tmp_free(counters);
    PB_HIP(e);
    counts[0] = res[0];
    counts[1] = res[1];
    return PB_OK;
}

// The bilinear mode's launch-order table: the tiles classified under PB_BIL_WIN_BUDGET, ordered by the bilinear mode's costs.  Leaves
// the tile tables' flags under THAT budget: pb_apply_budget (the nearest mode's) must follow.  A failure leaves the plan without the
// table (the bilinear launches then take the float64 kernels), never with a stale one.  Synchronous.
#define PB_BIL_WIN_BUDGET PB_WINLDS_MAX
#define PB_BIL_POOL_RULE 50u  // a pool is accepted when at most 1 / PB_BIL_POOL_RULE of the tiles lose their window to it
#ifndef PB_BIL_POOL_SMALL
#define PB_BIL_POOL_SMALL 40448u  // a four-wave bilinear workgroup's small LDS pool: four of them, sixteen waves, per CU (160 KiB of LDS); two waves: half
#endif
//  // measured on MI355X (experiments/r4/budget_bil.sh): c1 28.7 us at 7 KiB, 25.2 at 12; c2 68.9 / 62.0; c5 109.8 / 103.4; c3 58.3 / 58.9
static int pb_build_bilinear_launch(pb_plan* pl) {
    if (!(pl->fast_ready || pl->dbl_ready)) return PB_OK;
    pb_table_release(pl->device, pl->ltable_bil);
    pl->ltable_bil = nullptr;
    pl->launch_groups_bil = 0;
    if (!pb_bilinear_tiles_allowed(pl->P)) return PB_OK;
    pl->bil_budget = pb_clamp_budget(PB_BIL_WIN_BUDGET);
    int rc = pb_classify_under_budget(pl, pl->bil_budget, nullptr);
    if (rc == PB_OK) rc = pb_build_launch_table(pl, true);
    if (rc != PB_OK) return rc;
    // Workgroup size and LDS pool, chosen together (a pool is accepted when at most 2 % of the tiles lose their window to it), best first:
    //   two waves, 19.75 KiB   sixteen waves per CU (four per SIMD: what the tile code's 117-120 VGPRs allow) in eight workgroups
    //   four waves, 39.5 KiB   sixteen waves in four workgroups: regions pooled over four slots fit where two slots do not (c1)
    //   two waves, 22.8 KiB    fourteen waves per CU
    //   two waves, 2 x budget  always fits: twelve waves per CU
    unsigned* counters = nullptr;
    const unsigned ng = pl->launch_groups_bil;
    unsigned res[2] = {0u, 0u};
    hipError_t e = pb_tmp_alloc((void**)&counters, 8 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemsetAsync(counters, 0, 8 * sizeof(unsigned), 0);
    if (e == hipSuccess && !pb_bil_off(1))  // direct-gather slots that can be served as two half windows
    {
        hipLaunchKernelGGL(pb_bilinear_halves_kernel, dim3(ng), dim3(256), 0, 0, pl->ltable_bil, 4u * ng, pl->bil_budget, pl->P.src.height, pl->P.src.width,
                           pb_bil_off(8) ? 1 : 0, counters);
    }
    const unsigned budget32 = (unsigned)pl->bil_budget + 32u;
    int waves = 2;
    unsigned pool = 2u * budget32;
    if (e == hipSuccess && !pb_bil_off(16)) {
        // (the one-eye slots of a virtual workgroup dealt to its two halves so that their LDS needs balance: largest with smallest)
        hipLaunchKernelGGL(pb_bilinear_balance_kernel, dim3(ng), dim3(64), 0, 0, pl->ltable_bil, ng);
        const struct { int waves; unsigned bytes; } tiers[3] = {{2, PB_BIL_POOL_SMALL / 2u}, {4, PB_BIL_POOL_SMALL}, {2, (163840u / 7u) & ~15u}};
        // the three dry passes back to back, ONE read-back (a round trip is ~25 us; a plan that takes the third tier made three)
        unsigned dry[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        e = hipMemsetAsync(counters, 0, 6 * sizeof(unsigned), 0);
        for (int t = 0; t < 3 && e == hipSuccess; ++t) {
            if (tiers[t].bytes >= (unsigned)tiers[t].waves * budget32) continue;
            hipLaunchKernelGGL(pb_bilinear_pool_kernel, dim3((ng * (4u / (unsigned)tiers[t].waves) + 127) / 128), dim3(128), 0, 0, pl->ltable_bil, ng, tiers[t].bytes, 1,
                               counters + 2 * t, tiers[t].waves);
        }
        if (e == hipSuccess) e = hipMemcpy(dry, counters, sizeof(dry), hipMemcpyDeviceToHost);
        for (int t = 0; t < 3 && e == hipSuccess; ++t) {
            if (tiers[t].bytes >= (unsigned)tiers[t].waves * budget32) continue;
            if (dry[2 * t + 1] == 0u && dry[2 * t] * PB_BIL_POOL_RULE <= pl->n_tiles) {
                waves = tiers[t].waves;
                pool = tiers[t].bytes;
                break;
            }
        }
    } else {
        waves = 4;  // (the diagnostic build's "no small pool": round 4's shape)
        pool = 4u * budget32;
    }
    if (e == hipSuccess) e = hipMemsetAsync(counters, 0, 2 * sizeof(unsigned), 0);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pb_bilinear_pool_kernel, dim3((ng * (4u / (unsigned)waves) + 127) / 128), dim3(128), 0, 0, pl->ltable_bil, ng, pool, 0, counters, waves);
        e = hipMemcpy(res, counters, sizeof(res), hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) (void)hipDeviceSynchronize();
    pb_tmp_free(counters);
    if (e != hipSuccess || res[1] != 0u) {
        pb_table_release(pl->device, pl->ltable_bil);
        pl->ltable_bil = nullptr;
        pl->launch_groups_bil = 0;
        return pb_fail(PB_ERR_HIP, e != hipSuccess ? std::string("bilinear LDS pool: ") + hipGetErrorString(e) : std::string("bilinear LDS pool: a workgroup does not fit"));
    }
    if (pl->dbl_ready) {
        const PbDblTables T = {pl->P.mrg_min, pl->P.mrg_max_safe, pl->P.mrg_max, pl->P.mrg_range, pl->sep_ready ? pl->sep_rows : nullptr, pl->lat_tab, pl->fix_px,
                               pl->bil_fix_xy, pl->dbl_tile_fix, pl->dbl_px_fix};
        if ((!pl->bil_dbl_tables && pb_tmp_alloc((void**)&pl->bil_dbl_tables, sizeof(PbDblTables)) != hipSuccess) ||
            hipMemcpy(pl->bil_dbl_tables, &T, sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
            pb_table_release(pl->device, pl->ltable_bil);
            pl->ltable_bil = nullptr;
            pl->launch_groups_bil = 0;
            return pb_fail(PB_ERR_HIP, "bilinear launch: out of device memory");
        }
    }
    pl->bil_pool_bytes = pool;
    pl->bil_waves = waves;
    return PB_OK;
}

// applies `budget` to the certified flags and rebuilds the nearest mode's launch-order table; synchronous on the default stream
static int pb_apply_budget(pb_plan* pl, int budget) {
    PbParams& P = pl->P;
    if (!(pl->fast_ready || pl->dbl_ready)) return PB_OK;
    budget = pb_clamp_budget(budget);
    // the parameter block as the launches will see it goes up BEFORE anything of the plan changes (a failure used to
    // leave the old launch table, classified under the old budget, next to the new budget)
    PbParams Q = P;
    Q.win_budget = budget;
    if (!pl->P_dev) PB_HIP(pb_tmp_alloc((void**)&pl->P_dev, sizeof(PbParams)));
    // Nothing here waits for the device by itself (a round trip is 12-25 us of a 0.5 ms plan): the class counters of the budget pass stay
    // on the device and come back with the launch-order pass's costs, the parameter block is stored by a kernel from its own arguments.
    unsigned counts[2] = {pl->n_lean_tiles, pl->n_direct_tiles};
    unsigned* class_dev = nullptr;
    PB_HIP(pb_tmp_alloc((void**)&class_dev, 4 * sizeof(unsigned)));
    PB_STAGE("ab-pdev-malloc");
    int rc = pb_classify_under_budget(pl, budget, nullptr, class_dev);
    PB_STAGE("ab-classify");
    if (rc == PB_OK) {
        hipLaunchKernelGGL(pb_store_params_kernel, dim3(1), dim3(64), 0, 0, Q, pl->P_dev);
        if (hipGetLastError() != hipSuccess) rc = pb_fail(PB_ERR_HIP, "parameter block upload failed");
    }
    PB_STAGE("ab-pdev-upload");
    if (rc != PB_OK) {
        // the table may be half reclassified: no launch table at all (launches take the direct-gather kernels, which read no budget)
        (void)hipDeviceSynchronize();  // (the counter block goes back idle)
        pb_tmp_free(pl->ltable);
        pb_tmp_free(class_dev);
        pl->ltable = nullptr;
        pl->launch_groups = 0;
        return rc;
    }
    P.win_budget = budget;
    rc = pb_build_launch_table(pl, false, class_dev, counts);
    if (rc != PB_OK) (void)hipDeviceSynchronize();
    pb_tmp_free(class_dev);  // (pb_build_launch_table has waited for the device)
    pl->n_lean_tiles = counts[0];
    pl->n_direct_tiles = counts[1];
    return rc;
}

// OPT-IN (PB_PLAN_TUNE): picks the budget by measurement - four candidates x a few launches on scratch frames
// (allocates and fills frame-sized scratch; tens of frames' worth of GPU time).  Returns the winner.
static int pb_tune_window_budget(pb_plan* pl) {
    PbParams& P = pl->P;
    int best = P.win_budget;
    if (!(pl->fast_ready || pl->dbl_ready)) return best;
    const size_t sb = 3ull * P.src.height * P.src.width, db = 3ull * P.dst.height * P.dst.width;
    if (sb + db > (1ull << 30)) return best;
    uint8_t *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    do {
        // scratch frames in rotation, more than the 256 MiB Infinity Cache in total: the launches being timed
        // must stream from HBM like real frames do, not hit a cache-resident copy
        const size_t sb16 = (sb + 255) & ~(size_t)255, db16 = (db + 255) & ~(size_t)255;
        int n_scratch = (int)(((size_t)320 << 20) / (sb16 + db16)) + 1;
        if (n_scratch < 2) n_scratch = 2;
        if (n_scratch > 8) n_scratch = 8;
        if (hipMalloc((void**)&src, n_scratch * sb16) != hipSuccess || hipMalloc((void**)&dst, n_scratch * db16) != hipSuccess) break;
        if (hipMemsetAsync(src, 0x55, n_scratch * sb16, 0) != hipSuccess) break;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
        const int cand[4] = {PB_WINLDS_MAX, 10224, 8176, 7168};  // 3, 4, 5 and 5 workgroups per CU (LDS-wise)
        float t_min[4] = {1e30f, 1e30f, 1e30f, 1e30f};
        bool failed = false;
        int launch_no = 0;
        for (int pass = 0; pass < 2 && !failed; ++pass)  // two interleaved passes: clock ramps and noise hit all alike
            for (int c = 0; c < 4 && !failed; ++c) {
                if (pb_apply_budget(pl, cand[c]) != PB_OK) { failed = true; break; }
                for (int rep = 0; rep < 4; ++rep) {  // the first launch after a reclassification is not counted
                    (void)hipEventRecord(e0, 0);
                    const int slot = launch_no++ % n_scratch;
                    if (pb_remap_launch(pl, src + slot * sb16, dst + slot * db16, 1, 0, 0, 0) != PB_OK) { failed = true; break; }
                    (void)hipEventRecord(e1, 0);
                    if (hipEventSynchronize(e1) != hipSuccess) { failed = true; break; }
                    float ms = 0.f;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < t_min[c]) t_min[c] = ms;
                }
            }
        // the library default unless another budget is at least 3 % faster (a handful of launches is a noisy yardstick)
        int bi = 3;
        for (int c = 0; c < 3; ++c)
            if (t_min[c] < t_min[3] * 0.97f && t_min[c] < t_min[bi]) bi = c;
        if (!failed) best = cand[bi];
        // then the launch order under the chosen budget: the policy's walk against the three fixed ones
        if (!failed && pb_apply_budget(pl, best) == PB_OK) {
            std::vector<float> t_walk[4];
            for (int pass = 0; pass < 3 && !failed; ++pass)
                for (int wk = 0; wk < 4 && !failed; ++wk) {
                    pl->walk = wk;
                    if (pb_build_launch_table(pl) != PB_OK) { failed = true; break; }
                    for (int rep = 0; rep < 7; ++rep) {
                        (void)hipEventRecord(e0, 0);
                        const int slot = launch_no++ % n_scratch;
                        if (pb_remap_launch(pl, src + slot * sb16, dst + slot * db16, 1, 0, 0, 0) != PB_OK) { failed = true; break; }
                        (void)hipEventRecord(e1, 0);
                        if (hipEventSynchronize(e1) != hipSuccess) { failed = true; break; }
                        float ms = 0.f;
                        (void)hipEventElapsedTime(&ms, e0, e1);
                        if (rep > 0) t_walk[wk].push_back(ms);
                    }
                }
            // medians (a launch in isolation varies by several per cent); a fixed walk must beat the policy's by 5 %
            float med[4] = {1e30f, 1e30f, 1e30f, 1e30f};
            for (int wk = 0; wk < 4 && !failed; ++wk) {
                std::sort(t_walk[wk].begin(), t_walk[wk].end());
                if (!t_walk[wk].empty()) med[wk] = t_walk[wk][t_walk[wk].size() / 2];
            }
            int bw = 0;
            for (int wk = 1; wk < 4; ++wk)
                if (med[wk] < med[0] * 0.95f && med[wk] < med[bw]) bw = wk;
            pl->walk = failed ? 0 : bw;
        }
    } while (0);
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(dst);
    return best;
}

// the parameter block of a request, as pb_plan_create_ex fills it before any device work
static void pb_params_of_request(PbParams& P, const PbEnd& dst, const double* rot3x3, int n_rot, const PbEnd& src) {
    memset(&P, 0, sizeof(PbParams));
    P.dst = dst;
    P.src = src;
    P.n_rot = n_rot;
    for (int k = 0; k < n_rot; ++k)
        for (int e = 0; e < 9; ++e) P.R[k][e] = rot3x3[9 * k + e];
    pb_derive(P);
}
// everything of a parameter block that follows from the request alone (not: budget, experiment flags, validity thresholds)
static bool pb_same_request(const PbParams& a, const PbParams& b) {
    PbParams x = a, y = b;
    for (PbParams* p : {&x, &y}) {
        p->win_budget = 0;
        p->exp_flags = 0;
        p->thresholds_ready = 0;
        p->fast_tiles = 0;
        for (int i = 0; i < 2; ++i) p->inv_lo[i] = p->inv_hi[i] = 0;
    }
    return memcmp(&x, &y, sizeof(PbParams)) == 0;
}

static double pb_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// device preparation of a plan on the current device + its window budget (0 = default; PB_WIN_BUDGET overrides
// the default for experiments); PB_PLAN_TUNE times candidates instead
static int pb_plan_prepare_full(pb_plan* pl, unsigned flags, int win_budget) {
    if (pl->fast_ready || pl->dbl_ready || pl->sep_ready || pl->device >= 0) {
        // already prepared: the budget may change, and the opt-in bilinear mode's tables may be asked for (PB_PLAN_BILINEAR) by a plan
        // that was created without them
        if ((flags & PB_PLAN_BILINEAR) && (pl->fast_ready || pl->dbl_ready) && !pl->bil_tiles) {
            pl->bil_wanted = 1;
            int rc = pb_build_bilinear_list(pl);
            if (rc == PB_OK) rc = pb_build_bilinear_launch(pl);
            // pb_build_bilinear_launch leaves the tile tables' flags under the MODE's budget: the nearest mode's classification is written
            // back (one small kernel).  Its launch table is NOT rebuilt - nothing it holds has changed, and no device memory a launch may
            // hold a pointer to is freed here.  The tile tables do pass through the other classification on the way, and a double-fisheye
            // plan's nearest kernel reads them: hence the header's rule - no launch of the plan in flight (the Python host, which builds
            // the mode's tables at its first use on whichever thread that happens, holds its own launches off: _native.py _LaunchGate).
            const int rc2 = win_budget > 0 ? pb_apply_budget(pl, win_budget) : pb_classify_under_budget(pl, pl->P.win_budget, nullptr);
            if (rc != PB_OK) return rc;
            if (rc2 != PB_OK) return rc2;
            PB_HIP(hipDeviceSynchronize());
            return PB_OK;
        }
        if (win_budget > 0) return pb_apply_budget(pl, win_budget);
        return PB_OK;
    }
    PB_STAGE("begin");
    const double t0 = pb_now_ms();
    int rc = pb_plan_prepare_on_device(pl);
    if (rc != PB_OK) {
        pl->fast_ready = 0;  // (pb_plan_prepare_on_device has released the tables): the plan is unprepared again and may be retried
        pl->device = -1;
        return rc;
    }
    int budget = win_budget > 0 ? win_budget : PB_DEFAULT_WIN_BUDGET;
    if (win_budget <= 0 && pb_knob("PB_WIN_BUDGET", 0) > 0) budget = pb_knob("PB_WIN_BUDGET", 0);
    if (pl->bil_wanted) rc = pb_build_bilinear_launch(pl);
    if (rc != PB_OK) return rc;
    PB_STAGE("bil-launch");
    rc = pb_apply_budget(pl, budget);
    if (rc != PB_OK) return rc;
    PB_HIP(hipDeviceSynchronize());
    PB_STAGE("apply-budget-end");
    pb_stage_report();
    pl->prepare_ms = pb_now_ms() - t0;
    if ((flags & PB_PLAN_TUNE) && win_budget <= 0) {
        const double t1 = pb_now_ms();
        const int best = pb_tune_window_budget(pl);
        rc = pb_apply_budget(pl, best);
        pl->tune_ms = pb_now_ms() - t1;
    }
    return rc;
}

extern "C" {

int pb_abi_version(void) { return PB_ABI_VERSION; }
int pb_math_flavour(void) { return PB_MATH_FLAVOUR; }
const char* pb_last_error(void) { return g_err.c_str(); }

int pb_init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return pb_fail(PB_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return pb_fail(PB_ERR_INVALID, "device ordinal out of range");
    PB_HIP(hipSetDevice(device));
    return PB_OK;
}
int pb_shutdown(void) {
    // the idle blocks of the plan-preparation cache (at most 96 MiB) go back to the driver; live plans keep theirs
    std::vector<void*> idle;
    {
        std::lock_guard<std::mutex> g(g_tmp.lock);
        for (const auto& b : g_tmp.idle) idle.push_back(b.ptr);
        g_tmp.idle.clear();
        g_tmp.held = 0;
    }
    for (void* p : idle) (void)hipFree(p);
    return PB_OK;
}

int pb_device_name(char* buf, size_t buflen) {
    if (!buf || !buflen) return pb_fail(PB_ERR_INVALID, "null buffer");
    int dev = 0;
    PB_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PB_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return PB_OK;
}

int pb_plan_create_ex(const pb_proj* dst, const double* rot3x3, int n_rot, const pb_proj* src, unsigned flags, int win_budget,
                      pb_plan** out) {
    std::string why;
    if (!out) return pb_fail(PB_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!pb_end_ok(dst, why, PB_ROLE_DST) || !pb_end_ok(src, why, PB_ROLE_SRC)) return pb_fail(PB_ERR_INVALID, why);
    if (n_rot < 0 || n_rot > PB_MAX_ROTATIONS) return pb_fail(PB_ERR_INVALID, "n_rot outside [0, PB_MAX_ROTATIONS]");
    if (n_rot > 0 && !rot3x3) return pb_fail(PB_ERR_INVALID, "null rotation matrices");
    if (flags & ~(unsigned)(PB_PLAN_DEFER | PB_PLAN_TUNE | PB_PLAN_MATH_SVML | PB_PLAN_MATH_LIBM | PB_PLAN_NO_BILINEAR)) return pb_fail(PB_ERR_INVALID, "unknown plan flags");
    if ((flags & PB_PLAN_MATH_SVML) && (flags & PB_PLAN_MATH_LIBM)) return pb_fail(PB_ERR_INVALID, "a plan has one math flavour");
    if (((flags & PB_PLAN_MATH_SVML) && PB_MATH_FLAVOUR != 0) || ((flags & PB_PLAN_MATH_LIBM) && PB_MATH_FLAVOUR != 1))
        return pb_fail(PB_ERR_UNSUPPORTED, PB_MATH_FLAVOUR ? "this is libphotonbend_hip_libm.so (the float64 chain runs glibc's asin / acos / atan / tan): load libphotonbend_hip.so for the AVX-512 (SVML) flavour"
                                                           : "this is libphotonbend_hip.so (the float64 chain runs NumPy's AVX-512 arcsin / arccos / arctan / tan): load libphotonbend_hip_libm.so for the libm flavour");
    flags &= ~(unsigned)(PB_PLAN_MATH_SVML | PB_PLAN_MATH_LIBM);
    if (win_budget < 0) return pb_fail(PB_ERR_INVALID, "negative window budget");
    pb_plan* pl = new (std::nothrow) pb_plan();
    if (!pl) return pb_fail(PB_ERR_INVALID, "out of host memory");
    pb_params_of_request(pl->P, pb_to_end(dst), rot3x3, n_rot, pb_to_end(src));
    pl->P.win_budget = PB_WINLDS_MAX;
    pl->P.exp_flags = PB_DEFAULT_EXP;
    pl->P.exp_flags = pb_knob("PB_EXP", PB_DEFAULT_EXP);
    pl->mode = PB_MODE_AUTO;
    pl->bil_wanted = (flags & PB_PLAN_NO_BILINEAR) ? 0 : 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    (void)hipGetLastError();
    if (ndev > 0 && !(flags & PB_PLAN_DEFER)) {
        const int rc = pb_plan_prepare_full(pl, flags, win_budget);
        if (rc != PB_OK) {
            pb_plan_destroy(pl);
            return rc;
        }
    }
    *out = pl;
    return PB_OK;
}

int pb_plan_matches(const pb_plan* plan, const pb_proj* dst, const double* rot3x3, int n_rot, const pb_proj* src) {
    std::string why;
    if (!plan) return pb_fail(PB_ERR_INVALID, "null argument");
    if (!pb_end_ok(dst, why, PB_ROLE_DST) || !pb_end_ok(src, why, PB_ROLE_SRC)) return pb_fail(PB_ERR_INVALID, why);
    if (n_rot < 0 || n_rot > PB_MAX_ROTATIONS || (n_rot > 0 && !rot3x3)) return pb_fail(PB_ERR_INVALID, "bad rotation arguments");
    PbParams Q;
    pb_params_of_request(Q, pb_to_end(dst), rot3x3, n_rot, pb_to_end(src));
    return pb_same_request(Q, plan->P) ? 1 : 0;
}

int pb_plan_create(const pb_proj* dst, const double* rot3x3, int n_rot, const pb_proj* src, pb_plan** out) {
    return pb_plan_create_ex(dst, rot3x3, n_rot, src, 0u, 0, out);
}

int pb_plan_prepare(pb_plan* plan, unsigned flags, int win_budget) {
    if (!plan) return pb_fail(PB_ERR_INVALID, "null argument");
    if (flags & ~(unsigned)(PB_PLAN_TUNE | PB_PLAN_BILINEAR)) return pb_fail(PB_ERR_INVALID, "unknown plan flags");
    if (win_budget < 0) return pb_fail(PB_ERR_INVALID, "negative window budget");
    if (flags & PB_PLAN_BILINEAR) plan->bil_wanted = 1;  // (a deferred plan: its preparation now includes the mode's tables)
    if (plan->device >= 0) {
        int dev = -1;
        PB_HIP(hipGetDevice(&dev));
        if (dev != plan->device) return pb_fail(PB_ERR_INVALID, "plan was prepared on another device; create one plan per device");
    }
    return pb_plan_prepare_full(plan, flags, win_budget);
}

int pb_plan_set_window_budget(pb_plan* plan, int win_budget) {
    if (!plan || win_budget <= 0) return pb_fail(PB_ERR_INVALID, "null plan or non-positive budget");
    if (!(plan->fast_ready || plan->dbl_ready)) return pb_fail(PB_ERR_UNSUPPORTED, "the plan has no tile tables (not prepared, or a geometry without a fast path)");
    int dev = -1;
    PB_HIP(hipGetDevice(&dev));
    if (dev != plan->device) return pb_fail(PB_ERR_INVALID, "plan was prepared on another device; create one plan per device");
    return pb_apply_budget(plan, win_budget);
}

int pb_plan_timing(const pb_plan* plan, double* prepare_ms, double* tune_ms) {
    if (!plan) return pb_fail(PB_ERR_INVALID, "null argument");
    if (prepare_ms) *prepare_ms = plan->prepare_ms;
    if (tune_ms) *tune_ms = plan->tune_ms;
    return PB_OK;
}

void pb_plan_destroy(pb_plan* plan) {
    if (!plan) return;
    // ONE wait (launches of the plan may still be in flight on the caller's streams), then every table back to the cache
    if (plan->table || plan->table_r || plan->ltable || plan->P_dev || plan->sep_rows) pb_wait_device(plan->device);
    pb_tmp_free(plan->table);
    pb_tmp_free(plan->fail_tiles);
    pb_tmp_free(plan->fix_px);
    pb_tmp_free(plan->idx_tab);
    pb_tmp_free(plan->fix_idx);
    pb_sep_release(plan);
    pb_tmp_free(plan->sep_rows);
    pb_tmp_free(plan->sep_cols);
    pb_tmp_free(plan->table_r);
    pb_tmp_free(plan->lat_tab);
    pb_tmp_free(plan->dbl_tile_fix);
    pb_tmp_free(plan->dbl_px_fix);
    pb_tmp_free(plan->saved_l);
    pb_tmp_free(plan->saved_r);
    pb_tmp_free(plan->ltable);
    pb_tmp_free(plan->ltable_bil);
    pb_tmp_free(plan->bil_dbl_tables);
    pb_tmp_free(plan->P_dev);
    pb_tmp_free(plan->bil_tiles);
    pb_tmp_free(plan->bil_xy);
    pb_tmp_free(plan->bil_fix_xy);
    delete plan;
}

int pb_plan_dst_shape(const pb_plan* plan, int* height, int* width) {
    if (!plan || !height || !width) return pb_fail(PB_ERR_INVALID, "null argument");
    *height = plan->P.dst.height;
    *width = plan->P.dst.width;
    return PB_OK;
}
int pb_plan_bilinear_float64_tiles(const pb_plan* plan) {
    if (!plan || !(plan->fast_ready || plan->dbl_ready)) return 0;
    if (!plan->bil_tiles) return (int)plan->n_tiles;  // (created with PB_PLAN_NO_BILINEAR and not asked since: pb_remap_bilinear_u8 runs the per-pixel float64 kernels)
    if (plan->bil_xy) return 0;  // every tile the models cannot serve has its exact coordinates in the plan's table
    return (int)(plan->n_fail_tiles + (plan->bil_tiles ? plan->n_bil_tiles : 0u));
}
int pb_plan_bilinear_tile_mix(const pb_plan* plan, long long mix[8]) {
    if (!plan || !mix) return pb_fail(PB_ERR_INVALID, "null argument");
    for (int k = 0; k < 8; ++k) mix[k] = 0;
    if (!(plan->fast_ready || plan->dbl_ready) || !plan->ltable_bil || plan->launch_groups_bil == 0) return PB_OK;
    if (plan->device >= 0) {
        int dev = -1;
        PB_HIP(hipGetDevice(&dev));
        if (dev != plan->device) return pb_fail(PB_ERR_INVALID, "plan was prepared on another device");
    }
    // the entries the bilinear launch's waves read, classified as the kernel takes them (pb_bilinear_mix_kernel)
    unsigned* counters = nullptr;
    PB_HIP(pb_tmp_alloc((void**)&counters, 8 * sizeof(unsigned)));
    hipError_t e = hipMemsetAsync(counters, 0, 8 * sizeof(unsigned), 0);
    if (e == hipSuccess) {
        const unsigned n_slots = 4u * plan->launch_groups_bil;
        hipLaunchKernelGGL(pb_bilinear_mix_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, 0, plan->ltable_bil, n_slots, counters);
        unsigned res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpy(res, counters, sizeof(res), hipMemcpyDeviceToHost);
        for (int k = 0; k < 8; ++k) mix[k] = res[k];
    }
    if (e != hipSuccess) (void)hipDeviceSynchronize();
    pb_tmp_free(counters);
    PB_HIP(e);
    return PB_OK;
}
int pb_plan_bilinear_launch_shape(const pb_plan* plan, int* lds_bytes, int* workgroups_per_frame) {
    if (!plan || !lds_bytes || !workgroups_per_frame) return pb_fail(PB_ERR_INVALID, "null argument");
    const bool tiles = (plan->fast_ready || plan->dbl_ready) && plan->ltable_bil && plan->launch_groups_bil > 0;
    *lds_bytes = tiles ? (int)plan->bil_pool_bytes : 0;
    *workgroups_per_frame = tiles ? (int)(plan->launch_groups_bil * (4u / (unsigned)plan->bil_waves)) : 0;
    return PB_OK;
}
int pb_plan_window_budget(const pb_plan* plan) {
    return (plan && (plan->fast_ready || plan->dbl_ready)) ? plan->P.win_budget : 0;
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
    if (plan->device >= 0) {  // the plan's tables live on the device it was created on
        int dev = -1;
        PB_HIP(hipGetDevice(&dev));
        if (dev != plan->device) return pb_fail(PB_ERR_INVALID, "plan was prepared on another device; create one plan per device");
    }
    const PbParams& P = plan->P;
    const unsigned long long npx = (unsigned long long)P.dst.height * P.dst.width;
    if (!src_frame_stride) src_frame_stride = 3ull * P.src.height * P.src.width;
    if (!dst_frame_stride) dst_frame_stride = 3ull * npx;
    if (dst_frame_stride < 3ull * npx) return pb_fail(PB_ERR_INVALID, "dst_frame_stride smaller than a frame");
    if (n_frames > 1 && src_frame_stride < 3ull * P.src.height * P.src.width)
        return pb_fail(PB_ERR_INVALID, "src_frame_stride smaller than a frame");
    return pb_remap_launch(plan, src_dev, dst_dev, n_frames, src_frame_stride, dst_frame_stride, (hipStream_t)stream);
}
}  // extern "C"

static int pb_remap_launch(const pb_plan* plan, const uint8_t* src_dev, uint8_t* dst_dev, int n_frames, size_t src_frame_stride,
                           size_t dst_frame_stride, hipStream_t st) {
    const PbParams& P = plan->P;
    const unsigned long long npx = (unsigned long long)P.dst.height * P.dst.width;
    if (!src_frame_stride) src_frame_stride = 3ull * P.src.height * P.src.width;
    if (!dst_frame_stride) dst_frame_stride = 3ull * npx;
    const bool windowable = ((((uintptr_t)src_dev) | src_frame_stride) & 15u) == 0;  // LDS-DMA row segments
    if (plan->dbl_ready && plan->ltable && plan->mode != PB_MODE_FAITHFUL && plan->mode != PB_MODE_FAST_DIRECT && windowable) {
        const dim3 block(64 * PB_TILE_WAVES);
        // one launch per frame: failed tiles and fix pixels go through the plan's stored faithful taps
        const PbSepRow* rows = plan->sep_ready ? plan->sep_rows : nullptr;
        static const int fpw_env = pb_knob("PB_DOUBLE_FPW", 0);
        const int fpw = fpw_env > 0 ? fpw_env : PB_DOUBLE_FRAMES_PER_WAVE;
        const unsigned gpf = plan->launch_groups;  // workgroups per frame of the plan's launch-order table
        const unsigned chunks = (unsigned)((n_frames + fpw - 1) / fpw);
        const dim3 bgrid(gpf * chunks);  // (h*w < 2^29 and n_frames an int: far below the grid limit for any batch that fits memory)
#define PB_LAUNCH_DOUBLE(WMODE, ONE)                                                                                              \
    hipLaunchKernelGGL((pb_hot_double_kernel<WMODE, ONE>), bgrid, block, pb_window_lds_bytes(P, 8) + (size_t)pb_knob("PB_LDS_PAD", 0), st, P, plan->table, plan->table_r, plan->ltable, rows, plan->lat_tab, \
                       plan->fix_px, plan->dbl_px_fix, plan->dbl_tile_fix, src_dev, dst_dev, n_frames, src_frame_stride,           \
                       dst_frame_stride, gpf, fpw, PbNoFrameTab{0})
        const bool one = n_frames == 1 || fpw == 1;
        if (rows && one) PB_LAUNCH_DOUBLE(1, true);
        else if (rows) PB_LAUNCH_DOUBLE(1, false);
        else if (plan->n_lat_tiles && one) PB_LAUNCH_DOUBLE(2, true);
        else if (plan->n_lat_tiles) PB_LAUNCH_DOUBLE(2, false);
        else if (one) PB_LAUNCH_DOUBLE(0, true);
        else PB_LAUNCH_DOUBLE(0, false);
#undef PB_LAUNCH_DOUBLE
    } else if (plan->sep_ready && plan->mode != PB_MODE_FAITHFUL && pb_sep_usable(plan, st)) {
        hipLaunchKernelGGL(pb_sep_double_kernel, dim3(pb_hot_blocks(P)), dim3(64 * PB_TILE_WAVES), 0, st, P, plan->sep_rows, plan->sep_cols,
                           src_dev, dst_dev, n_frames, src_frame_stride, dst_frame_stride);
    } else if (pb_use_fast(plan)) {
        pb_launch_fast<0>(plan, src_dev, dst_dev, n_frames, src_frame_stride, dst_frame_stride, nullptr, st);
    } else if (P.src.kind == PB_KIND_PANO) {
        pb_launch_faithful_remap<PB_KIND_PANO>(P, src_dev, dst_dev, n_frames, src_frame_stride, dst_frame_stride, st);
    } else if (P.src.kind == PB_KIND_CAMERA) {
        pb_launch_faithful_remap<PB_KIND_CAMERA>(P, src_dev, dst_dev, n_frames, src_frame_stride, dst_frame_stride, st);
    } else {
        pb_launch_faithful_remap<PB_KIND_DOUBLE>(P, src_dev, dst_dev, n_frames, src_frame_stride, dst_frame_stride, st);
    }
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// pb_remap_u8v: can the plan's hot kernel take a table of frame pointers?  (The windowed single-source kernel or the windowed
// double-fisheye kernel, as pb_remap_launch would choose them for 16-byte aligned frames.)
static bool pb_vec_launchable(const pb_plan* plan) {
    const PbParams& P = plan->P;
    if (plan->mode == PB_MODE_FAITHFUL || plan->mode == PB_MODE_FAST_DIRECT || !plan->ltable || plan->launch_groups == 0) return false;
    if (plan->dbl_ready) return true;
    return pb_use_fast(plan) && plan->P_dev && P.src.width < 32768 && P.src.height < 32768;
}

extern "C" {

int pb_remap_u8v(const pb_plan* plan, const uint8_t* const* src_dev, uint8_t* const* dst_dev, int n_frames, void* stream) {
    if (!plan || (n_frames > 0 && (!src_dev || !dst_dev))) return pb_fail(PB_ERR_INVALID, "null argument");
    if (n_frames < 0) return pb_fail(PB_ERR_INVALID, "negative frame count");
    if (n_frames == 0) return PB_OK;
    if (plan->device >= 0) {
        int dev = -1;
        PB_HIP(hipGetDevice(&dev));
        if (dev != plan->device) return pb_fail(PB_ERR_INVALID, "plan was prepared on another device; create one plan per device");
    }
    bool aligned = true;
    for (int f = 0; f < n_frames; ++f) {
        if (!src_dev[f] || !dst_dev[f]) return pb_fail(PB_ERR_INVALID, "null frame pointer");
        aligned = aligned && (((uintptr_t)src_dev[f]) & 15u) == 0;
    }
    hipStream_t st = (hipStream_t)stream;
    if (!aligned || !pb_vec_launchable(plan)) {
        // frames LDS-DMA cannot address, deferred plans, the float64 mode: the frames one by one (same bytes, n launches)
        for (int f = 0; f < n_frames; ++f) {
            const int rc = pb_remap_launch(plan, src_dev[f], dst_dev[f], 1, 0, 0, st);
            if (rc != PB_OK) return rc;
        }
        return PB_OK;
    }
    const PbParams& P = plan->P;
    const unsigned gpf = plan->launch_groups;
    const dim3 block(64 * PB_TILE_WAVES);
    for (int f0 = 0; f0 < n_frames; f0 += PB_MAX_VFRAMES) {
        const int nf = n_frames - f0 < PB_MAX_VFRAMES ? n_frames - f0 : PB_MAX_VFRAMES;
        PbFrameTab tab;
        for (int f = 0; f < PB_MAX_VFRAMES; ++f) {
            tab.src[f] = src_dev[f0 + (f < nf ? f : 0)];
            tab.dst[f] = dst_dev[f0 + (f < nf ? f : 0)];
        }
        const dim3 grid(gpf * (unsigned)nf);
        if (plan->dbl_ready) {
            const PbSepRow* rows = plan->sep_ready ? plan->sep_rows : nullptr;
#define PB_LAUNCH_DOUBLE_V(WMODE)                                                                                                          \
    hipLaunchKernelGGL((pb_hot_double_kernel<WMODE, true, true>), grid, block, pb_window_lds_bytes(P, 8), st, P, plan->table, plan->table_r, plan->ltable, \
                       rows, plan->lat_tab, plan->fix_px, plan->dbl_px_fix, plan->dbl_tile_fix, tab.src[0], tab.dst[0], nf, 0ull, 0ull, gpf, 1, tab)
            if (rows) PB_LAUNCH_DOUBLE_V(1);
            else if (plan->n_lat_tiles) PB_LAUNCH_DOUBLE_V(2);
            else PB_LAUNCH_DOUBLE_V(0);
#undef PB_LAUNCH_DOUBLE_V
        } else {
#define PB_LAUNCH_WIN_V(KIND)                                                                                                              \
    hipLaunchKernelGGL((pb_hot_win_kernel<KIND, true>), grid, block, pb_window_lds_bytes(P), st, (const PbParams*)plan->P_dev, pb_hot_of_host(P), plan->ltable, \
                       tab.src[0], tab.dst[0], gpf, 0ull, 0ull, plan->idx_tab, plan->fix_px, plan->fix_idx, (unsigned)nf, tab)
            if (P.src.kind == PB_KIND_PANO) PB_LAUNCH_WIN_V(PB_KIND_PANO);
            else PB_LAUNCH_WIN_V(PB_KIND_CAMERA);
#undef PB_LAUNCH_WIN_V
        }
    }
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_remap_bilinear_u8(const pb_plan* plan, const uint8_t* src_dev, uint8_t* dst_dev, int n_frames, size_t src_frame_stride,
                         size_t dst_frame_stride, void* stream) {
    if (!plan || !src_dev || !dst_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (n_frames < 0) return pb_fail(PB_ERR_INVALID, "negative frame count");
    if (n_frames == 0) return PB_OK;
    const PbParams& P = plan->P;
    if (plan->device >= 0) {
        int dev = -1;
        PB_HIP(hipGetDevice(&dev));
        if (dev != plan->device) return pb_fail(PB_ERR_INVALID, "plan was prepared on another device; create one plan per device");
    }
    const unsigned long long npx = (unsigned long long)P.dst.height * P.dst.width;
    if (!src_frame_stride) src_frame_stride = 3ull * P.src.height * P.src.width;
    if (!dst_frame_stride) dst_frame_stride = 3ull * npx;
    if (dst_frame_stride < 3ull * npx) return pb_fail(PB_ERR_INVALID, "dst_frame_stride smaller than a frame");
    if (n_frames > 1 && src_frame_stride < 3ull * P.src.height * P.src.width)
        return pb_fail(PB_ERR_INVALID, "src_frame_stride smaller than a frame");
    hipStream_t st = (hipStream_t)stream;
    if (P.src.kind == PB_KIND_DOUBLE) {
        if (plan->dbl_ready && plan->ltable_bil && plan->launch_groups_bil > 0 && plan->bil_tiles && plan->bil_dbl_tables && plan->mode != PB_MODE_FAITHFUL) {
            // the per-eye tile models of the nearest mode's plan + the exact coordinate tables: one wave per tile, ONE launch
            const unsigned gpf = plan->launch_groups_bil * (4u / (unsigned)plan->bil_waves);  // real workgroups per frame
            PbHot Hb = pb_hot_of_host(P);  // (the bilinear mode's window budget travels in it)
            Hb.win_budget = plan->bil_budget;
            const int windows = plan->mode != PB_MODE_FAST_DIRECT && ((((uintptr_t)src_dev) | src_frame_stride) & 15u) == 0;
            const PbSepRow* rows = plan->sep_ready ? plan->sep_rows : nullptr;
            const int per_launch = (int)(0x7FFFFFFFu / gpf);
            for (int f0 = 0; f0 < n_frames; f0 += per_launch) {
                const int nf = n_frames - f0 < per_launch ? n_frames - f0 : per_launch;
                const dim3 grid(gpf * (unsigned)nf), block(64 * plan->bil_waves);
                const uint8_t* sf = src_dev + (unsigned long long)f0 * src_frame_stride;
                uint8_t* df = dst_dev + (unsigned long long)f0 * dst_frame_stride;
#define PB_LAUNCH_BILINEAR_DOUBLE(WMODE)                                                                                                       \
    if (plan->bil_waves == 2)                                                                                                                  \
        hipLaunchKernelGGL((pb_bilinear_double_hot_kernel<WMODE, 2>), grid, block, (size_t)plan->bil_pool_bytes, st, Hb, P.src_eye_w, plan->ltable_bil, plan->bil_dbl_tables, \
                           sf, df, gpf, (unsigned long long)src_frame_stride, (unsigned long long)dst_frame_stride, windows, plan->bil_xy);    \
    else                                                                                                                                       \
    hipLaunchKernelGGL((pb_bilinear_double_hot_kernel<WMODE, 4>), grid, block, (size_t)plan->bil_pool_bytes, st, Hb, P.src_eye_w, plan->ltable_bil, plan->bil_dbl_tables, \
                       sf, df, gpf, (unsigned long long)src_frame_stride, (unsigned long long)dst_frame_stride, windows, plan->bil_xy)
                if (rows) PB_LAUNCH_BILINEAR_DOUBLE(1);
                else if (plan->n_lat_tiles) PB_LAUNCH_BILINEAR_DOUBLE(2);
                else PB_LAUNCH_BILINEAR_DOUBLE(0);
#undef PB_LAUNCH_BILINEAR_DOUBLE
            }
            if (!plan->bil_xy) {  // no coordinate tables (they would not fit): the failed / listed tiles and the fix pixels from the float64 chain
                const unsigned n_tiles64 = plan->n_fail_tiles + plan->n_bil_tiles;
                const unsigned fix_blocks = 4u * n_tiles64 + (plan->n_fix_px + PB_BLOCK - 1) / PB_BLOCK;
                if (fix_blocks)
                    hipLaunchKernelGGL(pb_bilinear_double_fix_kernel, dim3(fix_blocks), dim3(PB_BLOCK), 0, st, P, plan->fail_tiles, (int)plan->n_fail_tiles, plan->bil_tiles,
                                       (int)n_tiles64, plan->fix_px, (int)plan->n_fix_px, src_dev, dst_dev, n_frames, src_frame_stride, dst_frame_stride);
            }
        } else {
            hipLaunchKernelGGL(pb_bilinear_double_kernel, dim3(pb_blocks(npx)), dim3(PB_BLOCK), 0, st, P, src_dev, dst_dev, n_frames, src_frame_stride,
                               dst_frame_stride);
        }
        PB_HIP(hipGetLastError());
        return PB_OK;
    }
    if (pb_use_fast(plan) && plan->ltable_bil && plan->launch_groups_bil > 0 && plan->bil_tiles) {
        // launched like the nearest hot kernel: the mode's launch-order table, frames of a batch as a grid dimension, ONE launch;
        // LEAN tiles take their taps from LDS windows, except for frames LDS-DMA cannot address (not 16-byte aligned)
        const unsigned gpf = plan->launch_groups_bil * (4u / (unsigned)plan->bil_waves);  // real workgroups per frame
        PbParams Pb = P;
        Pb.win_budget = plan->bil_budget;
        const dim3 block(64 * plan->bil_waves);
        const int windows = plan->mode != PB_MODE_FAST_DIRECT && P.src.width < 32768 && P.src.height < 32768 &&
                            ((((uintptr_t)src_dev) | src_frame_stride) & 15u) == 0;
        const int per_launch = (int)(0x7FFFFFFFu / gpf);
#define PB_LAUNCH_BILINEAR(KIND)                                                                                                     \
    do {                                                                                                                             \
        for (int f0 = 0; f0 < n_frames; f0 += per_launch) {                                                                          \
            const int nf = n_frames - f0 < per_launch ? n_frames - f0 : per_launch;                                                  \
            if (plan->bil_waves == 2)                                                                                                \
                hipLaunchKernelGGL((pb_bilinear_hot_kernel<KIND, 2>), dim3(gpf * (unsigned)nf), block, (size_t)plan->bil_pool_bytes, st, pb_hot_of_host(Pb), plan->ltable_bil, \
                                   src_dev + (unsigned long long)f0 * src_frame_stride, dst_dev + (unsigned long long)f0 * dst_frame_stride, gpf, \
                                   (unsigned long long)src_frame_stride, (unsigned long long)dst_frame_stride, windows, plan->bil_xy, plan->fix_px, \
                                   plan->bil_fix_xy);                                                                                \
            else                                                                                                                     \
            hipLaunchKernelGGL((pb_bilinear_hot_kernel<KIND, 4>), dim3(gpf * (unsigned)nf), block, (size_t)plan->bil_pool_bytes, st, pb_hot_of_host(Pb), plan->ltable_bil, \
                               src_dev + (unsigned long long)f0 * src_frame_stride, dst_dev + (unsigned long long)f0 * dst_frame_stride, gpf, \
                               (unsigned long long)src_frame_stride, (unsigned long long)dst_frame_stride, windows, plan->bil_xy, plan->fix_px, \
                               plan->bil_fix_xy);                                                                                    \
        }                                                                                                                            \
        const unsigned n64 = plan->n_fail_tiles + plan->n_bil_tiles; /* failed + listed tiles */                                    \
        if (!plan->bil_xy && (n64 || plan->n_fix_px)) /* no coordinate tables (they would not fit): the float64 chain */             \
            hipLaunchKernelGGL(pb_bilinear_fix_kernel<KIND>, dim3(4u * n64 + (plan->n_fix_px + PB_BLOCK - 1) / PB_BLOCK),            \
                               dim3(PB_BLOCK), 0, st, P, plan->fail_tiles, 0, src_dev, dst_dev, n_frames, src_frame_stride,          \
                               dst_frame_stride, (int)n64, plan->fix_px, (int)plan->n_fix_px, plan->bil_tiles,                       \
                               (int)plan->n_fail_tiles);                                                                             \
    } while (0)
        if (P.src.kind == PB_KIND_PANO) PB_LAUNCH_BILINEAR(PB_KIND_PANO);
        else PB_LAUNCH_BILINEAR(PB_KIND_CAMERA);
#undef PB_LAUNCH_BILINEAR
    } else {
        if (P.src.kind == PB_KIND_PANO)
            hipLaunchKernelGGL(pb_bilinear_fix_kernel<PB_KIND_PANO>, dim3(pb_blocks(npx)), dim3(PB_BLOCK), 0, st, P, nullptr, 1, src_dev,
                               dst_dev, n_frames, src_frame_stride, dst_frame_stride);
        else
            hipLaunchKernelGGL(pb_bilinear_fix_kernel<PB_KIND_CAMERA>, dim3(pb_blocks(npx)), dim3(PB_BLOCK), 0, st, P, nullptr, 1, src_dev,
                               dst_dev, n_frames, src_frame_stride, dst_frame_stride);
    }
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_index_map_i32(const pb_plan* plan, int32_t* idx_dev, double* weights_dev, void* stream) {
    if (!plan || !idx_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    const PbParams& P = plan->P;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = pb_blocks((unsigned long long)P.dst.height * P.dst.width);
    if (pb_use_fast(plan)) {
        pb_launch_fast<1>(plan, nullptr, nullptr, 0, 0, 0, idx_dev, st);
    } else if (P.src.kind == PB_KIND_PANO) {
        PB_LAUNCH_BY_ROT(P.n_rot, pb_index_kernel, PB_KIND_PANO, dim3(blocks), dim3(PB_BLOCK), 0, st, P, idx_dev, weights_dev);
    } else if (P.src.kind == PB_KIND_CAMERA) {
        PB_LAUNCH_BY_ROT(P.n_rot, pb_index_kernel, PB_KIND_CAMERA, dim3(blocks), dim3(PB_BLOCK), 0, st, P, idx_dev, weights_dev);
    } else {
        PB_LAUNCH_BY_ROT(P.n_rot, pb_index_kernel, PB_KIND_DOUBLE, dim3(blocks), dim3(PB_BLOCK), 0, st, P, idx_dev, weights_dev);
    }
    PB_HIP(hipGetLastError());
    return PB_OK;
}

#ifdef PB_ABLATION  // diagnostic build only (tests/test_hip_math.py): pb_math.hpp's correctly rounded functions as the gfx950 build evaluates them
}  // extern "C"
__global__ void pb_debug_math_kernel(int fn, const double* __restrict__ in, double* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (fn) {
        case 0: out[i] = pb_asin_svml(in[i]); break;
        case 1: out[i] = pb_acos_svml(in[i]); break;
        case 2: out[i] = pb_atan_svml(in[i]); break;
        case 3: out[i] = pb_tan_svml(in[i]); break;
        case 8: out[i] = pb_asin_libm(in[i]); break;
        case 9: out[i] = pb_acos_libm(in[i]); break;
        case 10: out[i] = pb_atan_libm(in[i]); break;
        case 11: out[i] = pb_tan_libm(in[i]); break;
        case 4: out[i] = pb_sin_np(in[i]); break;
        case 5: out[i] = pb_cos_np(in[i]); break;
        case 6: pb_expi_np(in[i], &out[2 * i], &out[2 * i + 1]); break;
        default: out[i] = pb_arg_np(in[2 * i], in[2 * i + 1]);
    }
}
extern "C" {
// fn (the order of tests/npmath_args.py FUNCTIONS): 0 arcsin, 1 arccos, 2 arctan, 3 tan, 4 sin, 5 cos of n values; 6: np.exp(x * 1j) -> (imag, real) interleaved; 7: n (y, x) pairs -> np.log(x + 1j y).imag;
// 8-11: arcsin, arccos, arctan, tan of the second math flavour (pb_math_libm.hpp)
__attribute__((visibility("default"))) int pb_debug_math(int fn, const double* in_dev, double* out_dev, size_t n, void* stream) {
    if (!in_dev || !out_dev || fn < 0 || fn > 11) return pb_fail(PB_ERR_INVALID, "bad argument");
    if (n) hipLaunchKernelGGL(pb_debug_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fn, in_dev, out_dev, n);
    PB_HIP(hipGetLastError());
    return PB_OK;
}
#endif

#ifdef PB_TRACE
__attribute__((visibility("default"))) int pb_debug_trace(unsigned long long* out, size_t n_words, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pb_trace), n_words * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(pb_trace)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * 65536 * 16) != hipSuccess) return -2;
    }
    return 0;
}
__attribute__((visibility("default"))) int pb_debug_trace_frame(const pb_plan* plan, int frame) {
    // record the waves of frame `frame` of a batch launch (4-wave workgroups); frame < 0: whole launches (single frames)
    const unsigned wpf = (plan && frame >= 0) ? plan->launch_groups : 0xFFFFFFFFu, f = frame >= 0 ? (unsigned)frame : 0u;
    if (hipMemcpyToSymbol(HIP_SYMBOL(pb_trace_wpf), &wpf, sizeof(wpf)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(pb_trace_frame), &f, sizeof(f)) == hipSuccess ? 0 : -1;
}
__attribute__((visibility("default"))) int pb_debug_copy_table_r(const pb_plan* plan, void* host, size_t bytes) {  // the right eye's table of a double-fisheye plan
    if (!plan || !plan->table_r) return -1;
    const size_t have = (size_t)plan->n_tiles * sizeof(PbTileEntry);
    return hipMemcpy(host, plan->table_r, bytes < have ? bytes : have, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
__attribute__((visibility("default"))) int pb_debug_copy_table(const pb_plan* plan, void* host, size_t bytes) {
    if (!plan || !plan->table) return -1;
    const size_t have = (size_t)plan->n_tiles * sizeof(PbTileEntry);
    return hipMemcpy(host, plan->table, bytes < have ? bytes : have, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

int pb_plan_set_mode(pb_plan* plan, int mode) {
    if (!plan) return pb_fail(PB_ERR_INVALID, "null argument");
    if (mode < PB_MODE_AUTO || mode > PB_MODE_FAST_DIRECT) return pb_fail(PB_ERR_INVALID, "mode out of range");
    plan->mode = mode;
    return PB_OK;
}

int pb_plan_info(const pb_plan* plan, int* fast_path_enabled, long long* stats7, long long* thresholds4) {
    long long* stats5 = stats7;
    if (!plan) return pb_fail(PB_ERR_INVALID, "null argument");
    if (fast_path_enabled)
        *fast_path_enabled = (pb_use_fast(plan) || ((plan->sep_ready || plan->dbl_ready) && plan->mode != PB_MODE_FAITHFUL)) ? 1 : 0;
    if (stats5) {
        const bool have = plan->fast_ready || plan->dbl_ready;  // a double source counts both eyes' tables
        stats5[0] = have ? (long long)plan->n_tiles : -1;
        stats5[1] = have ? (long long)plan->n_fail_tiles : -1;
        stats5[2] = have ? (long long)plan->n_fix_px : -1;
        stats5[3] = have ? plan->diff_pixels : -1;
        stats5[4] = have ? (long long)plan->n_lean_tiles : -1;
        stats5[5] = have ? (long long)plan->n_black_tiles : -1;
        stats5[6] = have ? (long long)plan->n_direct_tiles : -1;
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
    if (!pb_end_ok(dst, why, PB_ROLE_DST)) return pb_fail(PB_ERR_INVALID, why);
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
    if (!pb_end_ok(src, why, PB_ROLE_SRC)) return pb_fail(PB_ERR_INVALID, why);
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

int pb_index_from_map_i32(const pb_proj* src, double* map_dev, int height, int width, const double* dist_l_dev, const double* dist_r_dev,
                          int32_t* idx_dev, double* weights_dev, void* stream) {
    std::string why;
    if (!map_dev || !idx_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (!pb_end_ok(src, why, PB_ROLE_SRC | PB_ROLE_CUSTOM_OK)) return pb_fail(PB_ERR_INVALID, why);
    if (height < 1 || width < 1 || (long long)height * width > 0x7FFFFFFFll / 4) return pb_fail(PB_ERR_INVALID, "map size out of range");
    if (src->kind != PB_KIND_PANO && src->lens == PB_LENS_CUSTOM && !dist_l_dev)
        return pb_fail(PB_ERR_INVALID, "a PB_LENS_CUSTOM source needs the host-evaluated distance plane(s)");
    if (src->kind == PB_KIND_DOUBLE && dist_l_dev && !dist_r_dev) return pb_fail(PB_ERR_INVALID, "a double source needs both distance planes");
    if (src->kind == PB_KIND_PANO && (dist_l_dev || dist_r_dev)) return pb_fail(PB_ERR_INVALID, "a panorama source has no lens");
    PbParams P;
    memset(&P, 0, sizeof(P));
    P.src = pb_to_end(src);
    P.dst = P.src;
    P.dst.kind = PB_KIND_PANO;  // only the source half of the parameters is used
    P.dst.height = height;
    P.dst.width = width;
    pb_derive(P);
    const unsigned total = (unsigned)height * (unsigned)width;
    hipStream_t st = (hipStream_t)stream;
    switch (P.src.kind) {
        case PB_KIND_PANO:
            hipLaunchKernelGGL(pb_index_from_map_kernel<PB_KIND_PANO>, dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, st, P, map_dev, total, dist_l_dev,
                               dist_r_dev, idx_dev, weights_dev);
            break;
        case PB_KIND_CAMERA:
            hipLaunchKernelGGL(pb_index_from_map_kernel<PB_KIND_CAMERA>, dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, st, P, map_dev, total, dist_l_dev,
                               dist_r_dev, idx_dev, weights_dev);
            break;
        default:
            hipLaunchKernelGGL(pb_index_from_map_kernel<PB_KIND_DOUBLE>, dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, st, P, map_dev, total, dist_l_dev,
                               dist_r_dev, idx_dev, weights_dev);
    }
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_sample_map_bilinear_px(const pb_proj* src, double* map_dev, int height, int width, const double* dist_l_dev, const double* dist_r_dev,
                              const void* img_dev, void* out_dev, int channels, int sample_bytes, void* stream) {
    std::string why;
    if (!map_dev || !img_dev || !out_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (!pb_end_ok(src, why, PB_ROLE_SRC | PB_ROLE_CUSTOM_OK)) return pb_fail(PB_ERR_INVALID, why);
    if (height < 1 || width < 1 || (long long)height * width > 0x7FFFFFFFll / 4) return pb_fail(PB_ERR_INVALID, "map size out of range");
    if (channels < 1 || channels > 16) return pb_fail(PB_ERR_INVALID, "channels outside [1, 16]");
    if (sample_bytes != 1 && sample_bytes != 2) return pb_fail(PB_ERR_INVALID, "sample_bytes must be 1 or 2");
    if (src->kind != PB_KIND_PANO && src->lens == PB_LENS_CUSTOM && !dist_l_dev)
        return pb_fail(PB_ERR_INVALID, "a PB_LENS_CUSTOM source needs the host-evaluated distance plane(s)");
    if (src->kind == PB_KIND_DOUBLE && dist_l_dev && !dist_r_dev) return pb_fail(PB_ERR_INVALID, "a double source needs both distance planes");
    if (src->kind == PB_KIND_PANO && (dist_l_dev || dist_r_dev)) return pb_fail(PB_ERR_INVALID, "a panorama source has no lens");
    PbParams P;
    memset(&P, 0, sizeof(P));
    P.src = pb_to_end(src);
    P.dst = P.src;
    P.dst.kind = PB_KIND_PANO;  // only the source half of the parameters is used
    P.dst.height = height;
    P.dst.width = width;
    pb_derive(P);
    const unsigned total = (unsigned)height * (unsigned)width;
    hipStream_t st = (hipStream_t)stream;
#define PB_LAUNCH_MAP_BIL(KIND, SAMPLE)                                                                                                  \
    hipLaunchKernelGGL((pb_sample_map_bilinear_kernel<KIND, SAMPLE>), dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, st, P, map_dev, total, dist_l_dev, \
                       dist_r_dev, static_cast<const SAMPLE*>(img_dev), out_dev, channels)
    if (sample_bytes == 1) {
        if (P.src.kind == PB_KIND_PANO) PB_LAUNCH_MAP_BIL(PB_KIND_PANO, uint8_t);
        else if (P.src.kind == PB_KIND_CAMERA) PB_LAUNCH_MAP_BIL(PB_KIND_CAMERA, uint8_t);
        else PB_LAUNCH_MAP_BIL(PB_KIND_DOUBLE, uint8_t);
    } else {
        if (P.src.kind == PB_KIND_PANO) PB_LAUNCH_MAP_BIL(PB_KIND_PANO, uint16_t);
        else if (P.src.kind == PB_KIND_CAMERA) PB_LAUNCH_MAP_BIL(PB_KIND_CAMERA, uint16_t);
        else PB_LAUNCH_MAP_BIL(PB_KIND_DOUBLE, uint16_t);
    }
#undef PB_LAUNCH_MAP_BIL
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_sample_map_bilinear_u8(const pb_proj* src, double* map_dev, int height, int width, const uint8_t* src_dev, uint8_t* dst_dev, void* stream) {
    return pb_sample_map_bilinear_px(src, map_dev, height, width, nullptr, nullptr, src_dev, dst_dev, 3, 1, stream);
}

int pb_gather_px(const int32_t* idx_dev, const void* src_dev, void* dst_dev, size_t n_px, int bytes_per_px, void* stream) {
    if (!idx_dev || !src_dev || !dst_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (bytes_per_px < 1 || bytes_per_px > 64) return pb_fail(PB_ERR_INVALID, "bytes_per_px outside [1, 64]");
    if (n_px > 0x7FFFFFFFull) return pb_fail(PB_ERR_INVALID, "too many pixels");
    if (!n_px) return PB_OK;
    hipLaunchKernelGGL(pb_gather_px_kernel, dim3(pb_blocks(n_px)), dim3(PB_BLOCK), 0, (hipStream_t)stream, idx_dev,
                       static_cast<const uint8_t*>(src_dev), static_cast<uint8_t*>(dst_dev), (unsigned long long)n_px, bytes_per_px);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_gather_blend_u8(const int32_t* idx2_dev, const double* weights2_dev, const void* src_dev, uint8_t* dst_dev, size_t n_px, int channels,
                       int sample_bytes, void* stream) {
    if (!idx2_dev || !weights2_dev || !src_dev || !dst_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (channels < 1 || channels > 16) return pb_fail(PB_ERR_INVALID, "channels outside [1, 16]");
    if (sample_bytes != 1 && sample_bytes != 2) return pb_fail(PB_ERR_UNSUPPORTED, "the double-fisheye blend takes 8- or 16-bit unsigned samples");
    if (n_px > 0x7FFFFFFFull) return pb_fail(PB_ERR_INVALID, "too many pixels");
    if (!n_px) return PB_OK;
    if (sample_bytes == 1)
        hipLaunchKernelGGL(pb_gather_blend_kernel<uint8_t>, dim3(pb_blocks(n_px)), dim3(PB_BLOCK), 0, (hipStream_t)stream, idx2_dev, weights2_dev,
                           static_cast<const uint8_t*>(src_dev), dst_dev, (unsigned long long)n_px, channels);
    else
        hipLaunchKernelGGL(pb_gather_blend_kernel<uint16_t>, dim3(pb_blocks(n_px)), dim3(PB_BLOCK), 0, (hipStream_t)stream, idx2_dev, weights2_dev,
                           static_cast<const uint16_t*>(src_dev), dst_dev, (unsigned long long)n_px, channels);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

int pb_map_projection_u8(double* map_dev, int height, int width, uint8_t* out_dev, void* workspace24_dev, void* stream) {
    if (!map_dev || !out_dev || !workspace24_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if (height < 1 || width < 1 || (long long)height * width > 0x7FFFFFFFll / 4)
        return pb_fail(PB_ERR_INVALID, "map size out of range");
    const unsigned total = (unsigned)height * (unsigned)width;
    hipStream_t st = (hipStream_t)stream;
    static const unsigned long long init[3] = {~0ull, 0ull, 0ull};
    PB_HIP(hipMemcpyAsync(workspace24_dev, init, sizeof(init), hipMemcpyHostToDevice, st));
    unsigned long long* ws = reinterpret_cast<unsigned long long*>(workspace24_dev);
    hipLaunchKernelGGL(pb_mapproj_minmax_kernel, dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, st, map_dev, total, ws);
    hipLaunchKernelGGL(pb_mapproj_colour_kernel, dim3(pb_blocks(total)), dim3(PB_BLOCK), 0, st, map_dev, total, ws, out_dev);
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

// ---- plan persistence ------------------------------------------------------------------
// A prepared plan as one host blob: header, PbParams, then the device tables in a fixed order.  The blob is a
// cache of a computation (like a compiled kernel), valid for this library build (ABI version, struct sizes) and
// integrity-checked (FNV-1a over the payload); it is not a hardened file format.
}  // extern "C"

namespace {
struct PbBlobHeader {
    uint32_t magic, version, params_size, entry_size;
    int32_t fast_ready, sep_ready, dbl_ready, reserved;
    uint32_t n_tiles, n_fail_tiles, n_fix_px, n_lean_tiles, n_black_tiles, n_direct_tiles, n_row_weight_tiles, n_lat_tiles;
    int64_t diff_pixels;
    uint64_t section_bytes[13];
    uint64_t checksum;  // of everything after the header
};
const uint32_t PB_BLOB_MAGIC = 0x4C504250u;  // "PBPL"
const uint32_t PB_BLOB_VERSION = 5;  // 4: tile flags carry PB_TILE_COARSE; 5: ... and the walk bits of the bilinear table slots; the header names the math flavour

struct PbSection {
    void** ptr;
    size_t bytes;
};
// the plan's device tables and their sizes (0 = absent), in blob order
void pb_plan_sections(pb_plan* pl, PbSection sec[13]) {
    const size_t nt = pl->n_tiles, nf = pl->n_fail_tiles ? pl->n_fail_tiles : 1, np = pl->n_fix_px ? pl->n_fix_px : 1;
    const bool tiles = pl->fast_ready || pl->dbl_ready;
    sec[0] = {(void**)&pl->table, tiles ? nt * sizeof(PbTileEntry) : 0};
    sec[1] = {(void**)&pl->table_r, pl->dbl_ready ? nt * sizeof(PbTileEntry) : 0};
    sec[2] = {(void**)&pl->fail_tiles, tiles ? nf * sizeof(int32_t) : 0};
    sec[3] = {(void**)&pl->fix_px, tiles ? np * sizeof(int32_t) : 0};
    sec[4] = {(void**)&pl->idx_tab, pl->fast_ready ? nf * PB_TILE * PB_TILE * sizeof(int32_t) : 0};
    sec[5] = {(void**)&pl->fix_idx, pl->fast_ready ? np * sizeof(int32_t) : 0};
    sec[6] = {(void**)&pl->sep_rows, pl->sep_rows ? (size_t)pl->P.dst.height * sizeof(PbSepRow) : 0};
    sec[7] = {(void**)&pl->sep_cols, pl->sep_cols ? (size_t)pl->P.dst.width * sizeof(PbSepCol) : 0};
    sec[8] = {(void**)&pl->dbl_tile_fix, pl->dbl_ready ? nf * PB_TILE * PB_TILE * sizeof(PbDoubleFix) : 0};
    sec[9] = {(void**)&pl->dbl_px_fix, pl->dbl_ready ? np * sizeof(PbDoubleFix) : 0};
    sec[10] = {(void**)&pl->lat_tab, (size_t)pl->n_lat_tiles * PB_LAT_TILE_DOUBLES * sizeof(double)};
    sec[11] = {(void**)&pl->saved_l, tiles ? nt * sizeof(int32_t) : 0};
    sec[12] = {(void**)&pl->saved_r, pl->dbl_ready ? nt * sizeof(int32_t) : 0};
}
uint64_t pb_fnv1a(const uint8_t* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}
}  // namespace

extern "C" {

int pb_plan_serialize(const pb_plan* plan_c, void* buf, size_t capacity, size_t* size_out) {
    if (!plan_c || !size_out) return pb_fail(PB_ERR_INVALID, "null argument");
    pb_plan* pl = const_cast<pb_plan*>(plan_c);  // sections are described through member addresses; nothing is modified
    if (!(pl->fast_ready || pl->dbl_ready || pl->sep_ready)) return pb_fail(PB_ERR_UNSUPPORTED, "the plan holds no device tables (not prepared)");
    PbSection sec[13];
    pb_plan_sections(pl, sec);
    size_t total = sizeof(PbBlobHeader) + sizeof(PbParams);
    for (int i = 0; i < 13; ++i) total += sec[i].bytes;
    *size_out = total;
    if (!buf) return PB_OK;  // size query
    if (capacity < total) return pb_fail(PB_ERR_INVALID, "buffer too small for the serialized plan");
    int dev = -1;
    PB_HIP(hipGetDevice(&dev));
    if (dev != pl->device) return pb_fail(PB_ERR_INVALID, "plan was prepared on another device; create one plan per device");
    uint8_t* out = static_cast<uint8_t*>(buf);
    PbBlobHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = PB_BLOB_MAGIC;
    h.version = PB_BLOB_VERSION;
    h.params_size = (uint32_t)sizeof(PbParams);
    h.entry_size = (uint32_t)sizeof(PbTileEntry);
    h.fast_ready = pl->fast_ready; h.sep_ready = pl->sep_ready; h.dbl_ready = pl->dbl_ready; h.reserved = pl->walk | (PB_MATH_FLAVOUR << 8);
    h.n_tiles = pl->n_tiles; h.n_fail_tiles = pl->n_fail_tiles; h.n_fix_px = pl->n_fix_px; h.n_lean_tiles = pl->n_lean_tiles;
    h.n_black_tiles = pl->n_black_tiles; h.n_direct_tiles = pl->n_direct_tiles; h.n_row_weight_tiles = pl->n_row_weight_tiles;
    h.n_lat_tiles = pl->n_lat_tiles; h.diff_pixels = pl->diff_pixels;
    uint8_t* q = out + sizeof(PbBlobHeader);
    memcpy(q, &pl->P, sizeof(PbParams));
    q += sizeof(PbParams);
    PB_HIP(hipDeviceSynchronize());
    for (int i = 0; i < 13; ++i) {
        h.section_bytes[i] = sec[i].bytes;
        if (!sec[i].bytes) continue;
        if (!*sec[i].ptr) return pb_fail(PB_ERR_INVALID, "inconsistent plan: a table is missing");
        PB_HIP(hipMemcpy(q, *sec[i].ptr, sec[i].bytes, hipMemcpyDeviceToHost));
        q += sec[i].bytes;
    }
    h.checksum = pb_fnv1a(out + sizeof(PbBlobHeader), total - sizeof(PbBlobHeader));
    memcpy(out, &h, sizeof(h));
    return PB_OK;
}

int pb_plan_deserialize(const void* buf, size_t size, pb_plan** out) {
    if (!buf || !out) return pb_fail(PB_ERR_INVALID, "null argument");
    *out = nullptr;
    if (size < sizeof(PbBlobHeader) + sizeof(PbParams)) return pb_fail(PB_ERR_INVALID, "serialized plan truncated");
    const uint8_t* in = static_cast<const uint8_t*>(buf);
    PbBlobHeader h;
    memcpy(&h, in, sizeof(h));
    if (h.magic != PB_BLOB_MAGIC || h.version != PB_BLOB_VERSION || h.params_size != sizeof(PbParams) || h.entry_size != sizeof(PbTileEntry))
        return pb_fail(PB_ERR_INVALID, "serialized plan is not from this library build (magic / version / struct sizes)");
    if (h.checksum != pb_fnv1a(in + sizeof(PbBlobHeader), size - sizeof(PbBlobHeader)))
        return pb_fail(PB_ERR_INVALID, "serialized plan is corrupt (checksum)");
    // the blob's exact tables, thresholds and certified flags are products of ONE math flavour's asin / acos / atan / tan
    if (((h.reserved >> 8) & 0xFF) != PB_MATH_FLAVOUR)
        return pb_fail(PB_ERR_UNSUPPORTED, PB_MATH_FLAVOUR ? "serialized plan was made by libphotonbend_hip.so (the AVX-512 / SVML math flavour); this is libphotonbend_hip_libm.so"
                                                            : "serialized plan was made by libphotonbend_hip_libm.so (glibc's asin / acos / atan / tan); this is libphotonbend_hip.so");
    pb_plan* pl = new (std::nothrow) pb_plan();
    if (!pl) return pb_fail(PB_ERR_INVALID, "out of host memory");
    memcpy(&pl->P, in + sizeof(PbBlobHeader), sizeof(PbParams));
    pl->fast_ready = h.fast_ready; pl->sep_ready = h.sep_ready; pl->dbl_ready = h.dbl_ready;
    pb_sep_setup(pl);
    pl->walk = ((h.reserved & 0xFF) <= 3) ? (h.reserved & 0xFF) : 0;
    pl->n_tiles = h.n_tiles; pl->n_fail_tiles = h.n_fail_tiles; pl->n_fix_px = h.n_fix_px; pl->n_lean_tiles = h.n_lean_tiles;
    pl->n_black_tiles = h.n_black_tiles; pl->n_direct_tiles = h.n_direct_tiles; pl->n_row_weight_tiles = h.n_row_weight_tiles;
    pl->n_lat_tiles = h.n_lat_tiles; pl->diff_pixels = h.diff_pixels;
    pl->mode = PB_MODE_AUTO;
    const bool tiles = pl->fast_ready || pl->dbl_ready;
    // the derived half of the parameter block must be what THIS build derives from the blob's own request (a blob whose tables
    // were made for other constants is foreign, whatever its checksum says)
    bool ok = pl->P.n_rot >= 0 && pl->P.n_rot <= PB_MAX_ROTATIONS;
    if (ok) {
        PbParams Q;
        pb_params_of_request(Q, pl->P.dst, &pl->P.R[0][0], pl->P.n_rot, pl->P.src);
        ok = pb_same_request(Q, pl->P);
    }
    ok = ok && (!tiles || pl->n_tiles == pb_num_tiles(pl->P)) && pl->n_fail_tiles <= 2u * pl->n_tiles + 1u &&
              pl->P.win_budget >= PB_DIRECT_LDS_BYTES && pl->P.win_budget <= PB_WINLDS_MAX && (pl->P.win_budget & 15) == 0 &&
              pl->P.n_rot >= 0 && pl->P.n_rot <= PB_MAX_ROTATIONS;
    // section presence is decided by pointers on the writing side: reproduce it from the recorded sizes
    PbSection sec[13];
    static uint8_t present_tag;  // any non-null value
    if (h.section_bytes[6]) pl->sep_rows = reinterpret_cast<PbSepRow*>(&present_tag);
    if (h.section_bytes[7]) pl->sep_cols = reinterpret_cast<PbSepCol*>(&present_tag);
    pb_plan_sections(pl, sec);
    pl->sep_rows = nullptr;
    pl->sep_cols = nullptr;
    size_t total = sizeof(PbBlobHeader) + sizeof(PbParams);
    for (int i = 0; i < 13; ++i) {
        ok = ok && h.section_bytes[i] == sec[i].bytes;
        total += h.section_bytes[i];
    }
    if (!ok || total != size) {
        delete pl;
        return pb_fail(PB_ERR_INVALID, "serialized plan is inconsistent (sizes do not match its geometry)");
    }
    int rc = PB_OK;
    if (hipGetDevice(&pl->device) != hipSuccess) rc = PB_ERR_HIP;
    const uint8_t* q = in + sizeof(PbBlobHeader) + sizeof(PbParams);
    for (int i = 0; i < 13 && rc == PB_OK; ++i) {
        if (!sec[i].bytes) continue;
        if (pb_tmp_alloc(sec[i].ptr, sec[i].bytes) != hipSuccess || hipMemcpy(*sec[i].ptr, q, sec[i].bytes, hipMemcpyHostToDevice) != hipSuccess)
            rc = PB_ERR_HIP;
        q += sec[i].bytes;
    }
    if (rc != PB_OK) {
        g_err = std::string("plan upload failed: ") + hipGetErrorString(hipGetLastError());
        pb_plan_destroy(pl);
        return rc;
    }
    if (pb_tmp_alloc((void**)&pl->P_dev, sizeof(PbParams)) != hipSuccess || hipMemcpy(pl->P_dev, &pl->P, sizeof(PbParams), hipMemcpyHostToDevice) != hipSuccess) {
        pb_plan_destroy(pl);
        return pb_fail(PB_ERR_HIP, "plan upload failed: parameter block");
    }
    if ((pl->dbl_ready || pl->fast_ready) && pb_build_bilinear_list(pl) != PB_OK) {
        pb_plan_destroy(pl);
        return PB_ERR_HIP;
    }
    rc = pb_build_bilinear_launch(pl);  // derived state: rebuilt, not stored
    if (rc == PB_OK) rc = pb_apply_budget(pl, pl->P.win_budget);
    if (rc != PB_OK) {
        pb_plan_destroy(pl);
        return rc;
    }
    *out = pl;
    return PB_OK;
}

// ---- measurement utility: a plain device copy (16 bytes per lane), the practical HBM ceiling next to which
// bench.py reports the remap kernel (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s by such a copy) -------------
}  // extern "C"
typedef unsigned pb_u32x4 __attribute__((ext_vector_type(4)));
// every workgroup owns one contiguous 32 KiB chunk: 8 non-temporal 16-byte loads per lane in flight, then 8 non-temporal
// stores - the fastest of the copy shapes measured on MI355X (experiments/exp_copy.hip: 6.35 TB/s read + write on 512 MiB;
// a grid-stride loop of single loads 4.9-5.8, hipMemcpyDtoD 5.35)
__global__ __launch_bounds__(256) void pb_copy16_kernel(const pb_u32x4* __restrict__ src, pb_u32x4* __restrict__ dst, size_t n16) {
    const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
    if (base + 7 * 256 < n16) {
        pb_u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + base + u * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(v[u], dst + base + u * 256);
    } else {
        for (size_t i = base; i < n16; i += 256) dst[i] = src[i];
    }
}
extern "C" {
int pb_stream_copy(void* dst_dev, const void* src_dev, size_t bytes, void* stream) {
    if (!dst_dev || !src_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    if ((((uintptr_t)dst_dev | (uintptr_t)src_dev | bytes) & 15u) != 0) return pb_fail(PB_ERR_INVALID, "pb_stream_copy needs 16-byte aligned pointers and size");
    const size_t n16 = bytes / 16;
    if (!n16) return PB_OK;
    const size_t blocks = (n16 + 2047) / 2048;
    if (blocks > 0x7FFFFFFFull) return pb_fail(PB_ERR_INVALID, "pb_stream_copy: too large");
    hipLaunchKernelGGL(pb_copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const pb_u32x4*>(src_dev),
                       static_cast<pb_u32x4*>(dst_dev), n16);
    PB_HIP(hipGetLastError());
    return PB_OK;
}

// ---- multi-GPU: one process per GPU, frames sharded, ONE collective (SURVEY 8 e) -----------------------------------------
// RCCL (librccl.so.1; "nccl" on ROCm, over xGMI inside a node) is bound at first use with dlopen - the remap library itself has
// no link-time dependency on it, and a host that brings its own communicator layer (torch.distributed in the Python package)
// never touches these entry points.  The only data-path collective is the broadcast of the parameter block.
}  // extern "C"
#include <dlfcn.h>
namespace {
struct PbNcclId {
    char internal[128];
};
typedef int (*pb_nccl_get_id_t)(PbNcclId*);
typedef int (*pb_nccl_init_t)(void**, int, PbNcclId, int);
typedef int (*pb_nccl_destroy_t)(void*);
typedef int (*pb_nccl_bcast_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*pb_nccl_err_t)(int);
struct PbRccl {
    void* lib = nullptr;
    pb_nccl_get_id_t get_id = nullptr;
    pb_nccl_init_t init = nullptr;
    pb_nccl_destroy_t destroy = nullptr;
    pb_nccl_bcast_t bcast = nullptr;
    pb_nccl_err_t err = nullptr;
};
PbRccl* pb_rccl() {
    static PbRccl r = [] {
        PbRccl q;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            q.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (q.lib) break;
        }
        if (q.lib) {
            q.get_id = (pb_nccl_get_id_t)dlsym(q.lib, "ncclGetUniqueId");
            q.init = (pb_nccl_init_t)dlsym(q.lib, "ncclCommInitRank");
            q.destroy = (pb_nccl_destroy_t)dlsym(q.lib, "ncclCommDestroy");
            q.bcast = (pb_nccl_bcast_t)dlsym(q.lib, "ncclBroadcast");
            q.err = (pb_nccl_err_t)dlsym(q.lib, "ncclGetErrorString");
        }
        return q;
    }();
    return (r.lib && r.get_id && r.init && r.destroy && r.bcast) ? &r : nullptr;
}
int pb_nccl_fail(const PbRccl* r, const char* what, int code) {
    return pb_fail(PB_ERR_HIP, std::string(what) + ": " + ((r && r->err) ? r->err(code) : "RCCL error"));
}
const int PB_BLOCK_DOUBLES = 2 + 2 * 7 + 9 * PB_MAX_ROTATIONS;  // the parameter block: photonbend_amd/parallel.py's layout
}  // namespace

struct pb_comm {
    void* comm = nullptr;
    int n_ranks = 1, rank = 0, device = -1;
    double* block_dev = nullptr;
};

extern "C" {

int pb_comm_unique_id(void* id128) {
    if (!id128) return pb_fail(PB_ERR_INVALID, "null argument");
    PbRccl* r = pb_rccl();
    if (!r) return pb_fail(PB_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    const int rc = r->get_id(static_cast<PbNcclId*>(id128));
    return rc == 0 ? PB_OK : pb_nccl_fail(r, "ncclGetUniqueId", rc);
}

int pb_comm_init(int n_ranks, int rank, const void* id128, pb_comm** out) {
    if (!out || !id128) return pb_fail(PB_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return pb_fail(PB_ERR_INVALID, "rank outside [0, n_ranks)");
    PbRccl* r = pb_rccl();
    if (!r) return pb_fail(PB_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    pb_comm* c = new (std::nothrow) pb_comm();
    if (!c) return pb_fail(PB_ERR_INVALID, "out of host memory");
    c->n_ranks = n_ranks;
    c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess || hipMalloc((void**)&c->block_dev, PB_BLOCK_DOUBLES * sizeof(double)) != hipSuccess) {
        delete c;
        return pb_fail(PB_ERR_HIP, "pb_comm_init: no device memory for the parameter block");
    }
    PbNcclId id;
    memcpy(&id, id128, sizeof(id));
    const int rc = r->init(&c->comm, n_ranks, id, rank);  // one communicator per process, on the current device
    if (rc != 0) {
        (void)hipFree(c->block_dev);
        delete c;
        return pb_nccl_fail(r, "ncclCommInitRank", rc);
    }
    *out = c;
    return PB_OK;
}

int pb_comm_destroy(pb_comm* comm) {
    if (!comm) return PB_OK;
    PbRccl* r = pb_rccl();
    if (r && comm->comm) (void)r->destroy(comm->comm);
    (void)hipFree(comm->block_dev);
    delete comm;
    return PB_OK;
}

int pb_comm_rank(const pb_comm* comm, int* n_ranks, int* rank) {
    if (!comm) return pb_fail(PB_ERR_INVALID, "null argument");
    if (n_ranks) *n_ranks = comm->n_ranks;
    if (rank) *rank = comm->rank;
    return PB_OK;
}

int pb_bcast_params(pb_comm* comm, pb_proj* dst, double* rot3x3, int* n_rot, pb_proj* src, int root, void* stream) {
    if (!comm || !dst || !src || !n_rot || !rot3x3) return pb_fail(PB_ERR_INVALID, "null argument");
    if (root < 0 || root >= comm->n_ranks) return pb_fail(PB_ERR_INVALID, "root outside [0, n_ranks)");
    PbRccl* r = pb_rccl();
    if (!r) return pb_fail(PB_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    hipStream_t st = (hipStream_t)stream;
    double block[PB_BLOCK_DOUBLES];
    memset(block, 0, sizeof(block));
    auto put = [](double* b, const pb_proj& p) { b[0] = p.kind; b[1] = p.lens; b[2] = p.height; b[3] = p.width; b[4] = p.fov; b[5] = p.magnitude; b[6] = p.f_distance; };
    auto get = [](const double* b, pb_proj& p) { p.kind = (int32_t)b[0]; p.lens = (int32_t)b[1]; p.height = (int32_t)b[2]; p.width = (int32_t)b[3]; p.fov = b[4]; p.magnitude = b[5]; p.f_distance = b[6]; };
    // A collective: once the arguments every rank can check alike have passed (above), EVERY rank reaches the broadcast - a root that
    // finds its own request invalid, or cannot upload it, sends a block with a poisoned magic instead of returning early, and all ranks
    // return PB_ERR_INVALID together (the other ranks used to block in ncclBroadcast forever).
    if (comm->rank == root) {
        const bool ok = *n_rot >= 0 && *n_rot <= PB_MAX_ROTATIONS;
        block[0] = ok ? 1346522692.0 : -1.0;  // "PBND": the Python package's magic (photonbend_amd/parallel.py), same layout
        if (ok) {
            block[1] = *n_rot;
            put(block + 2, *dst);
            put(block + 9, *src);
            for (int k = 0; k < 9 * *n_rot; ++k) block[16 + k] = rot3x3[k];
        }
        if (hipMemcpyAsync(comm->block_dev, block, sizeof(block), hipMemcpyHostToDevice, st) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipMemsetAsync(comm->block_dev, 0xFF, sizeof(block), st);  // (NaN magic: poisoned)
        }
    }
    const int rc = r->bcast(comm->block_dev, comm->block_dev, PB_BLOCK_DOUBLES, 8 /* ncclFloat64 */, root, comm->comm, st);
    if (rc != 0) return pb_nccl_fail(r, "ncclBroadcast", rc);
    PB_HIP(hipMemcpyAsync(block, comm->block_dev, sizeof(block), hipMemcpyDeviceToHost, st));
    PB_HIP(hipStreamSynchronize(st));
    if (block[0] != 1346522692.0 || !(block[1] >= 0 && block[1] <= PB_MAX_ROTATIONS))
        return pb_fail(PB_ERR_INVALID, "the root's parameter block is invalid (n_rot outside [0, PB_MAX_ROTATIONS], a failed upload, or corrupt)");
    *n_rot = (int)block[1];
    get(block + 2, *dst);
    get(block + 9, *src);
    for (int k = 0; k < 9 * *n_rot; ++k) rot3x3[k] = block[16 + k];
    return PB_OK;
}

int pb_shard_range(int n_items, int n_ranks, int rank, int* first, int* count) {
    if (!first || !count) return pb_fail(PB_ERR_INVALID, "null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || n_items < 0) return pb_fail(PB_ERR_INVALID, "bad shard request");
    const int q = n_items / n_ranks, r = n_items % n_ranks;
    *first = rank * q + (rank < r ? rank : r);
    *count = q + (rank < r ? 1 : 0);
    return PB_OK;
}

int pb_remap_batch_sharded(const pb_comm* comm, const pb_plan* plan, const uint8_t* src_dev, uint8_t* dst_dev, int n_frames_total,
                           size_t src_frame_stride, size_t dst_frame_stride, int* first_out, int* count_out, void* stream) {
    if (!comm || !plan || !src_dev || !dst_dev) return pb_fail(PB_ERR_INVALID, "null argument");
    int first = 0, count = 0;
    const int rc = pb_shard_range(n_frames_total, comm->n_ranks, comm->rank, &first, &count);
    if (rc != PB_OK) return rc;
    if (first_out) *first_out = first;
    if (count_out) *count_out = count;
    if (count == 0) return PB_OK;
    const PbParams& P = plan->P;
    if (!src_frame_stride) src_frame_stride = 3ull * P.src.height * P.src.width;
    if (!dst_frame_stride) dst_frame_stride = 3ull * P.dst.height * P.dst.width;
    // this rank's contiguous share of the batch, addressed in the caller's (rank-local) buffers from frame 0 on
    return pb_remap_u8(plan, src_dev, dst_dev, count, src_frame_stride, dst_frame_stride, stream);
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
int pb_device_count(int* n) {
    if (!n) return pb_fail(PB_ERR_INVALID, "null argument");
    *n = 0;
    if (hipGetDeviceCount(n) != hipSuccess) {
        (void)hipGetLastError();
        *n = 0;
    }
    return PB_OK;
}
int pb_set_device(int device) {
    PB_HIP(hipSetDevice(device));
    return PB_OK;
}
int pb_get_device(int* device) {
    if (!device) return pb_fail(PB_ERR_INVALID, "null argument");
    PB_HIP(hipGetDevice(device));
    return PB_OK;
}
int pb_device_sync(void) {
    PB_HIP(hipDeviceSynchronize());
    return PB_OK;
}
int pb_stream_wait_event(void* stream, void* event) {
    if (!event) return pb_fail(PB_ERR_INVALID, "null argument");
    PB_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return PB_OK;
}
int pb_host_alloc(void** host_ptr, size_t bytes) {
    if (!host_ptr) return pb_fail(PB_ERR_INVALID, "null argument");
    PB_HIP(hipHostMalloc(host_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return PB_OK;
}
int pb_host_free(void* host_ptr) {
    if (host_ptr) PB_HIP(hipHostFree(host_ptr));
    return PB_OK;
}
int pb_host_register(void* host_ptr, size_t bytes) {
    if (!host_ptr || !bytes) return pb_fail(PB_ERR_INVALID, "null argument");
    PB_HIP(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    return PB_OK;
}
int pb_host_unregister(void* host_ptr) {
    if (!host_ptr) return pb_fail(PB_ERR_INVALID, "null argument");
    PB_HIP(hipHostUnregister(host_ptr));
    return PB_OK;
}

}  // extern "C"
