/*
 * photonbend_hip.h - C ABI of the MI355X (gfx950) per-pixel lens remapper.
 *
 * This is the drop-in boundary for photonbend's core remap path.  Every entry
 * point is extern "C", takes plain pointers and sizes, returns 0 on success or
 * a negative pb_status (message via pb_last_error()), never throws, never
 * allocates or synchronises inside a launch function (so a caller may capture
 * launches into a hipGraph), and works on caller-owned DEVICE buffers on a
 * caller-chosen HIP stream (passed as void*; NULL = the default stream).
 *
 * Reference interfaces replaced (all paths under /root/reference/photonbend):
 *   pb_proj            the state of CameraImage (core/projection.py:86-121),
 *                      DoubleCameraImage (:296-316) and PanoramaImage (:477-485)
 *   pb_plan_create     dst.get_coordinate_map() -> Rotation.rotate_coordinate_map()*
 *                      -> src.process_coordinate_map()  chained lazily
 *                      (core/__init__.py:66-92)
 *   pb_remap_u8        that chain, fused: one work-item per output pixel
 *   pb_index_map_i32   the integer coordinate map inside process_coordinate_map
 *                      (projection.py:254-259, :545)
 *   pb_coordmap_f64    get_coordinate_map()  (projection.py:147, :341, :487)
 *   pb_rotate_f64      Rotation.rotate_coordinate_map()  (core/rotation.py:102-176)
 *   pb_sample_map_u8   process_coordinate_map(ndarray)  (projection.py:197, :408, :515)
 *
 * Images are uint8 (H, W, 3) RGB, row-major, tightly packed (core/__init__.py:31-36).
 * Coordinate maps are float64 (H, W, 3) = (latitude, longitude, invalid flag)
 * (core/__init__.py:42-49).  Angles are radians.
 */
#ifndef PHOTONBEND_HIP_H
#define PHOTONBEND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define PB_ABI_VERSION 5
#define PB_MAX_ROTATIONS 8

typedef enum pb_status {
    PB_OK = 0,
    PB_ERR_INVALID = -1,   /* bad argument (null pointer, size, enum value) */
    PB_ERR_HIP = -2,       /* a HIP runtime call failed */
    PB_ERR_UNSUPPORTED = -3,
    PB_ERR_NO_DEVICE = -4
} pb_status;

/* projection.py class -> kind */
typedef enum pb_kind {
    PB_KIND_CAMERA = 0, /* CameraImage        projection.py:69  */
    PB_KIND_DOUBLE = 1, /* DoubleCameraImage  projection.py:277 */
    PB_KIND_PANO = 2    /* PanoramaImage      projection.py:465 */
} pb_kind;

/* core/lens.py factories :341-401 */
typedef enum pb_lens {
    PB_LENS_EQUIDISTANT = 0,
    PB_LENS_EQUISOLID = 1,
    PB_LENS_RECTILINEAR = 2,
    PB_LENS_STEREOGRAPHIC = 3,
    PB_LENS_ORTHOGRAPHIC = 4,
    PB_LENS_THOBY = 5,
    PB_LENS_CUSTOM = 6 /* Lens(forward_function, reverse_function) of arbitrary Python callables (lens.py:48-64): the HOST
                          evaluates them; accepted only by pb_index_from_map_i32 together with its distance planes */
} pb_lens;

/* One end of a remap.  f_distance is computed by the HOST with the reference
 * formula magnitude / forward_lens(fov / 2) (projection.py:141-144) so that its
 * bits are the caller's; the kernels never recompute it. */
typedef struct pb_proj {
    int32_t kind;      /* pb_kind */
    int32_t lens;      /* pb_lens (ignored for PB_KIND_PANO) */
    int32_t height;
    int32_t width;
    double fov;        /* radians; the per-sensor fov for PB_KIND_DOUBLE */
    double magnitude;  /* informational (already folded into f_distance) */
    double f_distance; /* focal distance in pixels */
} pb_proj;

typedef struct pb_plan pb_plan; /* opaque, immutable after creation */

/* ---- library / device ------------------------------------------------- */
int pb_abi_version(void);
/* 0: the float64 chain runs NumPy's AVX-512 arcsin / arccos / arctan / tan (hosts with AVX512_SKX); 1: glibc's (hosts without).  See
 * PB_PLAN_MATH_SVML / PB_PLAN_MATH_LIBM. */
int pb_math_flavour(void);
const char* pb_last_error(void); /* thread-local, valid until the next failing call */
int pb_init(int device);         /* hipSetDevice + sanity checks (gfx950 expected) */
int pb_shutdown(void);           /* returns the idle blocks of the plan-preparation cache (<= 96 MiB of device memory) to the driver; plans stay valid */
int pb_device_name(char* buf, size_t buflen);

/* ---- the fused hot path ------------------------------------------------ */
/* rot3x3: n_rot row-major 3x3 float64 matrices = Rotation.rotation_matrix
 * (core/rotation.py:100), applied in order; n_rot in [0, PB_MAX_ROTATIONS].
 * When a HIP device is visible, creation prepares the plan on the CURRENT device (synchronously,
 * about the cost of two faithful frames): destination-validity thresholds, per-tile models (one
 * table per eye for a double-fisheye source), the exhaustive comparison with the faithful chain, and
 * the exact lookup tables for everything the models miss (source indices of failed tiles and fix
 * pixels; for double sources also their blend factors and the latitudes of merge-band tiles).  The
 * plan then owns device memory on that device (256 B per 32x32 output tile and eye + the tables)
 * until pb_plan_destroy. */
int pb_plan_create(const pb_proj* dst, const double* rot3x3, int n_rot, const pb_proj* src, pb_plan** out);
void pb_plan_destroy(pb_plan* plan);

/* Explicit form of pb_plan_create (which is pb_plan_create_ex(..., 0, 0, out)).
 *   flags       PB_PLAN_DEFER  no device work at creation: the plan is a parameter block and its launches run the
 *                              faithful float64 kernel (the cheapest way to remap ONE image of a geometry, the
 *                              reference CLI's case) until pb_plan_prepare builds the fast path;
 *               PB_PLAN_TUNE   after preparation, pick the LDS window budget (four candidates) and then the launch order
 *                              (the policy's walk against plain / heaviest rows first / heaviest super-tiles first) by
 *                              timing launches on scratch frames (allocates frame-sized scratch, about a hundred frames'
 *                              worth of GPU time; opt-in).  Both decide paths and order only, never a byte; a serialized
 *                              plan remembers them.
 *   win_budget  bytes of LDS window per wave for the hot kernels (multiple of 16 in [4224, 12288]; clamped);
 *               0 = the library default (7168).  It decides which PATH a tile takes, never its pixels.
 * Creation never times anything or allocates frame-sized memory unless PB_PLAN_TUNE is given. */
#define PB_PLAN_DEFER 1u
#define PB_PLAN_TUNE 2u
/* MATH FLAVOURS (ABI 5).  The reference's float64 results depend on which code NumPy dispatches to on the HOST it runs on: with
 * AVX512_SKX np.arcsin / arccos / arctan / tan (core/lens.py:71-307, core/rotation.py:158) are NumPy's own AVX-512 kernels, without it
 * they are libm's (glibc 2.35: asin / acos / atan / tan) - 7-8 % of the arcsin / arccos results differ in the last bit, and with them
 * whole texels of identity-type remaps.  The library comes in both flavours, built from the same sources: libphotonbend_hip.so
 * (pb_math_flavour() == 0: the AVX-512 kernels' bits, the platform of tests/golden/) and libphotonbend_hip_libm.so (== 1: libm's bits,
 * tests/golden/npmath_libm.npz, libm_flavour.json); a host loads the one that matches ITS reference (photonbend_amd/_native.py does:
 * NumPy's own dispatch decides, PB_MATH_FLAVOUR=svml|libm overrides).  The two flags below let a caller STATE which flavour a plan must
 * have: pb_plan_create_ex returns PB_ERR_UNSUPPORTED from the library of the other flavour instead of silently giving other bits. */
#define PB_PLAN_MATH_SVML 4u
#define PB_PLAN_MATH_LIBM 8u
/* THE OPT-IN BILINEAR MODE'S TABLES (flag values added in round 6; ABI 5 unchanged).  A prepared plan carries, besides what the reference's
 * nearest sampler needs, the opt-in bilinear mode's state (exact coordinate tables, its own launch table and LDS pool: 0.2-0.3 ms of a
 * c2 plan's 0.85 ms).  PB_PLAN_NO_BILINEAR (pb_plan_create_ex) leaves it out - for callers of pb_remap_u8, the reference's path;
 * PB_PLAN_BILINEAR (pb_plan_prepare, synchronous, no launch of the plan in flight) builds it later.  pb_remap_bilinear_u8 on a plan
 * without it is still correct - it runs the mode's per-pixel float64 kernels, several times slower - and pb_plan_bilinear_float64_tiles
 * says so (every tile).  A deferred plan (PB_PLAN_DEFER) remembers either choice for its preparation.  Default (neither flag): as before.
 * A double-fisheye source whose field of view lies within one degree of 180 (180 itself excepted) never gets the tables: the reference's blend factor
 * past the end of so narrow a merge band (projection.py:440-444) multiplies any sampler's rounding; the mode's float64 kernels serve it. */
#define PB_PLAN_NO_BILINEAR 16u
#define PB_PLAN_BILINEAR 32u
int pb_plan_create_ex(const pb_proj* dst, const double* rot3x3, int n_rot, const pb_proj* src, unsigned flags,
                      int win_budget, pb_plan** out);
/* Builds the fast path of a deferred plan on the current device (flags: 0, PB_PLAN_TUNE, PB_PLAN_BILINEAR).  On a prepared plan
 * it applies win_budget (when > 0) and, with PB_PLAN_BILINEAR, builds the bilinear mode's tables if the plan lacks them.  Synchronous. */
int pb_plan_prepare(pb_plan* plan, unsigned flags, int win_budget);
/* Re-classifies the tiles of a prepared plan under another window budget (synchronous, cheap: one small kernel). */
int pb_plan_set_window_budget(pb_plan* plan, int win_budget);
/* Host wall time (ms) of the device preparation and of the optional tuning; either pointer may be NULL. */
int pb_plan_timing(const pb_plan* plan, double* prepare_ms, double* tune_ms);
/* A prepared plan as a host blob and back, so that a process (the CLI's one-image case) need not pay the
 * per-pixel certification again for a geometry it has seen: pass buf = NULL to query the size.  The blob is
 * valid for this library build only (checked) and carries a checksum; deserialisation uploads the tables to
 * the CURRENT device.  The blob names the math flavour that made its exact tables: the library of the other flavour
 * refuses it with PB_ERR_UNSUPPORTED (a blob cache shared by hosts with and without AVX-512 keeps one blob per flavour). */
int pb_plan_serialize(const pb_plan* plan, void* buf, size_t capacity, size_t* size_out);
int pb_plan_deserialize(const void* buf, size_t size, pb_plan** out);
/* Execution mode of a plan.  AUTO (default) and FAST: ONE launch of the hot kernel per call (per-tile
 * float32 polynomial models of the coordinate field; what the models miss is looked up in the plan's
 * exact tables), when the plan was prepared on a device; otherwise, and under FAITHFUL, the per-pixel
 * float64 chain for every pixel.  Both produce identical bytes: the tables hold, by construction at
 * plan creation, the faithful result for every pixel whose model index differs from the faithful one. */
#define PB_MODE_AUTO 0
#define PB_MODE_FAITHFUL 1
#define PB_MODE_FAST 2
#define PB_MODE_FAST_DIRECT 3 /* FAST without LDS windows: direct-gather hot kernel + fix kernel, the path of frames that are
                                 not 16-byte aligned (double sources: the separable / faithful kernels) */
int pb_plan_set_mode(pb_plan* plan, int mode);
/* fast_path_enabled: 0/1 under the current mode; stats7: {32x32 tiles, tiles handled whole by the
 * tables ("failed"), single pixels on the fix list, pixels where model and faithful index differed,
 * tiles on the lean LDS-window path, all-black tiles, tiles on the lean direct-gather path} (-1 when the
 * plan has no device state; a double-fisheye source counts both eyes' tables in the last four);
 * thresholds4:
 * {invalid_lo, invalid_hi} on (2x)^2+(2y)^2 for the left/single and the right eye of the
 * destination.  Any pointer may be NULL. */
int pb_plan_info(const pb_plan* plan, int* fast_path_enabled, long long* stats7, long long* thresholds4);
int pb_plan_dst_shape(const pb_plan* plan, int* height, int* width);
int pb_plan_src_shape(const pb_plan* plan, int* height, int* width);
/* bytes of LDS window per wave the plan's hot launches use (it decides which path a tile takes, never its
 * pixels); 0 without device state */
int pb_plan_window_budget(const pb_plan* plan);
/* How many 32x32 tiles of a prepared plan the opt-in bilinear mode (pb_remap_bilinear_u8) recomputes with the float64 chain per
 * pixel instead of the tile models: failed tiles, tiles whose model is further than 1/1024 px from the faithful coordinate
 * somewhere (fine for the reference's truncating sampler, whose exceptions are tabulated; too coarse to interpolate at), and
 * - double-fisheye sources - tiles that are not plain for an eye that sees them.  0 for a deferred plan. */
int pb_plan_bilinear_float64_tiles(const pb_plan* plan);
/* Diagnostics of the opt-in bilinear mode (ABI 5; synchronous, not for the hot loop): how its launch serves the tile entries its waves
 * read - mix[0] LDS-window entries, [1] direct-gather entries, [2] entries read from the exact coordinate table, [3] black entries,
 * [4] window / direct entries whose coordinate is evaluated on the certified low-degree part of the tile model (PB_TILE_TD3),
 * [5] entries counted (one per tile; a double-fisheye plan's two-eye tiles have one per eye), [6] of the window entries [0], those
 * staged as two half windows (their source box exceeds the budget, its top and bottom halves do not), [7] of the table entries [2],
 * those whose taps all lie inside the frame (read without clamps or wrap).  All zero without tile tables. */
int pb_plan_bilinear_tile_mix(const pb_plan* plan, long long mix[8]);
/* ... and the shape of its launch (ABI 5): the dynamic LDS of a workgroup (the pool its four waves' windows are packed into: 40 448 bytes -
 * four workgroups per CU - where at most 2 % of the tiles lose their window to it, else four full-budget regions) and the workgroups
 * of one frame (four waves = four tiles each; the grid of an n-frame launch is n times that); both 0 without tile tables. */
int pb_plan_bilinear_launch_shape(const pb_plan* plan, int* lds_bytes, int* workgroups_per_frame);

/* 1 when `plan` was made for exactly this request (same projections, same rotation bits), 0 when not, negative on bad
 * arguments (ABI 3).  What a cache of serialized plans checks after pb_plan_deserialize: the blob's checksum says it is
 * intact, not that it belongs to the geometry the caller has in mind. */
int pb_plan_matches(const pb_plan* plan, const pb_proj* dst, const double* rot3x3, int n_rot, const pb_proj* src);

/* Remap n_frames frames that share the plan's geometry.  Frame f is read at
 * src_dev + f * src_frame_stride and written at dst_dev + f * dst_frame_stride
 * (strides in bytes, at least a frame; pass 0 for tightly packed frames).  ONE kernel launch whatever n_frames:
 * the frames of a batch are a dimension of the grid, so the launch ramp and drain are paid once per call
 * (measured on MI355X: 8 frames per call run 1.3x faster per frame than 8 calls).  Asynchronous on `stream`;
 * never allocates or synchronises (graph-capture safe).  A deferred plan (PB_PLAN_DEFER, not yet prepared) runs
 * the faithful float64 kernel.  (A double-fisheye -> unrotated panorama plan keeps a second, separable fast path for frames the windowed
 * kernel cannot take - source pointer or stride not a multiple of 16 bytes, PB_MODE_FAST_DIRECT; its tables are verified against the
 * float64 chain by a kernel the FIRST such launch enqueues on `stream`: that launch, launches inside a stream capture, and any launch
 * issued before the verification has finished run the float64 kernel instead - same bytes, slower.) */
int pb_remap_u8(const pb_plan* plan, const uint8_t* src_dev, uint8_t* dst_dev, int n_frames,
                size_t src_frame_stride, size_t dst_frame_stride, void* stream);

/* The same batch launch for frames that are NOT at a uniform stride - a ring of separately allocated buffers (ABI 5).
 * src_dev / dst_dev are HOST arrays of n_frames DEVICE pointers (tightly packed frames); frame f is read at src_dev[f] and
 * written at dst_dev[f].  ONE kernel launch per 64 frames on `stream`: the pointers travel in the kernel-argument segment
 * (read at launch: the arrays may be reused when the call returns), so the call stays allocation- and synchronisation-free
 * (graph-capture safe) like pb_remap_u8, whose bytes it reproduces.  Frames the windowed kernels cannot take (a source
 * pointer not 16-byte aligned), deferred plans and PB_MODE_FAITHFUL / PB_MODE_FAST_DIRECT run as n_frames single launches.
 * Replaces: a host loop over process_coordinate_map() (core/__init__.py:66-92) on separately allocated arrays. */
int pb_remap_u8v(const pb_plan* plan, const uint8_t* const* src_dev, uint8_t* const* dst_dev, int n_frames, void* stream);

/* OPT-IN extension with no reference counterpart (the reference samples nearest-by-truncation only):
 * bilinear interpolation at the reference's pre-truncation coordinate (pixel k covers [k, k+1), centre
 * k + 0.5; taps clamped to the image, panorama columns wrap; round half to even).  Pixels the nearest mode
 * paints black stay black.  A double-fisheye source is the reference's blend (projection.py:439-460) of the two eyes'
 * bilinear samples (each eye sampled like a camera source on its half, rounded to uint8, then blended and cast as the
 * nearest mode does); it runs through the per-eye tile models like the nearest mode (ABI 3; pixels the models cannot serve
 * are recomputed by the float64 chain in the same call). */
int pb_remap_bilinear_u8(const pb_plan* plan, const uint8_t* src_dev, uint8_t* dst_dev, int n_frames,
                         size_t src_frame_stride, size_t dst_frame_stride, void* stream);

/* Integer coordinate map: for camera / pano sources idx_dev is int32 [H*W], the
 * linear source pixel index (row * src_width + col) or -1 where the output is
 * black.  For a double source idx_dev is int32 [2][H*W] (left-eye index, then
 * right-eye index, both into the full side-by-side frame) and, when weights_dev
 * is not NULL, float64 [2][H*W] receives the two blend factors
 * (projection.py:439-457). */
int pb_index_map_i32(const pb_plan* plan, int32_t* idx_dev, double* weights_dev, void* stream);

/* ---- multi-GPU: one process per GPU (SURVEY 8 e; ABI 3) ------------------------------------------------------------------
 * Frames are independent: a batch shards contiguously over the ranks and no pixel crosses a link.  The ONE data-path collective
 * is the RCCL broadcast (ncclBroadcast over xGMI inside a node) of the parameter block - both projections with the host-side
 * f_distance bits, the rotation matrices - from the root rank, so every device computes with identical bits; each rank then
 * creates its own plan (pb_plan_create on its device).  The reference has no counterpart (single process, no collective);
 * these entry points are what a non-Python host binds where the Python package uses torch.distributed
 * (photonbend_amd/parallel.py: same block layout).  librccl.so.1 is loaded at first use; PB_ERR_UNSUPPORTED without it.
 *   pb_comm_unique_id   rank 0 makes the 128-byte rendezvous id (ncclGetUniqueId); the launcher hands it to the other ranks
 *   pb_comm_init        collective: every rank, on ITS current device (ncclCommInitRank)
 *   pb_bcast_params     collective: root's dst / rot3x3[9 * n_rot] / n_rot / src overwrite everyone else's (rot3x3 must hold
 *                       9 * PB_MAX_ROTATIONS doubles); synchronises `stream`.  PRECONDITION: non-null pointers and a root inside
 *                       [0, n_ranks) on EVERY rank - those are checked before anyone enters the collective and must not differ
 *                       between ranks; everything only the root can know (its n_rot, its upload) travels WITH the broadcast: an
 *                       invalid root request makes every rank return PB_ERR_INVALID together, none is left waiting
 *   pb_shard_range      first = rank * q + min(rank, r), count = q + (rank < r)  (q, r = divmod(n_items, n_ranks))
 *   pb_remap_batch_sharded  remaps THIS rank's share of a batch of n_frames_total frames: its count frames, resident in its own
 *                       buffers from src_dev / dst_dev on (frame k of the share = batch frame first + k); one pb_remap_u8 */
typedef struct pb_comm pb_comm;
int pb_comm_unique_id(void* id128);
int pb_comm_init(int n_ranks, int rank, const void* id128, pb_comm** out);
int pb_comm_destroy(pb_comm* comm);
int pb_comm_rank(const pb_comm* comm, int* n_ranks, int* rank);
int pb_bcast_params(pb_comm* comm, pb_proj* dst, double* rot3x3, int* n_rot, pb_proj* src, int root, void* stream);
int pb_shard_range(int n_items, int n_ranks, int rank, int* first, int* count);
int pb_remap_batch_sharded(const pb_comm* comm, const pb_plan* plan, const uint8_t* src_dev, uint8_t* dst_dev, int n_frames_total,
                           size_t src_frame_stride, size_t dst_frame_stride, int* first_out, int* count_out, void* stream);

/* ---- materialised coordinate-map API (protocol compatibility) ---------
 * The float64 values are the reference's BITS as computed on x86-64 with FMA and AVX512_SKX, glibc 2.35, NumPy 2.2.6 (the platform of
 * tests/golden/): the device runs that platform's sin / cos / sincos / atan2 and NumPy's arcsin / arccos / arctan / tan restated bit
 * for bit (csrc/pb_math_glibc.hpp, csrc/pb_math_np.hpp); the same functions define every index map and remap. */
int pb_coordmap_f64(const pb_proj* dst, double* map_dev, void* stream);
/* Zeroes lat/lon of invalid pixels IN map_in_dev (rotation.py:119-125), like the
 * reference; map_out_dev must not alias map_in_dev. */
int pb_rotate_f64(const double* rot3x3, double* map_in_dev, double* map_out_dev, int height, int width,
                  void* stream);
/* map is (height, width, 3); PB_KIND_PANO zeroes invalid lat/lon in map_dev
 * (projection.py:534-536). */
int pb_sample_map_u8(const pb_proj* src, double* map_dev, int height, int width, const uint8_t* src_dev,
                     uint8_t* dst_dev, void* stream);

/* The sampling half of process_coordinate_map without the gather (projection.py:197-260 camera, :408-462 double,
 * :515-547 panorama): coordinate map (height, width, 3) -> int32 source index per pixel (-1 = black; a double source:
 * [2][height*width], left eye then right eye, and weights_dev float64 [2][height*width] when not NULL).  PB_KIND_PANO
 * zeroes invalid lat/lon in map_dev like the reference.  dist_l_dev / dist_r_dev (float64 [height*width], optional):
 * forward_lens(latitude) * f_distance evaluated by the host for a PB_LENS_CUSTOM source (the right eye's plane is
 * forward_lens(pi - latitude) * f_distance); with them the built-in forward lens is not used.
 * Together with pb_gather_px / pb_gather_blend_u8 this serves every image the reference accepts but the fused
 * uint8 RGB path does not: grey (H, W), RGBA (H, W, 4), 16-bit samples - the reference fancy-indexes whatever array
 * it is given (projection.py:234-243, :545-546). */
int pb_index_from_map_i32(const pb_proj* src, double* map_dev, int height, int width, const double* dist_l_dev,
                          const double* dist_r_dev, int32_t* idx_dev, double* weights_dev, void* stream);
/* The opt-in bilinear mode on a MATERIALISED map and on ANY image (ABI 5): what process_coordinate_map(map, interpolation="bilinear")
 * runs when the map is an array (looked at or edited between the stages; a destination Lens of user callables), when the image is not
 * uint8 RGB (grey, RGBA, 16-bit samples) or when the source Lens is made of user callables (dist_l_dev / dist_r_dev as in
 * pb_index_from_map_i32).  The definition of pb_remap_bilinear_u8, evaluated per pixel in float64 from the map's (lat, lon, invalid):
 * taps floor(f - 0.5), + 1 clamped to the image (a panorama's columns wrap, an eye's taps stay in its half), round half to even,
 * clamped to the sample range; PB_KIND_PANO zeroes invalid lat/lon in map_dev like the reference.  img_dev: (h, w, channels)
 * samples of sample_bytes (1 or 2) bytes; out_dev: (height, width, channels) of the same sample type - a double-fisheye source
 * returns uint8 like the reference's (left * fl + right * fr).astype(np.uint8) (projection.py:447-460).
 * Replaces: nothing in the reference (it truncates: projection.py:254-259, :545); completes the ProjectionImage protocol
 * (projection.py:40-66, :197-245, :515-547) for the opt-in mode.  pb_sample_map_bilinear_u8 = 3 channels of 1 byte. */
int pb_sample_map_bilinear_px(const pb_proj* src, double* map_dev, int height, int width, const double* dist_l_dev, const double* dist_r_dev,
                              const void* img_dev, void* out_dev, int channels, int sample_bytes, void* stream);
int pb_sample_map_bilinear_u8(const pb_proj* src, double* map_dev, int height, int width, const uint8_t* src_dev, uint8_t* dst_dev, void* stream);
/* dst[p] = idx[p] < 0 ? 0 : src[idx[p]], bytes_per_px bytes each (1..64). */
int pb_gather_px(const int32_t* idx_dev, const void* src_dev, void* dst_dev, size_t n_px, int bytes_per_px, void* stream);
/* The double-fisheye blend for `channels` interleaved samples of 1 or 2 bytes (unsigned): per channel
 * (left * fl + right * fr).astype(uint8) - uint8 output whatever the input width, like the reference
 * (projection.py:447-460).  idx2_dev / weights2_dev as written by pb_index_map_i32 / pb_index_from_map_i32. */
int pb_gather_blend_u8(const int32_t* idx2_dev, const double* weights2_dev, const void* src_dev, uint8_t* dst_dev,
                       size_t n_px, int channels, int sample_bytes, void* stream);

/* map_projection (projection.py:550-599): coordinate map -> colour map (red = latitude stretched to
 * 0..255 over the valid pixels, green = longitude * 255 / 2pi, blue = invalid * 255).  Zeroes invalid
 * lat/lon in map_dev like the reference.  workspace24_dev: 24 bytes of device scratch; after the call it
 * holds {min key, max key, number of valid pixels} (uint64 each). */
int pb_map_projection_u8(double* map_dev, int height, int width, uint8_t* out_dev, void* workspace24_dev, void* stream);

/* ---- deterministic synthetic frames (bench + tests) -------- */
/* circle_mask: 0 none, 1 black outside the inscribed circle, 2 black outside
 * the two side-by-side inscribed circles. */
int pb_synth_frame_u8(uint8_t* frame_dev, int height, int width, uint32_t frame, uint32_t seed, int circle_mask,
                      void* stream);

/* ---- plumbing for hosts without their own device allocator ------------ */
int pb_malloc(void** dev_ptr, size_t bytes);
int pb_free(void* dev_ptr);
int pb_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int pb_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int pb_memset(void* dst_dev, int value, size_t bytes, void* stream);
int pb_stream_create(void** stream);
int pb_stream_destroy(void* stream);
int pb_stream_sync(void* stream);
int pb_event_create(void** event);
int pb_event_destroy(void* event);
int pb_event_record(void* event, void* stream);
int pb_event_sync(void* event);
int pb_event_elapsed_ms(void* start, void* stop, float* ms);
/* ABI 4: what a host that moves frames between ITS memory and the device needs beyond the above (the reference's contract is
 * ndarray in, fresh ndarray out: core/__init__.py:66-92; the Python package's NumPy path is built on these, without PyTorch).
 *   pb_device_count / pb_set_device / pb_get_device   the device ordinals pb_init takes
 *   pb_device_sync          hipDeviceSynchronize
 *   pb_stream_wait_event    work queued on `stream` after this call waits for `event` (cross-stream ordering: upload -> remap -> download)
 *   pb_host_alloc / _free   page-locked host memory (hipHostMalloc): DMA reads / writes it without a staging copy
 *   pb_host_register / _unregister   page-lock memory the CALLER allocated (hipHostRegister) for as long as it is registered; the
 *                           caller must unregister before the memory is freed or unmapped */
int pb_device_count(int* n);
int pb_set_device(int device);
int pb_get_device(int* device);
int pb_device_sync(void);
int pb_stream_wait_event(void* stream, void* event);
int pb_host_alloc(void** host_ptr, size_t bytes);
int pb_host_free(void* host_ptr);
int pb_host_register(void* host_ptr, size_t bytes);
int pb_host_unregister(void* host_ptr);
/* measurement utility: a plain 16-byte-per-lane device copy (pointers and size multiples of 16) - the practical
 * HBM ceiling bench.py reports next to the remap kernel */
int pb_stream_copy(void* dst_dev, const void* src_dev, size_t bytes, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PHOTONBEND_HIP_H */
