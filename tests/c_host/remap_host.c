/* remap_host.c - a plain C host of the C ABI (no Python, no torch, no HIP headers): what a maintainer binding
 * include/photonbend_hip.h from a compiled language writes.  Builds a plan for a rotated fisheye view of a panorama,
 * remaps a batch of synthetic frames, and checks the bytes three ways: the fast path against the faithful float64 kernel,
 * against a gather through pb_index_map_i32, and frame k of the batch against a single-frame launch.  The same for a
 * double-fisheye stitch.  Exit code 0 and "c host ok" on success.
 *   gcc -std=c99 -O1 -Iinclude tests/c_host/remap_host.c -Lphotonbend_amd -lphotonbend_hip -Wl,-rpath,$PWD/photonbend_amd -lm
 * Reference call sequence reproduced: core/__init__.py:66-92. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "photonbend_hip.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != 0) {                                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pb_last_error());            \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static const double PI = 3.141592653589793;

/* Rotation(pitch, yaw, roll).rotation_matrix (core/rotation.py:27-62, :100: the angles are negated) */
static void rotation_matrix(double pitch, double yaw, double roll, double R[9]) {
    const double p = -pitch, y = -yaw, r = -roll;
    const double P[9] = {1, 0, 0, 0, cos(p), sin(p), 0, -sin(p), cos(p)};
    const double Y[9] = {cos(y), 0, -sin(y), 0, 1, 0, sin(y), 0, cos(y)};
    const double L[9] = {cos(r), sin(r), 0, -sin(r), cos(r), 0, 0, 0, 1};
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T[3 * i + j] = 0;
            for (int k = 0; k < 3; ++k) T[3 * i + j] += P[3 * i + k] * Y[3 * k + j];
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            R[3 * i + j] = 0;
            for (int k = 0; k < 3; ++k) R[3 * i + j] += T[3 * i + k] * L[3 * k + j];
        }
}

static int run_case(const char* name, const pb_proj* dst, const double* rot, int n_rot, const pb_proj* src, int mask) {
    const size_t sb = (size_t)3 * src->height * src->width, db = (size_t)3 * dst->height * dst->width, npx = (size_t)dst->height * dst->width;
    const int n_frames = 3, is_double = src->kind == PB_KIND_DOUBLE;
    pb_plan* plan = NULL;
    void *s_dev = NULL, *d_fast = NULL, *d_faith = NULL, *d_one = NULL, *idx = NULL, *wts = NULL, *stream = NULL;
    CHECK(pb_stream_create(&stream));
    CHECK(pb_plan_create(dst, rot, n_rot, src, &plan));
    if (pb_plan_matches(plan, dst, rot, n_rot, src) != 1) return fprintf(stderr, "%s: pb_plan_matches says no\n", name), 1;
    CHECK(pb_malloc(&s_dev, sb * n_frames));
    CHECK(pb_malloc(&d_fast, db * n_frames));
    CHECK(pb_malloc(&d_faith, db * n_frames));
    CHECK(pb_malloc(&d_one, db));
    CHECK(pb_malloc(&idx, npx * 4 * (is_double ? 2 : 1)));
    if (is_double) CHECK(pb_malloc(&wts, npx * 8 * 2));
    for (int f = 0; f < n_frames; ++f) CHECK(pb_synth_frame_u8((uint8_t*)s_dev + f * sb, src->height, src->width, (uint32_t)f, 0u, mask, stream));
    int fast = 0;
    long long stats[7], thr[4];
    CHECK(pb_plan_info(plan, &fast, stats, thr));
    if (!fast) return fprintf(stderr, "%s: the plan has no fast path\n", name), 1;
    CHECK(pb_remap_u8(plan, s_dev, d_fast, n_frames, 0, 0, stream));               /* one launch, three frames */
    CHECK(pb_remap_u8(plan, (uint8_t*)s_dev + sb, d_one, 1, 0, 0, stream));         /* frame 1 alone */
    CHECK(pb_plan_set_mode(plan, PB_MODE_FAITHFUL));
    CHECK(pb_remap_u8(plan, s_dev, d_faith, n_frames, 0, 0, stream));
    CHECK(pb_plan_set_mode(plan, PB_MODE_AUTO));
    CHECK(pb_index_map_i32(plan, idx, wts, stream));
    uint8_t *h_fast = malloc(db * n_frames), *h_faith = malloc(db * n_frames), *h_one = malloc(db), *h_g = malloc(db);
    CHECK(pb_memcpy_d2h(h_fast, d_fast, db * n_frames, stream));
    CHECK(pb_memcpy_d2h(h_faith, d_faith, db * n_frames, stream));
    CHECK(pb_memcpy_d2h(h_one, d_one, db, stream));
    /* the same pixels through the integer coordinate map (frame 0) */
    if (is_double)
        CHECK(pb_gather_blend_u8(idx, wts, s_dev, d_one, npx, 3, 1, stream));
    else
        CHECK(pb_gather_px(idx, s_dev, d_one, npx, 3, stream));
    CHECK(pb_memcpy_d2h(h_g, d_one, db, stream));
    CHECK(pb_stream_sync(stream));
    size_t nz = 0;
    for (size_t i = 0; i < db; ++i) nz += h_fast[i] != 0;
    int bad = 0;
    if (memcmp(h_fast, h_faith, db * n_frames)) bad |= 1;
    if (memcmp(h_fast + db, h_one, db)) bad |= 2;
    if (memcmp(h_fast, h_g, db)) bad |= 4;
    if (!memcmp(h_fast, h_fast + db, db)) bad |= 8; /* distinct frames must give distinct outputs */
    if (nz < db / 4) bad |= 16;
    printf("%-18s %dx%d <- %dx%d, %d frames: fast %s faithful, batch frame %s single launch, %s index-map gather; %zu of %zu bytes non-zero\n", name,
           dst->height, dst->width, src->height, src->width, n_frames, (bad & 1) ? "!=" : "==", (bad & 2) ? "!=" : "==", (bad & 4) ? "!=" : "==", nz, db);
    free(h_fast); free(h_faith); free(h_one); free(h_g);
    pb_free(s_dev); pb_free(d_fast); pb_free(d_faith); pb_free(d_one); pb_free(idx);
    if (wts) pb_free(wts);
    pb_plan_destroy(plan);
    pb_stream_destroy(stream);
    return bad;
}

int main(void) {
    if (pb_abi_version() != PB_ABI_VERSION) return fprintf(stderr, "ABI %d, header %d\n", pb_abi_version(), PB_ABI_VERSION), 1;
    CHECK(pb_init(0));
    char dev[128];
    CHECK(pb_device_name(dev, sizeof dev));
    printf("device: %s\n", dev);
    int bad = 0;
    {   /* make-photo: 1024x2048 panorama -> 768x768 equisolid fisheye, 200 degrees, rotated */
        const double fov = 200.0 / 180.0 * PI, mag = 768 / 2.0 - 0.5;
        pb_proj dst = {PB_KIND_CAMERA, PB_LENS_EQUISOLID, 768, 768, fov, mag, mag / (2.0 * sin(fov / 2.0 / 2.0))}; /* lens.py:240-243 */
        pb_proj src = {PB_KIND_PANO, 0, 1024, 2048, 0.0, 0.0, 0.0};
        double R[18];
        rotation_matrix(30.0 / 180.0 * PI, 45.0 / 180.0 * PI, 10.0 / 180.0 * PI, R);
        rotation_matrix(-5.0 / 180.0 * PI, 0.0, 77.0 / 180.0 * PI, R + 9);
        bad |= run_case("photo_from_pano", &dst, R, 2, &src, 0);
    }
    {   /* make-pano --type double: 960x1920 Gear-360-like frame (2 x 195 degrees) -> 1024x2048 panorama */
        const double fov = 195.0 / 180.0 * PI, mag = 960 / 2.0;
        pb_proj src = {PB_KIND_DOUBLE, PB_LENS_EQUIDISTANT, 960, 1920, fov, mag, mag / (fov / 2.0)}; /* lens.py:187 */
        pb_proj dst = {PB_KIND_PANO, 0, 1024, 2048, 0.0, 0.0, 0.0};
        bad |= run_case("stitch_195", &dst, NULL, 0, &src, 2);
    }
    CHECK(pb_shutdown());
    if (bad) return fprintf(stderr, "c host FAILED (mask %d)\n", bad), 1;
    printf("c host ok\n");
    return 0;
}
