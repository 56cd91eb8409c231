/* remap_host.c - a plain C host of the C ABI (no Python, no torch, no HIP headers): what a maintainer binding
 * include/photonbend_hip.h from a compiled language writes.  Builds a plan for a rotated fisheye view of a panorama,
 * remaps a batch of synthetic frames, and checks the bytes three ways: the fast path against the faithful float64 kernel,
 * against a gather through pb_index_map_i32, and frame k of the batch against a single-frame launch.  The same for a
 * double-fisheye stitch.  Then PARITY: two geometries of tests/golden/ - a mid-size identity remap through a rotation and BASELINE
 * config 2 at full size - whose output's SHA-256 must be the reference's own (run_pinned).  Exit code 0 and "c host ok" on success.
 *   gcc -std=c99 -O1 -Iinclude tests/c_host/remap_host.c -Lphotonbend_amd -lphotonbend_hip -Wl,-rpath,$PWD/photonbend_amd -lm
 * Reference call sequence reproduced: core/__init__.py:66-92. */
#include <math.h>
#include <stdint.h>
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

/* SHA-256 (FIPS 180-4), written here so that the program needs nothing but libc */
static void sha256_hex(const uint8_t* data, size_t len, char hex[65]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
        0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
        0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
        0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
        0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
        0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const size_t total = ((len + 8) / 64 + 1) * 64;
    uint8_t tail[128];
    const size_t body = len - len % 64, tail_len = total - body;
    memset(tail, 0, sizeof tail);
    memcpy(tail, data + body, len - body);
    tail[len - body] = 0x80;
    for (int i = 0; i < 8; ++i) tail[tail_len - 1 - i] = (uint8_t)(((uint64_t)len * 8) >> (8 * i));
    for (size_t off = 0; off < total; off += 64) {
        const uint8_t* blk = off < body ? data + off : tail + (off - body);
        uint32_t w[64], a[8];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)blk[4 * i] << 24 | (uint32_t)blk[4 * i + 1] << 16 | (uint32_t)blk[4 * i + 2] << 8 | blk[4 * i + 3];
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
        for (int i = 16; i < 64; ++i)
            w[i] = w[i - 16] + (ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10));
        memcpy(a, h, sizeof a);
        for (int i = 0; i < 64; ++i) {
            const uint32_t t1 = a[7] + (ROR(a[4], 6) ^ ROR(a[4], 11) ^ ROR(a[4], 25)) + ((a[4] & a[5]) ^ (~a[4] & a[6])) + K[i] + w[i];
            const uint32_t t2 = (ROR(a[0], 2) ^ ROR(a[0], 13) ^ ROR(a[0], 22)) + ((a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]));
            a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
        }
#undef ROR
        for (int i = 0; i < 8; ++i) h[i] += a[i];
    }
    for (int i = 0; i < 8; ++i) sprintf(hex + 8 * i, "%08x", h[i]);
}

/* PARITY from a compiled language: one frame through the fast path, its SHA-256 against the REFERENCE's (the json files under tests/golden:
 * u8_sha256 is the hash of photonbend v1.0.1's own output for the geometry on the synthetic frame whose hash is frame_sha256;
 * tests/test_c_host.py checks that the strings below are the JSON's). */
static int run_pinned(const char* name, const pb_proj* dst, const double* rot, int n_rot, const pb_proj* src, int mask, const char* frame_sha,
                      const char* u8_sha) {
    const size_t sb = (size_t)3 * src->height * src->width, db = (size_t)3 * dst->height * dst->width;
    pb_plan* plan = NULL;
    void *s_dev = NULL, *d_dev = NULL, *stream = NULL;
    CHECK(pb_stream_create(&stream));
    CHECK(pb_plan_create(dst, rot, n_rot, src, &plan));
    CHECK(pb_malloc(&s_dev, sb));
    CHECK(pb_malloc(&d_dev, db));
    CHECK(pb_synth_frame_u8(s_dev, src->height, src->width, 0u, 0u, mask, stream));
    CHECK(pb_remap_u8(plan, s_dev, d_dev, 1, 0, 0, stream));
    uint8_t *h_src = malloc(sb), *h_dst = malloc(db);
    CHECK(pb_memcpy_d2h(h_src, s_dev, sb, stream));
    CHECK(pb_memcpy_d2h(h_dst, d_dev, db, stream));
    CHECK(pb_stream_sync(stream));
    char got_src[65], got_dst[65];
    sha256_hex(h_src, sb, got_src);
    sha256_hex(h_dst, db, got_dst);
    const int bad = (strcmp(got_src, frame_sha) ? 32 : 0) | (strcmp(got_dst, u8_sha) ? 64 : 0);
    printf("%-18s %dx%d <- %dx%d: input frame %s the golden's, output %s the reference's (sha256 %.16s...)\n", name, dst->height, dst->width,
           src->height, src->width, (bad & 32) ? "!=" : "==", (bad & 64) ? "!=" : "==", got_dst);
    free(h_src); free(h_dst);
    pb_free(s_dev); pb_free(d_dev);
    pb_plan_destroy(plan);
    pb_stream_destroy(stream);
    return bad;
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

#ifdef SHA_SELFTEST /* tests/test_c_host.py: the SHA-256 routine alone, no library */
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    static uint8_t buf[1 << 20];
    const size_t n = fread(buf, 1, sizeof buf, f);
    fclose(f);
    char hex[65];
    sha256_hex(buf, n, hex);
    puts(hex);
    (void)run_case; (void)run_pinned; (void)rotation_matrix;
    return 0;
}
#else
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
    {   /* tests/golden/mid.json: M_ident_eqd_rot0 - a 768 x 768 equidistant-180 fisheye onto itself through Rotation(0, 0, 0): every
         * pre-truncation coordinate sits on an integer, the texel follows the last bit of the chain's sin / cos / arccos / atan2 */
        const double fov = 180.0 / 180.0 * PI, mag = 768 / 2.0 - 0.5;
        pb_proj cam = {PB_KIND_CAMERA, PB_LENS_EQUIDISTANT, 768, 768, fov, mag, mag / (fov / 2.0)}; /* lens.py:187 */
        double R[9];
        rotation_matrix(0.0, 0.0, 0.0, R);
        bad |= run_pinned("M_ident_eqd_rot0", &cam, R, 1, &cam, 1, "a2dc973a60c97def9bce66fdc0faf1edb5723ad033fadf190ee89b905f50d39b",
                          "f664651d8dbc3f6918febc3ae25aed071d1e5dbb6f5fc4b393952c435c946cfc");
    }
    {   /* tests/golden/full.json: c2 - BASELINE config 2 at full size, 8192 x 4096 panorama -> 4096 x 4096 equidistant-360 inscribed */
        const double fov = 360.0 / 180.0 * PI, mag = 4096 / 2.0 - 0.5;
        pb_proj dst = {PB_KIND_CAMERA, PB_LENS_EQUIDISTANT, 4096, 4096, fov, mag, mag / (fov / 2.0)};
        pb_proj src = {PB_KIND_PANO, 0, 4096, 8192, 0.0, 0.0, 0.0};
        bad |= run_pinned("c2", &dst, NULL, 0, &src, 0, "7355fc4889a0e46062ea6c7c0bd97bee781ba7c6d030bfadbb2e7f0934442bad",
                          "0d2149b6d8ed32375471d6e04a169a4dd15bedc7594b31b55f14e3b6d6b9b874");
    }
    CHECK(pb_shutdown());
    if (bad) return fprintf(stderr, "c host FAILED (mask %d)\n", bad), 1;
    printf("c host ok\n");
    return 0;
}
#endif
