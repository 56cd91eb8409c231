#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY - golden-vector generator.

Runs ONLY in the build container: it imports the real reference read-only from
/root/reference (never copied, never shipped), pushes the shared case matrix
(tests/cases.py) through the reference's public API and writes small fixtures
under tests/golden/:

  lens.npz        G1  forward/inverse of the six lenses on a fixed grid (f64 bits)
  small.npz       G2-G6  per case: f_distance bits, rotation matrices, integer
                  source-index map, uint8 output on the synthetic frame, fragile
                  mask, and (for keep_map cases) the float64 coordinate maps
  mid.json        1-2 k pixel pins for the lenses / degenerate geometries the BASELINE configs do not touch
  full.json       G7  full-size pins for the BASELINE configs: SHA-256 of the
                  index map and of the uint8 output, 65 536 seeded samples,
                  valid-pixel and distinct-texel counts

Usage:  python oracle/make_goldens.py [--small] [--full] [--lens]
(no flag = everything; --full needs ~10 GB of RAM and a few minutes).
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import warnings

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

from oracle import reference_path as orc  # noqa: E402
from oracle.synth import synth_frame, synth_image  # noqa: E402
from tests import cases as tc  # noqa: E402
from tests.cases import Case, full_cases, mid_cases, small_cases  # noqa: E402

import photonbend.core.lens as ref_lens  # noqa: E402
from photonbend.core.projection import CameraImage, DoubleCameraImage, PanoramaImage  # noqa: E402
from photonbend.core.rotation import Rotation  # noqa: E402
from photonbend.utils import to_radians  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
warnings.simplefilter("ignore")
np.seterr(all="ignore")


def ref_lens_obj(lens):
    if lens == "custom":
        return ref_lens.Lens(tc.custom_forward, tc.custom_reverse)
    if lens == "thobylike":
        return ref_lens.Lens(tc.thoby_like_forward, tc.thoby_like_reverse)
    return getattr(ref_lens, lens)()


def ref_obj(p, image=None):
    kind, h, w, lens, fov, mag = p
    if image is None:
        image = np.zeros((h, w, 3), np.uint8)
    if kind == "pano":
        return PanoramaImage(image)
    L = ref_lens_obj(lens)
    if kind == "camera":
        return CameraImage(image, to_radians(fov), L, magnitude=mag)
    return DoubleCameraImage(image, to_radians(fov), L)


def orc_proj(p):
    kind, h, w, lens, fov, mag = p
    if kind == "pano":
        return orc.Proj("pano", h, w)
    return orc.Proj(kind, h, w, lens, to_radians(fov), mag)


def ref_map(case: Case):
    dst = ref_obj(case.dst)
    m = dst.get_coordinate_map()
    stages = [np.copy(m)]
    mats = []
    for rot in case.rotations:
        r = Rotation(*map(to_radians, rot))
        mats.append(np.copy(r.rotation_matrix))
        m = r.rotate_coordinate_map(m)
        stages.append(np.copy(m))
    return dst, m, stages, mats


def ref_index(case: Case, cmap):
    """Integer source-index map straight from the reference: feed it an int32
    'image' whose pixel value is its own linear index + 1 (the reference only
    fancy-indexes and zeroes, so any dtype passes through)."""
    kind, h, w, lens, fov, mag = case.src
    ids = (np.arange(h * w, dtype=np.int32) + 1).reshape(h, w)
    if kind != "double":
        src = ref_obj(case.src, ids)
        return (src.process_coordinate_map(np.copy(cmap)) - 1).astype(np.int32)
    # the two CameraImage objects DoubleCameraImage builds internally, made here
    # through the public constructor on the same halves
    L = getattr(ref_lens, lens)()
    w2 = w // 2
    left = CameraImage(ids[:, :w2], to_radians(fov), L)
    right = CameraImage(np.copy(ids[:, w2:])[:, ::-1], to_radians(fov), L)
    rmap = np.copy(cmap)
    rmap[:, :, 0] *= -1
    rmap[:, :, 0] += np.pi
    il = left.process_coordinate_map(np.copy(cmap)) - 1
    ir = right.process_coordinate_map(rmap) - 1
    return il.astype(np.int32), ir.astype(np.int32)


def gen_full_raw():
    """Adds to full.json, for the double-fisheye configs, the reference's output on the UNMASKED synthetic frame (both
    eyes' taps carry data everywhere: outside the circles the two samples ADD and wrap mod 256, projection.py:439-460) -
    an index error of either eye that black-on-black would hide changes these bytes.  The other pins are left as they are."""
    path = os.path.join(GOLD, "full.json")
    with open(path) as f:
        pins = json.load(f)
    for case in full_cases():
        if case.src[0] != "double":
            continue
        _, cmap, _, _ = ref_map(case)
        kind, h, w, *_ = case.src
        frame = synth_frame(h, w, frame=0, seed=0, circle_mask=0)
        src = ref_obj(case.src, frame)
        u8 = src.process_coordinate_map(np.copy(cmap))
        want = orc.remap(orc_proj(case.dst), orc_proj(case.src), frame, [tuple(map(to_radians, r)) for r in case.rotations])
        assert np.array_equal(u8, want), f"{case.name} raw: oracle != reference"
        H, W = u8.shape[:2]
        pos = np.random.default_rng(pins[case.name]["sample_seed"]).integers(0, H * W, size=65536)
        pins[case.name]["raw_frame_sha256"] = sha(frame)
        pins[case.name]["raw_u8_sha256"] = sha(u8)
        pins[case.name]["raw_u8_samples"] = [int(v) for v in u8.reshape(-1, 3)[pos[:2048]].ravel()]
        pins[case.name]["raw_nonzero_bytes"] = int(np.count_nonzero(u8))
        print(f"  {case.name} raw: {pins[case.name]['raw_nonzero_bytes']} non-zero bytes, sha {pins[case.name]['raw_u8_sha256'][:16]}")
        del cmap, u8, frame
    with open(path, "w") as f:
        json.dump(pins, f, indent=1)
    print("full.json updated (raw pins of the double-fisheye configs)")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def gen_lens():
    grid = np.array(
        [0.0, 1e-300, 1e-12, 0.1, 0.5, 0.7071067811865476, 1.0, 1.2, 1.47, 1.5, 2.0, 2.0000000000000004, 2.5, 3.0, 7.0, 100.0]
        + [to_radians(d) / 2 for d in (140, 178, 180, 195, 360)]
        + list(np.linspace(0.01, 3.2, 40)),
        dtype=np.float64,
    )
    out = {"grid": bits(grid)}
    for name in orc.LENSES:
        L = getattr(ref_lens, name)()
        out[f"{name}_fwd"] = bits(L.forward_function(np.copy(grid)))
        out[f"{name}_inv"] = bits(L.reverse_function(np.copy(grid)))
    np.savez_compressed(os.path.join(GOLD, "lens.npz"), **out)
    print("lens.npz written")


def gen_small():
    out = {}
    for case in small_cases():
        dst, cmap, stages, mats = ref_map(case)
        kind, h, w, *_ = case.src
        frame = synth_frame(h, w, frame=0, seed=0, circle_mask=case.mask)
        src = ref_obj(case.src, frame)
        n = case.name
        if case.dst[0] != "pano":
            out[f"{n}/dst_f"] = bits(np.array([dst.f_distance]))
        if case.src[0] != "pano":
            out[f"{n}/src_f"] = bits(np.array([src.f_distance]))
        if mats:
            out[f"{n}/R"] = bits(np.stack(mats))
        idx = ref_index(case, cmap)
        u8 = src.process_coordinate_map(np.copy(cmap))
        # oracle restatement on the same case: must agree bit for bit, then it
        # supplies what the reference keeps in local variables
        od, os_ = orc_proj(case.dst), orc_proj(case.src)
        rots = [tuple(map(to_radians, r)) for r in case.rotations]
        oidx = orc.remap_index(od, os_, rots)
        if kind == "double":
            assert np.array_equal(oidx[0], idx[0]) and np.array_equal(oidx[1], idx[1]), n
            out[f"{n}/idx_l"], out[f"{n}/idx_r"] = idx
            out[f"{n}/w_l"], out[f"{n}/w_r"] = bits(oidx[2]), bits(oidx[3])
        else:
            assert np.array_equal(oidx, idx), n
            out[f"{n}/idx"] = idx
        assert np.array_equal(orc.remap(od, os_, frame, rots), u8), n
        out[f"{n}/u8"] = u8
        out[f"{n}/fragile"] = np.packbits(orc.fragile_mask(orc.pretrunc(od, os_, rots)))
        if case.keep_map:
            for k, st in enumerate(stages):
                out[f"{n}/map{k}"] = bits(st)
        print(f"  {n}: ok ({u8.shape[0]}x{u8.shape[1]})")
    np.savez_compressed(os.path.join(GOLD, "small.npz"), **out)
    print("small.npz written,", len(out), "arrays")


def gen_mapproj():
    """f-3: map_projection (projection.py:550-599) on the float64 maps of the keep_map cases."""
    from photonbend.core.projection import map_projection

    out = {}
    for case in small_cases():
        if not case.keep_map:
            continue
        _, m, _, _ = ref_map(case)
        given = np.copy(m)
        img = map_projection(given)
        assert np.array_equal(orc.map_projection(np.copy(m)), img), case.name
        # the input is small.npz's last map of the case; the reference zeroes invalid lat/lon in it
        expect = np.copy(m)
        expect[:, :, :2][expect[:, :, 2] != 0.0] = 0
        assert np.array_equal(bits(given), bits(expect)), case.name
        out[f"{case.name}/out"] = img
    np.savez_compressed(os.path.join(GOLD, "mapproj.npz"), **out)
    print("mapproj.npz written,", len(out), "cases")


def gen_generic():
    """Images beyond uint8 RGB, user lenses, more than eight rotations, odd-width double frames: the REFERENCE's
    outputs for tests/cases.py generic_cases (it fancy-indexes whatever array it is given, projection.py:234-243)."""
    out = {}
    for name, case, layout in tc.generic_cases():
        _, cmap, _, _ = ref_map(case)
        kind, h, w, *_ = case.src
        img = synth_image(h, w, layout, frame=3, circle_mask=case.mask)
        res = ref_obj(case.src, img).process_coordinate_map(np.copy(cmap))
        out[name] = res
        print(f"  {name}: {img.dtype}{img.shape} -> {res.dtype}{res.shape}")
    # utils.calculate_size_panorama_to_photo (utils/__init__.py:81-118)
    from photonbend.utils import calculate_size_panorama_to_photo

    sizes = []
    for lname in ("equidistant", "equisolid", "stereographic", "orthographic", "thoby"):
        for wh in ((8192, 4096), (6144, 3072), (1000, 500), (2, 1)):
            for vert in (False, True):
                sizes.append((lname, wh[0], wh[1], int(vert), *calculate_size_panorama_to_photo(wh, getattr(ref_lens, lname)().forward_function, vert)))
    with open(os.path.join(GOLD, "size_rule.json"), "w") as f:  # Python ints: the orthographic vertical rule overflows int64
        json.dump([[r[0], *map(int, r[1:])] for r in sizes], f)
    np.savez_compressed(os.path.join(GOLD, "generic.npz"), **out)
    print("generic.npz written,", len(out), "arrays")


def gen_cli():
    """f-2: the reference's own CLI (click CliRunner) on small synthetic PNGs -> output pixel arrays."""
    import tempfile

    from click.testing import CliRunner
    from PIL import Image
    from photonbend.scripts.main import main as ref_main
    from tests.cases import cli_cases

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, cmd, opts, spec in cli_cases():
            h, w, mask, layout = (*spec, "RGB")[:4]
            inp, outp = os.path.join(tmp, name + "_in.png"), os.path.join(tmp, name + "_out.png")
            Image.fromarray(synth_image(h, w, layout, frame=5, circle_mask=mask)).save(inp)
            # argument order differs per command only in where OUTPUT sits; click accepts options anywhere
            res = CliRunner().invoke(ref_main, [cmd, inp, *opts, outp])
            if res.exception is not None and not isinstance(res.exception, SystemExit):
                # the reference CLI itself rejects the input (grey images: "height, width, _ = shape"): pin the exception type
                out[name] = np.array("raises:" + type(res.exception).__name__)
                print(f"  {name}: reference raises {type(res.exception).__name__}: {res.exception}")
                continue
            assert res.exit_code == 0, (name, res.output, res.exception)
            out[name] = np.asarray(Image.open(outp))
            print(f"  {name}: {out[name].dtype}{out[name].shape}")
    np.savez_compressed(os.path.join(GOLD, "cli.npz"), **out)
    print("cli.npz written,", len(out), "cases")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gen_real():
    """G8: BASELINE config c1 on the reference's real image (examples/equidistant.jpg stays in the reference;
    only hashes and samples of the reference's OUTPUT are kept)."""
    from PIL import Image

    img = np.asarray(Image.open("/root/reference/examples/equidistant.jpg"))
    h = img.shape[0]
    src = CameraImage(img, to_radians(360), ref_lens.equidistant(), magnitude=img.shape[1] / 2 - 0.5)
    dst = PanoramaImage(np.zeros((2048, 4096, 3), np.uint8))
    out = src.process_coordinate_map(dst.get_coordinate_map())
    pos = np.random.default_rng(2024).integers(0, 2048 * 4096, size=4096)
    pin = {
        "input": "examples/equidistant.jpg",
        "input_shape": list(img.shape),
        "input_sha256": sha(img),
        "u8_sha256": sha(out),
        "sample_seed": 2024,
        "u8_samples": [int(v) for v in out.reshape(-1, 3)[pos].ravel()],
    }
    with open(os.path.join(GOLD, "c1_real.json"), "w") as f:
        json.dump(pin, f)
    print("c1_real.json written", out.shape, h)


def gen_mid():
    """1-2 k pixel pins (tests/cases.py mid_cases): hashes, samples and counts from the REAL reference, the oracle
    asserted equal on every pixel, and the size of the fragile set (pre-truncation coordinate within 2^-40 of an
    integer) - the only pixels where a last-bit libm difference may legitimately flip a truncation."""
    pins = {}
    for case in mid_cases():
        dst, cmap, _, mats = ref_map(case)
        kind, h, w, *_ = case.src
        frame = synth_frame(h, w, frame=0, seed=0, circle_mask=case.mask)
        src = ref_obj(case.src, frame)
        idx = ref_index(case, cmap)
        u8 = src.process_coordinate_map(np.copy(cmap))
        od, os_ = orc_proj(case.dst), orc_proj(case.src)
        rots = [tuple(map(to_radians, r)) for r in case.rotations]
        assert np.array_equal(orc.remap_index(od, os_, rots), idx), case.name
        assert np.array_equal(orc.remap(od, os_, frame, rots), u8), case.name
        fragile = orc.fragile_mask(orc.pretrunc(od, os_, rots))
        H, W = u8.shape[:2]
        pos = np.random.default_rng(777).integers(0, H * W, size=4096)
        pins[case.name] = {
            "dst": list(case.dst), "src": list(case.src), "rotations": [list(r) for r in case.rotations], "mask": case.mask,
            "idx_sha256": sha(idx), "u8_sha256": sha(u8), "frame_sha256": sha(frame), "sample_seed": 777,
            "idx_samples": [int(v) for v in idx.ravel()[pos[:1024]]],
            "u8_samples": [int(v) for v in u8.reshape(-1, 3)[pos[:1024]].ravel()],
            "in_bounds_samples": int((idx >= 0).sum()),
            "fragile_pixels": int(fragile.sum()),
            "fragile_sha256": sha(np.packbits(fragile)),
        }
        print(f"  {case.name}: {H}x{W}, in-bounds {pins[case.name]['in_bounds_samples']}, fragile {pins[case.name]['fragile_pixels']}")
    with open(os.path.join(GOLD, "mid.json"), "w") as f:
        json.dump(pins, f, indent=1)
    print("mid.json written")


def gen_full():
    pins = {}
    for case in full_cases():
        dst, cmap, _, mats = ref_map(case)
        kind, h, w, *_ = case.src
        frame = synth_frame(h, w, frame=0, seed=0, circle_mask=case.mask)
        src = ref_obj(case.src, frame)
        idx = ref_index(case, cmap)
        u8 = src.process_coordinate_map(np.copy(cmap))
        H, W = u8.shape[:2]
        rng = np.random.default_rng(12345)
        pos = rng.integers(0, H * W, size=65536)
        pin = {
            "dst": list(case.dst),
            "src": list(case.src),
            "rotations": [list(r) for r in case.rotations],
            "mask": case.mask,
            "u8_sha256": sha(u8),
            "frame_sha256": sha(frame),
            "sample_seed": 12345,
            "u8_samples_sha256": sha(u8.reshape(-1, 3)[pos]),
        }
        if case.dst[0] != "pano":
            pin["dst_f_bits"] = int(bits(np.array([dst.f_distance]))[0])
        if case.src[0] != "pano":
            pin["src_f_bits"] = int(bits(np.array([src.f_distance]))[0])
        if mats:
            pin["R_bits"] = [int(v) for v in bits(np.stack(mats)).ravel()]
        if kind == "double":
            il, ir = idx
            pin["idx_l_sha256"], pin["idx_r_sha256"] = sha(il), sha(ir)
            pin["idx_l_samples"] = [int(v) for v in il.ravel()[pos[:2048]]]
            pin["idx_r_samples"] = [int(v) for v in ir.ravel()[pos[:2048]]]
            pin["valid_left"] = int((il >= 0).sum())
            pin["valid_right"] = int((ir >= 0).sum())
            pin["in_bounds_samples"] = pin["valid_left"] + pin["valid_right"]
        else:
            pin["idx_sha256"] = sha(idx)
            pin["idx_samples"] = [int(v) for v in idx.ravel()[pos[:2048]]]
            pin["in_bounds_samples"] = int((idx >= 0).sum())
            pin["distinct_texels"] = int(np.unique(idx[idx >= 0]).size)
        pin["algorithmic_bytes"] = 3 * H * W + 3 * pin["in_bounds_samples"]
        pin["u8_samples"] = [int(v) for v in u8.reshape(-1, 3)[pos[:2048]].ravel()]
        pins[case.name] = pin
        print(f"  {case.name}: in-bounds {pin['in_bounds_samples']}, algorithmic bytes {pin['algorithmic_bytes']}")
        del cmap, idx, u8, frame
    with open(os.path.join(GOLD, "full.json"), "w") as f:
        json.dump(pins, f, indent=1)
    print("full.json written")


def bilinear_crops(case: Case):
    """Two 128 x 128 windows of a full-size output kept whole next to the seeded samples: the image centre (a pole / the fisheye
    centre / the stitch's seam column) and a place on the rim of what the geometry paints - localized errors that 65 536 random
    samples of 16-33 M pixels would miss."""
    H, W = case.dst[1], case.dst[2]
    centre = (H // 2 - 64, W // 2 - 64)
    if case.dst[0] == "pano":
        rim = (H // 2 - 64, W // 4 - 64) if case.src[0] == "double" else (H - 128, W // 3)  # an eye's rim (the blend band) / the pole rows
    else:
        rim = (H // 2 - 64, 0)  # where the image circle touches the frame's left edge
    return {"centre": centre, "rim": rim}


def gen_full_bilinear():
    """f-4 at the sizes it is benchmarked at: for the five BASELINE geometries on the synthetic (noise) frame,
    65 536 seeded sample values of oracle.remap_bilinear - OUR definition of the opt-in mode; the reference has no bilinear
    behaviour, so these pin the HIP kernels to the written definition, not to the reference ("parity unpinned") - plus counts and
    two 128 x 128 crops.  Samples and crops go to full_bilinear.npz, counts and positions to full.json under "bilinear"."""
    path = os.path.join(GOLD, "full.json")
    with open(path) as f:
        pins = json.load(f)
    arrays = {}
    for case in full_cases():
        od, os_ = orc_proj(case.dst), orc_proj(case.src)
        rots = [tuple(map(to_radians, r)) for r in case.rotations]
        kind, h, w, *_ = case.src
        frame = synth_frame(h, w, frame=0, seed=0, circle_mask=case.mask)
        assert sha(frame) == pins[case.name]["frame_sha256"]
        out = orc.remap_bilinear(od, os_, frame, rots)
        H, W = out.shape[:2]
        pos = np.random.default_rng(pins[case.name]["sample_seed"]).integers(0, H * W, size=65536)
        flat = out.reshape(-1, 3)
        arrays[f"{case.name}/samples"] = flat[pos].copy()
        crops = bilinear_crops(case)
        for tag, (r0, c0) in crops.items():
            arrays[f"{case.name}/crop_{tag}"] = out[r0:r0 + 128, c0:c0 + 128].copy()
        black = (out == 0).all(axis=2)
        pins[case.name]["bilinear"] = {
            "definition": "oracle/reference_path.py:remap_bilinear (no reference behaviour: parity unpinned)",
            "samples_sha256": sha(arrays[f"{case.name}/samples"]),
            "black_pixels": int(black.sum()),
            "byte_sum": int(out.astype(np.uint64).sum()),
            "crops": {k: list(v) for k, v in crops.items()},
        }
        print(f"  {case.name} bilinear: {pins[case.name]['bilinear']['black_pixels']} black pixels, byte sum {pins[case.name]['bilinear']['byte_sum']}")
        del out, frame, flat, black
    np.savez_compressed(os.path.join(GOLD, "full_bilinear.npz"), **arrays)
    with open(path, "w") as f:
        json.dump(pins, f, indent=1)
    print("full_bilinear.npz written; full.json updated (bilinear counts)")


def gen_npmath():
    """The RESULT BITS of this container's NumPy - the NumPy that produced every other golden - for the transcendental calls the reference
    makes (rotation.py:129-164, lens.py:71-335, projection.py:193, :252), on the arguments of tests/npmath_args.py: np.arcsin / arccos /
    arctan / tan (NumPy's own AVX-512 kernels), np.sin / np.cos (glibc, `_fma` build), np.exp(x * 1j) (glibc's internal sincos, plain
    build) and np.log(z).imag (glibc atan2).  photonbend_amd/csrc/pb_math_np.hpp and pb_math_glibc.hpp restate them.  Scalars, strided
    views and short arrays go through the same kernels as whole arrays (checked here), so one table per function pins every use."""
    from tests import npmath_args

    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
    except ImportError:
        from numpy.core._multiarray_umath import __cpu_features__ as feats
    import ctypes

    libc = ctypes.CDLL(None)
    libc.gnu_get_libc_version.restype = ctypes.c_char_p
    arrays = {}
    for fn in npmath_args.FUNCTIONS:
        x = npmath_args.arguments(fn)
        with np.errstate(all="ignore"):
            y = npmath_args.reference(fn, x)
            if fn not in ("expi", "arg"):
                f = getattr(np, fn)
                for i in range(0, x.size, 997):
                    assert np.float64(f(float(x[i]))).tobytes() == y[i].tobytes()
                assert f(x[::3]).tobytes() == y[::3].tobytes() and f(x[5:8]).tobytes() == y[5:8].tobytes()
            elif fn == "expi":
                assert npmath_args.reference(fn, x[7:10]).tobytes() == y[14:20].tobytes()
            else:
                assert npmath_args.reference(fn, x[14:20]).tobytes() == y[7:10].tobytes()
        arrays[fn] = y
    meta = {"numpy": np.__version__, "glibc": libc.gnu_get_libc_version().decode(), "AVX512_SKX": bool(feats.get("AVX512_SKX")),
            "FMA3": bool(feats.get("FMA3")), "arguments": "tests/npmath_args.py", "n": int(npmath_args.N_PER_FUNCTION)}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, "npmath.npz"), **arrays)
    print("npmath.npz written:", meta)


NO_AVX512 = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX512_KNL AVX512_KNM"


def gen_npmath_libm():
    """The SECOND math flavour's pins: NumPy's result bits for np.arcsin / arccos / arctan / tan on the arguments of
    tests/npmath_args.py when NumPy runs WITHOUT its AVX-512 kernels (NPY_DISABLE_CPU_FEATURES: what every x86-64 host without AVX512_SKX
    does) - there it calls libm, here glibc 2.35's asin / acos / atan / tan (`_fma` builds).  np.sin / np.cos / np.exp(1j x) / np.log(z).imag
    do not change (asserted equal to npmath.npz).  Written by a child process, since the dispatch is fixed when NumPy is imported."""
    import subprocess

    if os.environ.get("PB_NPMATH_LIBM_CHILD") != "1":
        env = dict(os.environ, NPY_DISABLE_CPU_FEATURES=NO_AVX512, PB_NPMATH_LIBM_CHILD="1")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--npmath-libm"], env=env, check=True)
        return
    from tests import npmath_args

    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
    except ImportError:
        from numpy.core._multiarray_umath import __cpu_features__ as feats
    assert not feats.get("AVX512_SKX") and not feats.get("AVX512F"), "the AVX-512 kernels are still dispatched"
    import ctypes

    libc = ctypes.CDLL(None)
    libc.gnu_get_libc_version.restype = ctypes.c_char_p
    first = np.load(os.path.join(GOLD, "npmath.npz"))
    arrays, share = {}, {}
    for fn in npmath_args.FUNCTIONS:
        x = npmath_args.arguments(fn)
        with np.errstate(all="ignore"):
            y = npmath_args.reference(fn, x)
        if fn in ("arcsin", "arccos", "arctan", "tan"):
            arrays[fn] = y
            nan = np.isnan(y.view(np.float64)) & np.isnan(first[fn].view(np.float64))
            share[fn] = float(((y != first[fn]) & ~nan).mean())
        else:
            assert np.array_equal(y, first[fn]), f"{fn} changed with the dispatch: it should be glibc on both kinds of host"
    meta = {"numpy": np.__version__, "glibc": libc.gnu_get_libc_version().decode(), "AVX512_SKX": False, "FMA3": bool(feats.get("FMA3")),
            "NPY_DISABLE_CPU_FEATURES": NO_AVX512, "arguments": "tests/npmath_args.py", "n": int(npmath_args.N_PER_FUNCTION),
            "share_of_results_differing_from_the_first_flavour": share}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, "npmath_libm.npz"), **arrays)
    print("npmath_libm.npz written:", meta)


def gen_libm_flavour():
    """tests/golden/libm_flavour.json: what the REFERENCE returns for the small and the mid cases on a host without AVX512_SKX (NumPy's
    arcsin / arccos / arctan / tan = libm's) - per case the SHA-256 of every float64 map stage (NaNs canonicalised), of the index map(s)
    and of the output bytes, plus how many index entries differ from the first flavour's goldens.  A child process under
    NPY_DISABLE_CPU_FEATURES (the dispatch is fixed at import); the oracle is asserted equal to the reference on every case there too."""
    import subprocess

    if os.environ.get("PB_NPMATH_LIBM_CHILD") != "1":
        env = dict(os.environ, NPY_DISABLE_CPU_FEATURES=NO_AVX512, PB_NPMATH_LIBM_CHILD="1")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--libm-flavour"], env=env, check=True)
        return
    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
    except ImportError:
        from numpy.core._multiarray_umath import __cpu_features__ as feats
    assert not feats.get("AVX512_SKX") and not feats.get("AVX512F"), "the AVX-512 kernels are still dispatched"
    small = np.load(os.path.join(GOLD, "small.npz"))
    mid = json.load(open(os.path.join(GOLD, "mid.json")))
    out = {}
    for case in small_cases() + mid_cases():
        dst, cmap, stages, mats = ref_map(case)
        kind, h, w, *_ = case.src
        frame = synth_frame(h, w, frame=0, seed=0, circle_mask=case.mask)
        src = ref_obj(case.src, frame)
        idx = ref_index(case, cmap)
        u8 = src.process_coordinate_map(np.copy(cmap))
        od, os_ = orc_proj(case.dst), orc_proj(case.src)
        rots = [tuple(map(to_radians, r)) for r in case.rotations]
        oidx = orc.remap_index(od, os_, rots)
        n = case.name
        rec = {"map_sha256": [canonical_map_sha(st) for st in stages], "u8_sha256": sha(u8)}
        if kind == "double":
            assert np.array_equal(oidx[0], idx[0]) and np.array_equal(oidx[1], idx[1]), n
            rec["idx_l_sha256"], rec["idx_r_sha256"] = sha(np.ascontiguousarray(idx[0], dtype=np.int32)), sha(np.ascontiguousarray(idx[1], dtype=np.int32))
            if f"{n}/idx_l" in small.files:
                rec["index_entries_differing_from_first_flavour"] = int((small[f"{n}/idx_l"] != idx[0]).sum() + (small[f"{n}/idx_r"] != idx[1]).sum())
        else:
            assert np.array_equal(oidx, idx), n
            rec["idx_sha256"] = sha(np.ascontiguousarray(idx, dtype=np.int32))
            if f"{n}/idx" in small.files:
                rec["index_entries_differing_from_first_flavour"] = int((small[f"{n}/idx"] != idx).sum())
            elif n in mid:
                rec["same_index_map_as_first_flavour"] = bool(mid[n]["idx_sha256"] == rec["idx_sha256"])
        assert np.array_equal(orc.remap(od, os_, frame, rots), u8), n
        out[n] = rec
        print(f"  {n}: ok", {k: v for k, v in rec.items() if not k.endswith("sha256")})
    meta = {"numpy": np.__version__, "NPY_DISABLE_CPU_FEATURES": NO_AVX512, "cases": len(out)}
    json.dump({"meta": meta, "cases": out}, open(os.path.join(GOLD, "libm_flavour.json"), "w"), indent=0)
    print("libm_flavour.json written,", len(out), "cases")


def gen_libm_flavour_full(names=None):
    """tests/golden/libm_flavour_full.json: the five BASELINE geometries at FULL size as the REFERENCE computes them on a
    host without AVX512_SKX - SHA-256 of every float64 map stage, of the index map(s) and of the output bytes on the synthetic frame, the
    seeded samples, and whether each equals the first flavour's pin (tests/golden/full.json).  Of the five only c3 can differ: arcsin in
    the equisolid inverse (lens.py:206-220) and arccos in the rotation (rotation.py:158) are the two calls that reach libm there; c1 / c2 /
    c5 run neither and must reproduce the first flavour's hashes.  A child process under NPY_DISABLE_CPU_FEATURES; the oracle is asserted
    equal to the reference on every case there too."""
    import subprocess

    if os.environ.get("PB_NPMATH_LIBM_CHILD") != "1":
        env = dict(os.environ, NPY_DISABLE_CPU_FEATURES=NO_AVX512, PB_NPMATH_LIBM_CHILD="1")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--libm-flavour-full", *(names or [])], env=env, check=True)
        return
    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
    except ImportError:
        from numpy.core._multiarray_umath import __cpu_features__ as feats
    assert not feats.get("AVX512_SKX") and not feats.get("AVX512F"), "the AVX-512 kernels are still dispatched"
    first = json.load(open(os.path.join(GOLD, "full.json")))
    path = os.path.join(GOLD, "libm_flavour_full.json")
    out = json.load(open(path))["cases"] if (names and os.path.exists(path)) else {}
    for case in full_cases():
        if names and case.name not in names:
            continue
        dst, cmap, stages, mats = ref_map(case)
        kind, h, w, *_ = case.src
        frame = synth_frame(h, w, frame=0, seed=0, circle_mask=case.mask)
        src = ref_obj(case.src, frame)
        idx = ref_index(case, cmap)
        u8 = src.process_coordinate_map(np.copy(cmap))
        H, W = u8.shape[:2]
        pos = np.random.default_rng(12345).integers(0, H * W, size=65536)
        od, os_ = orc_proj(case.dst), orc_proj(case.src)
        rots = [tuple(map(to_radians, r)) for r in case.rotations]
        oidx = orc.remap_index(od, os_, rots)
        f = first[case.name]
        rec = {"map_sha256": [canonical_map_sha(st) for st in stages], "u8_sha256": sha(u8), "frame_sha256": sha(frame), "sample_seed": 12345,
               "u8_samples": [int(v) for v in u8.reshape(-1, 3)[pos[:2048]].ravel()]}
        del stages
        if kind == "double":
            assert np.array_equal(oidx[0], idx[0]) and np.array_equal(oidx[1], idx[1]), case.name
            rec["idx_l_sha256"], rec["idx_r_sha256"] = sha(idx[0]), sha(idx[1])
            rec["same_as_first_flavour"] = bool(rec["idx_l_sha256"] == f["idx_l_sha256"] and rec["idx_r_sha256"] == f["idx_r_sha256"] and rec["u8_sha256"] == f["u8_sha256"])
        else:
            assert np.array_equal(oidx, idx), case.name
            rec["idx_sha256"] = sha(idx)
            rec["idx_samples"] = [int(v) for v in idx.ravel()[pos[:2048]]]
            rec["in_bounds_samples"] = int((idx >= 0).sum())
            rec["same_as_first_flavour"] = bool(rec["idx_sha256"] == f["idx_sha256"] and rec["u8_sha256"] == f["u8_sha256"])
        rec["maps_same_as_first_flavour"] = bool(rec["map_sha256"] == f.get("map_sha256"))
        assert rec["frame_sha256"] == f["frame_sha256"]
        del oidx
        assert np.array_equal(orc.remap(od, os_, frame, rots), u8), case.name
        out[case.name] = rec
        print(f"  {case.name}: same index map and bytes as the first flavour: {rec['same_as_first_flavour']}, same float64 maps: {rec['maps_same_as_first_flavour']}", flush=True)
        del cmap, idx, u8, frame
    meta = {"numpy": np.__version__, "NPY_DISABLE_CPU_FEATURES": NO_AVX512, "cases": len(out)}
    json.dump({"meta": meta, "cases": out}, open(path, "w"), indent=0)
    print("libm_flavour_full.json written,", len(out), "cases")


def canonical_map_sha(m):
    """SHA-256 of a float64 coordinate map's bits with every NaN replaced by the one canonical quiet NaN (payloads and signs of NaNs are
    not part of any contract; signed zeros and everything else are)."""
    a = np.ascontiguousarray(m, dtype=np.float64).copy()
    a[np.isnan(a)] = np.float64("nan")
    return sha(a.view(np.uint64))


def gen_map_pins():
    """f-1 at the sizes that matter: the SHA-256 of the reference's float64 coordinate map after get_coordinate_map and after every rotation
    (latitude, longitude and invalid-flag planes, NaNs canonicalised), for the 13 mid cases (every lens, both directions, rotations, 0.5-2 K)
    and the five BASELINE geometries at full size (16.8-33.5 M pixels each) - added to mid.json / full.json under "map_sha256"."""
    for fname, cases in (("mid.json", mid_cases()), ("full.json", full_cases())):
        path = os.path.join(GOLD, fname)
        pins = json.load(open(path))
        for case in cases:
            _, _, stages, _ = ref_map(case)
            pins[case.name]["map_sha256"] = [canonical_map_sha(st) for st in stages]
            pins[case.name]["map_shape"] = list(stages[0].shape)
            print(f"  {case.name}: {len(stages)} stage(s) of {stages[0].shape}")
            del stages
        with open(path, "w") as f:
            json.dump(pins, f, indent=1)
    print("mid.json / full.json updated (map_sha256)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lens", action="store_true")
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--full-raw", action="store_true", help="only add the unmasked-frame pins of the double-fisheye configs to full.json")
    ap.add_argument("--full-bilinear", action="store_true", help="only add the full-size pins of the opt-in bilinear mode (our definition) to full.json / full_bilinear.npz")
    ap.add_argument("--mapproj", action="store_true")
    ap.add_argument("--cli", action="store_true")
    ap.add_argument("--real", action="store_true")
    ap.add_argument("--mid", action="store_true")
    ap.add_argument("--generic", action="store_true")
    ap.add_argument("--npmath", action="store_true", help="NumPy's arcsin / arccos / arctan / tan result bits (tests/golden/npmath.npz)")
    ap.add_argument("--npmath-libm", action="store_true", help="the same four functions as NumPy computes them WITHOUT AVX-512 (libm): tests/golden/npmath_libm.npz")
    ap.add_argument("--libm-flavour", action="store_true", help="map / index / byte hashes of the small and mid cases under the no-AVX-512 dispatch: tests/golden/libm_flavour.json")
    ap.add_argument("--libm-flavour-full", nargs="*", default=None, metavar="CASE", help="the same for the BASELINE geometries at full size (all five, or the named ones): tests/golden/libm_flavour_full.json")
    ap.add_argument("--maps", action="store_true", help="only add the float64 map hashes of the mid and full cases to mid.json / full.json")
    a = ap.parse_args()
    if a.npmath_libm or a.libm_flavour or a.libm_flavour_full is not None:  # (second-flavour fixtures: generated on request only - they re-run this script under another NumPy dispatch)
        if a.npmath_libm:
            gen_npmath_libm()
        if a.libm_flavour:
            gen_libm_flavour()
        if a.libm_flavour_full is not None:
            gen_libm_flavour_full(a.libm_flavour_full)
        sys.exit(0)
    everything = not (a.full_bilinear or a.full_raw or a.lens or a.small or a.full or a.mapproj or a.cli or a.real or a.mid or a.generic or a.npmath or a.maps)
    os.makedirs(GOLD, exist_ok=True)
    if a.lens or everything:
        gen_lens()
    if a.small or everything:
        gen_small()
    if a.mid or everything:
        gen_mid()
    if a.generic or everything:
        gen_generic()
    if a.full or everything:
        gen_full()
    if a.full or a.full_raw or everything:
        gen_full_raw()
    if a.full or a.full_bilinear or everything:
        gen_full_bilinear()
    if a.mapproj or everything:
        gen_mapproj()
    if a.cli or everything:
        gen_cli()
    if a.real or everything:
        gen_real()
    if a.npmath or everything:
        gen_npmath()
    if a.maps or a.mid or a.full or everything:
        gen_map_pins()
