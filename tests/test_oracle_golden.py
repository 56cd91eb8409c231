"""The oracle (oracle/reference_path.py) against fixtures captured from the real
reference (oracle/make_goldens.py).  Bit-for-bit: this is what pins parity."""

import hashlib
import os

import numpy as np
import pytest

from oracle import reference_path as orc
from oracle.synth import synth_frame
from tests import helpers as H
from tests.cases import full_cases, small_cases

SMALL = H.load_small()
FULL = H.load_full()


def test_lens_grid_bits():
    g = np.load(H.GOLD + "/lens.npz")
    grid = g["grid"].view(np.float64)
    with np.errstate(all="ignore"):
        for name in orc.LENSES:
            fwd = orc.lens_forward(name, np.copy(grid))
            inv = orc.lens_inverse(name, np.copy(grid))
            assert np.array_equal(H.bits(fwd), g[f"{name}_fwd"]), name
            assert np.array_equal(H.bits(inv), g[f"{name}_inv"]), name


@pytest.mark.parametrize("case", small_cases(), ids=lambda c: c.name)
def test_small_case_bits(case):
    n = case.name
    od, os_ = H.orc_proj(case.dst), H.orc_proj(case.src)
    rots = H.orc_rots(case)
    if f"{n}/dst_f" in SMALL:
        assert H.bits(np.array([od.f_distance]))[0] == SMALL[f"{n}/dst_f"][0]
    if f"{n}/src_f" in SMALL:
        assert H.bits(np.array([os_.f_distance]))[0] == SMALL[f"{n}/src_f"][0]
    if rots:
        R = np.stack([orc.rotation_matrix(*r) for r in rots])
        assert np.array_equal(H.bits(R), SMALL[f"{n}/R"])
    idx = orc.remap_index(od, os_, rots)
    if case.src[0] == "double":
        assert np.array_equal(idx[0], SMALL[f"{n}/idx_l"])
        assert np.array_equal(idx[1], SMALL[f"{n}/idx_r"])
    else:
        assert np.array_equal(idx, SMALL[f"{n}/idx"])
    out = orc.remap(od, os_, H.case_frame(case), rots)
    assert out.dtype == np.uint8 and np.array_equal(out, SMALL[f"{n}/u8"])
    if case.keep_map:
        cmap = orc.coordinate_map(od)
        assert np.array_equal(H.bits(cmap), SMALL[f"{n}/map0"])
        for k, r in enumerate(rots):
            cmap = orc.rotate_map(orc.rotation_matrix(*r), cmap)
            assert np.array_equal(H.bits(cmap), SMALL[f"{n}/map{k + 1}"])


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _check_full(case):
    pin = FULL[case.name]
    od, os_ = H.orc_proj(case.dst), H.orc_proj(case.src)
    rots = H.orc_rots(case)
    frame = H.case_frame(case)
    assert _sha(frame) == pin["frame_sha256"]
    out = orc.remap(od, os_, frame, rots)
    assert _sha(out) == pin["u8_sha256"]
    idx = orc.remap_index(od, os_, rots)
    if case.src[0] == "double":
        assert _sha(idx[0]) == pin["idx_l_sha256"] and _sha(idx[1]) == pin["idx_r_sha256"]
    else:
        assert _sha(idx) == pin["idx_sha256"]
        assert int((idx >= 0).sum()) == pin["in_bounds_samples"]


def test_full_c2_pins():
    """The headline config at full size (about 15 s of NumPy)."""
    _check_full([c for c in full_cases() if c.name == "c2"][0])


@pytest.mark.slow
@pytest.mark.parametrize("case", [c for c in full_cases() if c.name != "c2"], ids=lambda c: c.name)
def test_full_other_pins(case):
    _check_full(case)


def test_synth_formula_spot_values():
    """The synthetic frame formula restated with Python ints."""
    img = synth_frame(5, 7, frame=3, seed=9)

    def mix(h):
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        h ^= h >> 16
        return h

    for r, c, ch in [(0, 0, 0), (4, 6, 2), (2, 3, 1)]:
        key = ((3 * 0x9E3779B1) ^ (r * 0x85EBCA6B) ^ (c * 0xC2B2AE35) ^ (ch * 0x27D4EB2F) ^ 9) & 0xFFFFFFFF
        assert img[r, c, ch] == mix(key) & 0xFF


def test_map_projection_matches_reference():
    g = np.load(H.GOLD + "/mapproj.npz")
    for case in small_cases():
        if not case.keep_map:
            continue
        shape = (case.dst[1], case.dst[2], 3)
        m = np.array(SMALL[f"{case.name}/map{len(case.rotations)}"].view(np.float64).reshape(shape))
        assert np.array_equal(orc.map_projection(m), g[f"{case.name}/out"]), case.name


def test_c1_real_image_pin():
    """G8: config c1 on the reference's own example image.  The JPEG never leaves /root/reference, so this runs
    only where the reference is mounted (the build container); the GPU box exercises c1 on a synthetic frame."""
    import json
    import os

    path = "/root/reference/examples/equidistant.jpg"
    if not os.path.exists(path):
        pytest.skip("reference examples not mounted")
    from PIL import Image

    pin = json.load(open(H.GOLD + "/c1_real.json"))
    img = np.asarray(Image.open(path))
    assert list(img.shape) == pin["input_shape"] and _sha(img) == pin["input_sha256"]
    src = orc.Proj("camera", img.shape[0], img.shape[1], "equidistant", orc.to_radians(360), img.shape[1] / 2 - 0.5)
    out = orc.remap(orc.Proj("pano", 2048, 4096), src, img)
    assert _sha(out) == pin["u8_sha256"]


def test_device_math_agrees_with_glibc(tmp_path):
    """photonbend_amd/csrc/pb_math.hpp (the sin / cos / atan2 / atan of the faithful device chain) compiled for the HOST: every
    result is the correctly rounded one (113-bit libquadmath reference) and therefore equals this machine's glibc - what the
    reference reaches through NumPy - wherever glibc is itself correctly rounded (all but ~1 argument in 1000).  Both steps of
    the evaluation are exercised: the table-driven fast path and, where it leaves the rounding undecided, the double-double series."""
    import re
    import shutil
    import subprocess

    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_math")
    res = subprocess.run([gxx, "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(root, "oracle", "check_math.cpp"), "-lquadmath"],
                         capture_output=True, text=True)
    if res.returncode != 0 and "quadmath" in res.stderr:
        pytest.skip("libquadmath missing")
    assert res.returncode == 0, res.stderr
    out = subprocess.run([exe, "400000"], capture_output=True, text=True, timeout=300).stdout
    rows = re.findall(r"(\w+)\s+n=\d+ values=(\d+)\s+vs_glibc=(\d+)\s+vs_correctly_rounded=(\d+)\s+glibc_vs_correctly_rounded=(\d+)", out)
    assert [r[0] for r in rows] == ["sincos", "atan2", "atan"], out
    for name, n, vs_glibc, vs_cr, glibc_cr in rows:
        assert int(vs_cr) == 0, f"{name}: {vs_cr} of {n} results are not correctly rounded"
        assert int(vs_glibc) == int(glibc_cr) and int(vs_glibc) <= int(n) * 3 // 1000, f"{name}: {vs_glibc} of {n} differ from glibc"
    assert "special values: 0 mismatches" in out, out
    # the two-step evaluation: the fast path decides all but a few results in 10^4, and its own error stays a factor of
    # four (2 bits) below the threshold its decision assumes
    fast = re.findall(r"fast path: undecided on (\d+) of (\d+) calls(?:, largest relative error 2\^(-[\d.]+) \(threshold 2\^(-\d+)\))?", out)
    assert len(fast) == 3, out
    for undecided, calls, err, thr in fast:
        assert int(undecided) <= int(calls) // 2000, out
        if err:
            assert float(err) <= float(thr) - 2.0, out


@pytest.mark.parametrize("fn", ["arcsin", "arccos", "arctan", "tan"])
def test_numpy_simd_functions_are_restated_bit_for_bit(fn, tmp_path):
    """photonbend_amd/csrc/pb_math_np.hpp (host build) against the RESULT BITS of the NumPy that made the goldens
    (tests/golden/npmath.npz: np.arcsin / np.arccos / np.arctan / np.tan on an AVX512_SKX machine - what rotation.py:158 and
    lens.py:71-307 run there), on 40 000 arguments per function.  Not a tolerance: every bit, NaN for NaN."""
    import shutil
    import subprocess

    from tests import npmath_args

    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_math")
    res = subprocess.run([gxx, "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(root, "oracle", "check_math.cpp"), "-lquadmath"],
                         capture_output=True, text=True)
    if res.returncode != 0 and "quadmath" in res.stderr:
        pytest.skip("libquadmath missing")
    assert res.returncode == 0, res.stderr
    gold = np.load(os.path.join(root, "tests", "golden", "npmath.npz"))
    x = npmath_args.arguments(fn)
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    x.tofile(src)
    code = 3 + npmath_args.FUNCTIONS.index(fn)
    assert subprocess.run([exe, "--eval", str(code), src, dst], timeout=120).returncode == 0
    got, want = np.fromfile(dst, dtype=np.uint64), gold[fn]
    assert got.size == want.size == x.size
    both_nan = np.isnan(got.view(np.float64)) & np.isnan(want.view(np.float64))
    bad = np.flatnonzero((got != want) & ~both_nan)
    assert bad.size == 0, f"{fn}: {bad.size} of {x.size} differ from NumPy, first at x = {x[bad[0]].hex()}: {got[bad[0]]:#018x} vs {want[bad[0]]:#018x}"
    if fn == "tan":
        return
    # ... and where NumPy here IS that NumPy, a million fresh arguments (skipped on machines whose NumPy takes another code path)
    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
    except ImportError:
        from numpy.core._multiarray_umath import __cpu_features__ as feats
    with np.errstate(all="ignore"):
        if not feats.get("AVX512_SKX") or getattr(np, fn)(x).view(np.uint64)[~both_nan].tobytes() != want[~both_nan].tobytes():
            return
    rng = np.random.default_rng(7)
    y = (2.0 * rng.random(1_000_000) - 1.0) * (8.0 if fn == "arctan" else 1.0)
    y.tofile(src)
    assert subprocess.run([exe, "--eval", str(code), src, dst], timeout=120).returncode == 0
    assert np.fromfile(dst, dtype=np.uint64).tobytes() == getattr(np, fn)(y).view(np.uint64).tobytes()
