"""the "correctly rounded" guarantee of csrc/pb_math.hpp was only tested on the HOST build (oracle/check_math.cpp against
glibc and libquadmath, tests/test_oracle_golden.py).  What the faithful chain runs is the gfx950 build of the same header: this test
evaluates pb_sincos_cr / pb_atan2_cr / pb_atan_cr on the device (a debug entry point of the -DPB_ABLATION diagnostic build, loaded
through PB_LIB_PATH in a child process: the product has no such hook) on a million arguments - longitudes, latitudes, lens arguments,
tiny values and the pixel-centre half-integers of a destination map - and asserts BIT equality with the host build."""

import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import ctypes as C, sys, numpy as np, torch
from photonbend_amd import _native as nat
lib = nat.load()
lib.pb_debug_math.restype = C.c_int
lib.pb_debug_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
fn, src, dst = int(sys.argv[1]), sys.argv[2], sys.argv[3]
x = torch.from_numpy(np.fromfile(src, dtype=np.float64)).cuda()
n = x.numel() // (2 if fn == 1 else 1)
out = torch.empty(n * (2 if fn == 0 else 1), dtype=torch.float64, device="cuda")
nat.check(lib.pb_debug_math(fn, x.data_ptr(), out.data_ptr(), n, None))
torch.cuda.synchronize()
out.cpu().numpy().tofile(dst)
"""


def _arguments(rng, n):
    q = n // 4
    lon = (2.0 * rng.random(q) - 1.0) * np.pi
    lat = rng.random(q) * np.pi
    lens = rng.random(q) * np.pi * 0.713
    tiny = np.ldexp(2.0 * rng.random(n - 3 * q) - 1.0, -rng.integers(0, 40, n - 3 * q))
    return np.concatenate([lon, lat, lens, tiny])


@pytest.mark.gpu
@pytest.mark.parametrize("fn", [0, 1, 2], ids=["sincos", "atan2", "atan"])
def test_device_math_bits_equal_the_host_build(fn, tmp_path):
    from photonbend_amd.build import DIAG_LIB_PATH

    gxx = shutil.which("g++")
    if not gxx or not os.path.exists(DIAG_LIB_PATH):
        pytest.skip("needs g++ and the diagnostic build (python -m photonbend_amd.build --diag)")
    exe = str(tmp_path / "check_math")
    res = subprocess.run([gxx, "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(ROOT, "oracle", "check_math.cpp"), "-lquadmath"], capture_output=True, text=True)
    if res.returncode != 0:
        pytest.skip("oracle/check_math.cpp did not build here: " + res.stderr[-200:])
    rng = np.random.default_rng(4242 + fn)
    n = 1_000_000
    if fn == 1:
        # (y, x): the pixel-centre half-integers of a destination map (projection.py:177-183), random pairs, unit vectors after a rotation
        half = np.stack([rng.integers(-4096, 4096, n // 2) + 0.5, rng.integers(-4096, 4096, n // 2) + 0.5], axis=1)
        rnd = (2.0 * rng.random((n // 4, 2)) - 1.0) * np.ldexp(1.0, rng.integers(-30, 12, (n // 4, 1)))
        ang = (2.0 * rng.random(n - n // 2 - n // 4) - 1.0) * np.pi
        unit = np.stack([np.sin(ang), np.cos(ang)], axis=1) * rng.random((ang.size, 1))
        x = np.concatenate([half, rnd, unit]).ravel()
    else:
        x = _arguments(rng, n)
        if fn == 2:
            x = np.concatenate([x[: n // 2], (2.0 * rng.random(n - n // 2) - 1.0) * 8.0])  # r / 2, r: stereographic and rectilinear inverses
    src, dev_out, host_out = str(tmp_path / "in.bin"), str(tmp_path / "dev.bin"), str(tmp_path / "host.bin")
    np.ascontiguousarray(x, dtype=np.float64).tofile(src)
    env = dict(os.environ, PB_LIB_PATH=DIAG_LIB_PATH)
    res = subprocess.run([sys.executable, "-c", _SCRIPT, str(fn), src, dev_out], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    res = subprocess.run([exe, "--eval", str(fn), src, host_out], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    dev = np.fromfile(dev_out, dtype=np.uint64)
    host = np.fromfile(host_out, dtype=np.uint64)
    assert dev.size == host.size == (2 * n if fn == 0 else n)
    bad = int((dev != host).sum())
    assert bad == 0, f"{bad} of {dev.size} results differ between the gfx950 build and the host build of pb_math.hpp"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["arcsin", "arccos", "arctan", "tan"])
def test_device_build_of_the_numpy_functions_returns_numpys_bits(name, tmp_path):
    """csrc/pb_math_np.hpp as the faithful chain runs it - the gfx950 build, instruction tables in device memory - against the result
    bits of the NumPy that produced the goldens (tests/golden/npmath.npz): np.arcsin / np.arccos / np.arctan / np.tan are what
    rotation.py:158 and lens.py:71-307 call.  Every bit of 40 000 results per function; NaN for NaN."""
    from photonbend_amd.build import DIAG_LIB_PATH
    from tests import npmath_args

    if not os.path.exists(DIAG_LIB_PATH):
        pytest.skip("needs the diagnostic build (python -m photonbend_amd.build --diag)")
    fn = 3 + npmath_args.FUNCTIONS.index(name)
    x = npmath_args.arguments(name)
    src, dev_out = str(tmp_path / "in.bin"), str(tmp_path / "dev.bin")
    x.tofile(src)
    env = dict(os.environ, PB_LIB_PATH=DIAG_LIB_PATH)
    res = subprocess.run([sys.executable, "-c", _SCRIPT, str(fn), src, dev_out], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    got = np.fromfile(dev_out, dtype=np.uint64)
    want = np.load(os.path.join(ROOT, "tests", "golden", "npmath.npz"))[name]
    assert got.size == want.size == x.size
    both_nan = np.isnan(got.view(np.float64)) & np.isnan(want.view(np.float64))
    bad = np.flatnonzero((got != want) & ~both_nan)
    assert bad.size == 0, f"{name}: {bad.size} of {x.size} device results differ from NumPy, first at x = {x[bad[0]].hex()}: {got[bad[0]]:#018x} vs {want[bad[0]]:#018x}"
