"""What the faithful chain runs is the gfx950 build of csrc/pb_math.hpp - NumPy's and glibc's transcendental functions restated bit for
bit.  The host build is checked against NumPy's result bits in tests/test_oracle_golden.py; this test evaluates the same functions ON THE
DEVICE (a debug entry point of the -DPB_ABLATION diagnostic build, loaded through PB_LIB_PATH in a child process: the product has no
such hook) and compares with the same fixture."""

import os
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
n = x.numel() // (2 if fn == 7 else 1)
out = torch.empty(n * (2 if fn == 6 else 1), dtype=torch.float64, device="cuda")
nat.check(lib.pb_debug_math(fn, x.data_ptr(), out.data_ptr(), n, None))
torch.cuda.synchronize()
out.cpu().numpy().tofile(dst)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["arcsin", "arccos", "arctan", "tan", "sin", "cos", "expi", "arg"])
def test_device_build_of_the_numpy_functions_returns_numpys_bits(name, tmp_path):
    """csrc/pb_math_np.hpp and pb_math_glibc.hpp as the faithful chain runs them - the gfx950 build, tables in device memory - against the
    result bits of the NumPy that produced the goldens (tests/golden/npmath.npz): np.arcsin / np.arccos / np.arctan / np.tan / np.sin /
    np.cos / np.exp(x * 1j) / np.log(z).imag are every transcendental call of rotation.py:129-164, lens.py:71-335 and
    projection.py:193, :252.  Every bit of 20 000 - 80 000 results per function; NaN for NaN."""
    from photonbend_amd.build import DIAG_LIB_PATH
    from tests import npmath_args

    if not os.path.exists(DIAG_LIB_PATH):
        pytest.skip("needs the diagnostic build (python -m photonbend_amd.build --diag)")
    fn = npmath_args.FUNCTIONS.index(name)
    x = npmath_args.arguments(name)
    src, dev_out = str(tmp_path / "in.bin"), str(tmp_path / "dev.bin")
    x.tofile(src)
    env = dict(os.environ, PB_LIB_PATH=DIAG_LIB_PATH)
    res = subprocess.run([sys.executable, "-c", _SCRIPT, str(fn), src, dev_out], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    got = np.fromfile(dev_out, dtype=np.uint64)
    want = np.load(os.path.join(ROOT, "tests", "golden", "npmath.npz"))[name]
    assert got.size == want.size
    both_nan = np.isnan(got.view(np.float64)) & np.isnan(want.view(np.float64))
    bad = np.flatnonzero((got != want) & ~both_nan)
    assert bad.size == 0, f"{name}: {bad.size} of {got.size} device results differ from NumPy, first at result {bad[0]}: {got[bad[0]]:#018x} vs {want[bad[0]]:#018x}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["arcsin", "arccos", "arctan", "tan"])
def test_device_build_of_the_second_flavour_returns_libms_bits(name, tmp_path):
    """csrc/pb_math_libm.hpp as the gfx950 build runs it (tables in device memory) against tests/golden/npmath_libm.npz - the result bits
    of NumPy WITHOUT its AVX-512 kernels, i.e. glibc 2.35's asin / acos / atan / tan: every bit of 40 000 results per function."""
    from photonbend_amd.build import DIAG_LIB_PATH
    from tests import npmath_args

    if not os.path.exists(DIAG_LIB_PATH):
        pytest.skip("needs the diagnostic build (python -m photonbend_amd.build --diag)")
    fn = 8 + ["arcsin", "arccos", "arctan", "tan"].index(name)
    x = npmath_args.arguments(name)
    src, dev_out = str(tmp_path / "in.bin"), str(tmp_path / "dev.bin")
    x.tofile(src)
    env = dict(os.environ, PB_LIB_PATH=DIAG_LIB_PATH)
    res = subprocess.run([sys.executable, "-c", _SCRIPT, str(fn), src, dev_out], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    got = np.fromfile(dev_out, dtype=np.uint64)
    want = np.load(os.path.join(ROOT, "tests", "golden", "npmath_libm.npz"))[name]
    assert got.size == want.size
    both_nan = np.isnan(got.view(np.float64)) & np.isnan(want.view(np.float64))
    bad = np.flatnonzero((got != want) & ~both_nan)
    assert bad.size == 0, f"{name}: {bad.size} of {got.size} device results differ from NumPy-without-AVX-512, first at result {bad[0]}: {got[bad[0]]:#018x} vs {want[bad[0]]:#018x}"
