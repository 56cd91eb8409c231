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


@pytest.mark.parametrize("fn", ["arcsin", "arccos", "arctan", "tan", "sin", "cos", "expi", "arg"])
def test_numpy_and_glibc_functions_are_restated_bit_for_bit(fn, tmp_path):
    """photonbend_amd/csrc/pb_math_np.hpp and pb_math_glibc.hpp (host build) against the RESULT BITS of the NumPy that made the goldens
    (tests/golden/npmath.npz): np.arcsin / np.arccos / np.arctan / np.tan (NumPy's AVX-512 kernels), np.sin / np.cos / np.exp(x * 1j) /
    np.log(z).imag (glibc 2.35) - what rotation.py:129-164, lens.py:71-335 and projection.py:193, :252 run there - on 40 000 arguments
    per function.  Not a tolerance: every bit, NaN for NaN.  Where this machine's NumPy IS that NumPy, a million fresh arguments too."""
    import shutil
    import subprocess

    from tests import npmath_args

    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_math")
    res = subprocess.run([gxx, "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(root, "oracle", "check_math.cpp")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    gold = np.load(os.path.join(root, "tests", "golden", "npmath.npz"))
    code = npmath_args.FUNCTIONS.index(fn)
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")

    def run(x):
        x.tofile(src)
        assert subprocess.run([exe, str(code), src, dst], timeout=120).returncode == 0
        return np.fromfile(dst, dtype=np.uint64)

    def differing(got, want):
        assert got.size == want.size
        both_nan = np.isnan(got.view(np.float64)) & np.isnan(want.view(np.float64))
        return np.flatnonzero((got != want) & ~both_nan)

    got, want = run(npmath_args.arguments(fn)), gold[fn]
    bad = differing(got, want)
    assert bad.size == 0, f"{fn}: {bad.size} of {got.size} differ from NumPy, first at result {bad[0]}: {got[bad[0]]:#018x} vs {want[bad[0]]:#018x}"
    if not H.live_numpy_is_the_goldens_numpy():
        return
    rng = np.random.default_rng(7)
    scale = {"arcsin": 1.0, "arccos": 1.0, "arctan": 8.0, "tan": np.pi, "sin": 7.0, "cos": 7.0, "expi": 7.0}
    if fn == "arg":
        y = ((2.0 * rng.random((500_000, 2)) - 1.0) * np.ldexp(1.0, rng.integers(-30, 12, (500_000, 1)))).ravel()
    else:
        y = (2.0 * rng.random(1_000_000) - 1.0) * scale[fn]
    with np.errstate(all="ignore"):
        bad = differing(run(y), npmath_args.reference(fn, y))
    assert bad.size == 0, f"{fn}: {bad.size} of a million fresh results differ from this machine's NumPy"


def test_this_hosts_numpy_against_the_goldens_numpy(capsys):
    """Reports whether THIS host's NumPy / libm reproduce the result bits of golden/npmath.npz - i.e. whether live-oracle comparisons on
    this host are bit-exact comparisons with the goldens' platform (tests/helpers.py).  Where the host matches the fixture's recorded
    platform (NumPy version, glibc version, AVX512_SKX, FMA3) it MUST reproduce it; elsewhere the outcome is only reported."""
    import ctypes
    import json

    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
    except ImportError:
        from numpy.core._multiarray_umath import __cpu_features__ as feats
    meta = json.loads(bytes(np.load(os.path.join(H.GOLD, "npmath.npz"))["meta"]).decode())
    libc = ctypes.CDLL(None)
    try:
        libc.gnu_get_libc_version.restype = ctypes.c_char_p
        glibc = libc.gnu_get_libc_version().decode()
    except AttributeError:
        glibc = "?"
    here = {"numpy": np.__version__, "glibc": glibc, "AVX512_SKX": bool(feats.get("AVX512_SKX")), "FMA3": bool(feats.get("FMA3"))}
    same = H.live_numpy_is_the_goldens_numpy()
    with capsys.disabled():
        print(f"\n[npmath] this host {here}; fixture {({k: meta[k] for k in here})}; live NumPy reproduces the fixture: {same}", end="")
    if all(here[k] == meta[k] for k in here):
        assert same, "same NumPy, glibc and CPU features as the fixture's platform, but different result bits"


@pytest.mark.parametrize("gen,out", [("gen_np_tables.py", "pb_np_tables.hpp"), ("gen_glibc_tables.py", "pb_glibc_tables.hpp")])
def test_generated_math_tables_are_what_their_generators_write(gen, out):
    """csrc/pb_np_tables.hpp (VRSQRT14PD / VRCP14PD as sampled from the CPU) and csrc/pb_glibc_tables.hpp (glibc 2.35's __sincostab and
    cij, read out of libm.so.6 and checked entry by entry against 60-digit arithmetic) are committed generator output: on a machine that can
    run the generator (AVX-512F; glibc 2.35) it must write the committed file byte for byte - on this CPU as on the one that wrote it."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "photonbend_amd", "csrc")
    res = subprocess.run([sys.executable, os.path.join(csrc, gen)], capture_output=True, text=True, timeout=120)
    if res.returncode != 0:
        pytest.skip(f"{gen} cannot run here: {(res.stderr or res.stdout).strip().splitlines()[-1][:160]}")
    want = open(os.path.join(csrc, out)).read()
    got = res.stdout
    if gen == "gen_np_tables.py":  # (the header names the CPU it was sampled on)
        strip = lambda t: "\n".join(l for l in t.splitlines() if not l.startswith("// VRSQRT14PD / VRCP14PD sampled on"))
        got, want = strip(got), strip(want)
    assert got.strip() == want.strip(), f"{out} differs from what {gen} writes on this machine"


def test_restated_math_is_clean_under_the_sanitizers(tmp_path):
    """The host build of csrc/pb_math.hpp under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on this pool:
    the CPU build is where table indices, shifts and conversions get checked): the whole fixture plus infinities, huge, tiny and
    out-of-domain arguments through all eight functions, no finding, same bits as the plain build."""
    import shutil
    import subprocess

    from tests import npmath_args

    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_math_san")
    res = subprocess.run([gxx, "-O1", "-g", "-ffp-contract=off", "-mfma", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                          os.path.join(root, "oracle", "check_math.cpp")], capture_output=True, text=True)
    if res.returncode != 0:
        pytest.skip("no sanitizer runtime here: " + res.stderr.strip()[-160:])
    gold = np.load(os.path.join(root, "tests", "golden", "npmath.npz"))
    wild = np.array([np.inf, -np.inf, 1e308, -1e308, 1.7e308, 5e-324, -5e-324, 2.0 ** 1023, 1e9, 65537.0, 1.05e8, np.nan, 0.0, -0.0])
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    for code, fn in enumerate(npmath_args.FUNCTIONS):
        x = np.concatenate([npmath_args.arguments(fn), wild])
        x.tofile(src)
        run = subprocess.run([exe, str(code), src, dst], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0 and not run.stderr.strip(), f"{fn}: {run.stderr[-600:]}"
        got, want = np.fromfile(dst, dtype=np.uint64)[: gold[fn].size], gold[fn]
        both_nan = np.isnan(got.view(np.float64)) & np.isnan(want.view(np.float64))
        assert bool(((got == want) | both_nan).all()), fn


def test_oracle_with_callable_lenses_reproduces_the_generic_goldens():
    """Round 5: the oracle takes a Lens of user callables as a (forward, reverse) pair (tests/helpers.py: orc_proj).  Pinned here against
    the REFERENCE's outputs for every generic case - grey / RGBA / 16-bit images, user lenses on either end, nine rotations, odd-width
    double frames (tests/golden/generic.npz, captured from photonbend itself) - before the GPU tests use that oracle as the checker of
    the bilinear mode's map path."""
    import warnings

    from oracle.synth import synth_image
    from tests import cases as tc
    from tests import helpers as H

    gold = np.load(os.path.join(H.GOLD, "generic.npz"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, case, layout in tc.generic_cases():
            _, h, w, *_ = case.src
            img = synth_image(h, w, layout, frame=3, circle_mask=case.mask)
            got = orc.remap(H.orc_proj(case.dst), H.orc_proj(case.src), img, H.orc_rots(case))
            assert got.dtype == gold[name].dtype and np.array_equal(got, gold[name]), name


@pytest.mark.parametrize("fn", ["arcsin", "arccos", "arctan", "tan"])
def test_second_math_flavour_is_libms_bit_for_bit(fn, tmp_path):
    """VERDICT r4 item 4.  photonbend_amd/csrc/pb_math_libm.hpp (host build; generated instruction by instruction from glibc 2.35's
    __asin_fma / __acos_fma / __atan_fma / __tan_fma) against the RESULT BITS of NumPy running WITHOUT its AVX-512 kernels
    (tests/golden/npmath_libm.npz: NPY_DISABLE_CPU_FEATURES, i.e. what an x86-64 host without AVX512_SKX computes for lens.py:71-307 and
    rotation.py:158) on the fixture's 40 000 arguments per function - every bit - and, on a glibc 2.35 x86-64 host, against this
    machine's own libm on a million fresh arguments (tan: the main path, |x| < 2^27)."""
    import ctypes
    import shutil
    import subprocess

    from tests import npmath_args

    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_math")
    res = subprocess.run([gxx, "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(root, "oracle", "check_math.cpp")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    gold = np.load(os.path.join(root, "tests", "golden", "npmath_libm.npz"))
    code = 8 + ["arcsin", "arccos", "arctan", "tan"].index(fn)
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")

    def run(x):
        x.tofile(src)
        assert subprocess.run([exe, str(code), src, dst], timeout=120).returncode == 0
        return np.fromfile(dst, dtype=np.uint64)

    def differing(got, want):
        both_nan = np.isnan(got.view(np.float64)) & np.isnan(want.view(np.float64))
        return np.flatnonzero((got != want) & ~both_nan)

    x = npmath_args.arguments(fn)
    got, want = run(x), gold[fn]
    if fn == "tan":  # (the huge-argument reduction is not restated: NaN there; the fixture's arguments stay below 2^17)
        assert np.abs(x[np.isfinite(x)]).max() < 2.0 ** 27
    bad = differing(got, want)
    assert bad.size == 0, f"{fn}: {bad.size} of {got.size} differ from NumPy-without-AVX-512, first at result {bad[0]}: {got[bad[0]]:#018x} vs {want[bad[0]]:#018x}"
    first = np.load(os.path.join(root, "tests", "golden", "npmath.npz"))[fn]
    assert differing(got, first).size > 0, "the two flavours must differ somewhere (arctan: 0.07 % of the results, arcsin / arccos: 7-8 %)"
    libc = ctypes.CDLL(None)
    libc.gnu_get_libc_version.restype = ctypes.c_char_p
    import platform

    if libc.gnu_get_libc_version().decode() != "2.35" or platform.machine() != "x86_64":
        return
    libm = ctypes.CDLL("libm.so.6")
    f = getattr(libm, {"arcsin": "asin", "arccos": "acos", "arctan": "atan", "tan": "tan"}[fn])
    f.restype, f.argtypes = ctypes.c_double, [ctypes.c_double]
    rng = np.random.default_rng(11)
    y = (2.0 * rng.random(200_000) - 1.0) * {"arcsin": 1.0, "arccos": 1.0, "arctan": 40.0, "tan": 1.0e5}[fn]
    ref = np.array([f(float(v)) for v in y]).view(np.uint64)
    bad = differing(run(y), ref)
    assert bad.size == 0, f"{fn}: {bad.size} of 200 000 fresh results differ from this machine's libm"


def test_second_flavour_header_is_what_its_generator_writes():
    """csrc/pb_math_libm.hpp is committed generator output (gen_libm_flavour.py reads the installed libm.so.6): on a machine with the very
    build the generator was written against it must write the committed file byte for byte."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "photonbend_amd", "csrc")
    res = subprocess.run([sys.executable, os.path.join(csrc, "gen_libm_flavour.py")], capture_output=True, text=True, timeout=300)
    if res.returncode != 0:
        pytest.skip(f"gen_libm_flavour.py cannot run here: {(res.stderr or res.stdout).strip().splitlines()[-1][:160]}")
    assert res.stdout.strip() == open(os.path.join(csrc, "pb_math_libm.hpp")).read().strip()
