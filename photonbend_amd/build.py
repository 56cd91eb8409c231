"""Builds the HIP shared library in-tree with hipcc for gfx950 (cross-compiles
without a GPU).  ``python -m photonbend_amd.build`` or ``build_library()``."""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libphotonbend_hip.so"
# the product library of the FIRST math flavour (what _native.py loads on a host with AVX512_SKX; PB_LIB_PATH there loads another build
# of the same sources - A/B experiments: -DPB_STAMPS, -DPB_ABLATION)
LIB_PATH = os.path.join(HERE, LIB_NAME)

# -ffp-contract=off: the reference rounds every multiply and add separately; the
# kernels fuse only where they say fma() (see csrc/pb_stages.hpp).
# -mllvm -disable-machine-licm: the float64 kernels evaluate five table-and-polynomial transcendental kernels per pixel inside loops (16
# pixels per lane in certification); hoisting those kernels' hundred-odd float64 constants out of the loops made certification need 212
# VGPRs (two waves per SIMD) where 94 suffice (five): plan preparation c3 1.15 -> 1.02 ms, c5 1.52 -> 1.31.  The hot kernels keep their
# register counts and their times with or without it (experiments/README.md, round 4).
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-mllvm",
    "-disable-machine-licm",
    "-fPIC",
    "-shared",
    "-fvisibility=hidden",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "photonbend_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, out: str = None, defines=()) -> str:
    out = out or LIB_PATH
    if not force and out == LIB_PATH and not _stale():
        return LIB_PATH
    cmd = [_hipcc(), *HIPCC_FLAGS, *[f"-D{d}" for d in defines], *sources(), "-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr:
        print(res.stderr, file=sys.stderr)
    return out


# The SECOND MATH FLAVOUR of the same sources (-DPB_MATH_LIBM, csrc/pb_math.hpp): np.arcsin / arccos / arctan / tan as an x86-64 host
# WITHOUT AVX512_SKX computes them (glibc's asin / acos / atan / tan instead of NumPy's AVX-512 kernels).  A product library like the
# first: _native.py loads the one that matches the host's own NumPy dispatch (PB_MATH_FLAVOUR overrides).
LIBM_LIB_NAME = "libphotonbend_hip_libm.so"
LIBM_LIB_PATH = os.path.join(HERE, LIBM_LIB_NAME)


def build_libm_flavour(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(LIBM_LIB_PATH) and not _stale_against(LIBM_LIB_PATH):
        return LIBM_LIB_PATH
    return build_library(force=True, verbose=verbose, out=LIBM_LIB_PATH, defines=("PB_MATH_LIBM",))


# NOT in the package directory: only the product library sits next to _native.py (build/ is git-ignored and travels to the GPU box)
DIAG_LIB_PATH = os.path.join(os.path.dirname(HERE), "build", "libphotonbend_hip_diag.so")


def build_diagnostic(force: bool = False, verbose: bool = False) -> str:
    """The -DPB_ABLATION build of the same sources (never the product: loaded only through PB_LIB_PATH by experiments/ and by
    the error-path test): it alone reads the experiment knobs and the allocation-failure hook from the environment."""
    if not force and os.path.exists(DIAG_LIB_PATH) and not _stale_against(DIAG_LIB_PATH):
        return DIAG_LIB_PATH
    os.makedirs(os.path.dirname(DIAG_LIB_PATH), exist_ok=True)
    # (in the HOST's math flavour, like the product library the tests compare it with - ADVICE r5)
    return build_library(force=True, verbose=verbose, out=DIAG_LIB_PATH, defines=("PB_ABLATION",) + (("PB_MATH_LIBM",) if _host_flavour() == "libm" else ()))


def _host_flavour() -> str:
    """"svml" or "libm": what this host's NumPy dispatches np.arcsin / arccos / arctan / tan to (_native.host_math_flavour, without importing it:
    _native imports this module)."""
    env = os.environ.get("PB_MATH_FLAVOUR", "").lower()
    if env in ("svml", "libm"):
        return env
    try:
        try:
            from numpy._core._multiarray_umath import __cpu_features__ as feats
        except ImportError:
            from numpy.core._multiarray_umath import __cpu_features__ as feats
    except Exception:
        return "svml"
    return "svml" if feats.get("AVX512_SKX") else "libm"


def _stale_against(path: str) -> bool:
    t = os.path.getmtime(path)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "photonbend_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    if "--libm" in sys.argv or _host_flavour() == "libm":  # (a host without AVX512_SKX loads the second flavour: build it without being asked)
        print(build_libm_flavour(force=True, verbose=True))
    if "--diag" in sys.argv:
        print(build_diagnostic(force=True, verbose=True))
