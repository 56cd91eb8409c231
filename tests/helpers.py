"""Shared helpers: build oracle / product objects from the case tuples."""

from __future__ import annotations

import json
import os

import numpy as np

from oracle import reference_path as orc
from oracle.synth import synth_frame
from tests.cases import Case

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


_LIVE_IS_GOLDEN = None


def live_numpy_is_the_goldens_numpy() -> bool:
    """True when THIS machine's NumPy / libm return the bits stored in golden/npmath.npz for np.arcsin, np.arccos, np.arctan, np.tan, np.sin,
    np.cos, np.exp(1j x) and np.log(z).imag - i.e. when a LIVE oracle run reproduces the goldens' platform (x86-64 with FMA and
    AVX512_SKX, glibc 2.35, NumPy 2.2.6).  The device chain restates that platform's functions; on another host the live oracle may
    differ from it in the last bit, and comparisons against the live oracle keep their fragile-set allowance there."""
    global _LIVE_IS_GOLDEN
    if _LIVE_IS_GOLDEN is None:
        from tests import npmath_args

        gold = np.load(os.path.join(GOLD, "npmath.npz"))
        ok = True
        with np.errstate(all="ignore"):
            for fn in npmath_args.FUNCTIONS:
                got, want = npmath_args.reference(fn, npmath_args.arguments(fn)), gold[fn]
                nan = np.isnan(got.view(np.float64)) & np.isnan(want.view(np.float64))
                ok = ok and bool(((got == want) | nan).all())
        _LIVE_IS_GOLDEN = ok
    return _LIVE_IS_GOLDEN


def canonical_map_sha(m) -> str:
    """SHA-256 of a float64 coordinate map's bits with every NaN replaced by the canonical quiet NaN (oracle/make_goldens.py does the same
    to the reference's maps): signed zeros, flags and every finite bit count, NaN payloads do not."""
    import hashlib

    a = np.ascontiguousarray(m, dtype=np.float64).copy()
    a[np.isnan(a)] = np.float64("nan")
    return hashlib.sha256(a.view(np.uint64).tobytes()).hexdigest()


def pb_map_stages(case: Case):
    """The package's materialised float64 maps of a case: after get_coordinate_map and after each rotation (host copies, one at a time)."""
    import photonbend_amd as pb

    cmap = pb_obj(case.dst).get_coordinate_map()
    yield np.array(np.asarray(cmap))
    for rot in case.rotations:
        cmap = pb.Rotation(*map(pb.utils.to_radians, rot)).rotate_coordinate_map(cmap)
        yield np.array(np.asarray(cmap))


def load_small():
    return np.load(os.path.join(GOLD, "small.npz"))


def load_full():
    with open(os.path.join(GOLD, "full.json")) as f:
        return json.load(f)


def orc_proj(p) -> orc.Proj:
    kind, h, w, lens, fov, mag = p
    if kind == "pano":
        return orc.Proj("pano", h, w)
    if lens in ("custom", "thobylike"):  # a Lens of user callables: the oracle takes the (forward, reverse) pair
        from tests import cases as tc

        lens = (tc.custom_forward, tc.custom_reverse) if lens == "custom" else (tc.thoby_like_forward, tc.thoby_like_reverse)
    return orc.Proj(kind, h, w, lens, orc.to_radians(fov), mag)


def orc_rots(case: Case):
    return [tuple(map(orc.to_radians, r)) for r in case.rotations]


def case_frame(case: Case, frame: int = 0) -> np.ndarray:
    _, h, w, *_ = case.src
    return synth_frame(h, w, frame=frame, seed=0, circle_mask=case.mask)


def pb_obj(p, image=None):
    """The product-side object for a case tuple (import deferred: needs torch)."""
    import photonbend_amd as pb

    kind, h, w, lens, fov, mag = p
    if image is None:
        image = np.zeros((h, w, 3), np.uint8)
    if kind == "pano":
        return pb.PanoramaImage(image)
    if lens == "custom":
        from tests import cases as tc

        L = pb.Lens(tc.custom_forward, tc.custom_reverse)
    elif lens == "thobylike":
        from tests import cases as tc

        L = pb.Lens(tc.thoby_like_forward, tc.thoby_like_reverse)
    else:
        L = getattr(pb, lens)()
    if kind == "camera":
        return pb.CameraImage(image, pb.utils.to_radians(fov), L, magnitude=mag)
    return pb.DoubleCameraImage(image, pb.utils.to_radians(fov), L)


def pb_chain(case: Case, image=None):
    """dst.get_coordinate_map() -> rotations -> (src object, lazy map)."""
    import photonbend_amd as pb

    dst = pb_obj(case.dst)
    cmap = dst.get_coordinate_map()
    for rot in case.rotations:
        cmap = pb.Rotation(*map(pb.utils.to_radians, rot)).rotate_coordinate_map(cmap)
    src = pb_obj(case.src, image if image is not None else case_frame(case))
    return src, cmap


def pb_plan(case: Case):
    from photonbend_amd.core.projection import _plan_for

    from photonbend_amd import _native as nat

    src, cmap = pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
    plan = _plan_for(cmap.dst_proj, cmap.rotations, src._proj())
    plan.set_mode(nat.MODE_AUTO)  # the facade's cache entry is shared: an earlier test of the same geometry may have left its mode
    return plan


def pb_plan_private(case: Case, **kw):
    """A plan of its own (not the facade's shared cache entry): for tests that re-budget or re-mode a plan."""
    from photonbend_amd import _native as nat

    src, cmap = pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
    kw.setdefault("bilinear", True)  # (most private plans are the bilinear tests': the mode's tables at creation, as the C ABI's default does)
    return nat.Plan(cmap.dst_proj, cmap.rotations, src._proj(), **kw)
