"""Everything the reference's core accepts beyond uint8 RGB + built-in lenses, against the REFERENCE's own outputs
(tests/golden/generic.npz, oracle/make_goldens.py --generic): grey (H, W) / RGBA / 16-bit images (it fancy-indexes
whatever array it is given, projection.py:234-243, :545-546), ``Lens`` objects made of user callables (lens.py:48-64)
on either end, more than eight chained rotations, odd-width double-fisheye frames, and the host-side size rule
utils.calculate_size_panorama_to_photo."""

import json
import os

import numpy as np
import pytest
import torch

import photonbend_amd as pb
from oracle.synth import synth_image
from tests import cases as tc
from tests import helpers as H

GOLD = np.load(os.path.join(H.GOLD, "generic.npz"))
CASES = tc.generic_cases()


def run_case(case, img):
    dst = H.pb_obj(case.dst)
    cmap = dst.get_coordinate_map()
    for rot in case.rotations:
        cmap = pb.Rotation(*map(pb.utils.to_radians, rot)).rotate_coordinate_map(cmap)
    return H.pb_obj(case.src, img).process_coordinate_map(cmap)


def test_size_rule_matches_reference_values():
    rows = json.load(open(os.path.join(H.GOLD, "size_rule.json")))
    assert len(rows) == 40
    for lens, w, h, vert, rw, rh in rows:
        got = pb.utils.calculate_size_panorama_to_photo((w, h), getattr(pb, lens)().forward_function, bool(vert))
        assert tuple(got) == (rw, rh), (lens, w, h, vert)
    with pytest.raises(AssertionError):
        pb.utils.calculate_size_panorama_to_photo((100, 60), pb.equidistant().forward_function)
    assert "calculate_size_panorama_to_photo" in pb.utils.__all__


def test_custom_lens_host_scalars_and_roles():
    from photonbend_amd import _native as nat

    L = pb.Lens(tc.custom_forward, tc.custom_reverse)
    cam = pb.CameraImage(np.zeros((40, 40, 3), np.uint8), pb.utils.to_radians(170), L, magnitude=19.5)
    assert cam.f_distance == 19.5 / tc.custom_forward(pb.utils.to_radians(170) / 2)
    assert cam._proj("dst").lens == nat.LENS_CUSTOM and cam._proj("src").lens == nat.LENS_CUSTOM
    # a lens is custom only in the role that uses the user callable
    mixed = pb.Lens(pb.thoby().forward_function, tc.custom_reverse)
    m = pb.CameraImage(np.zeros((40, 40, 3), np.uint8), 3.0, mixed)
    assert m._proj("src").lens == nat.LENS_IDS["thoby"] and m._proj("dst").lens == nat.LENS_CUSTOM
    # an odd-width double frame: its map has 2 * (W // 2) columns, its source projection keeps the real width
    d = pb.DoubleCameraImage(np.zeros((32, 65, 3), np.uint8), pb.utils.to_radians(195), pb.equidistant())
    assert d._proj("dst").width == 64 and d._proj("src").width == 65 and d.get_coordinate_map().shape == (32, 64, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("name,case,layout", CASES, ids=[c[0] for c in CASES])
def test_generic_case_matches_reference(name, case, layout):
    want = GOLD[name]
    _, h, w, *_ = case.src
    img = synth_image(h, w, layout, frame=3, circle_mask=case.mask)
    got = run_case(case, img)
    assert isinstance(got, np.ndarray) and got.shape == want.shape and got.dtype == want.dtype, (got.shape, got.dtype, want.shape, want.dtype)
    # every case, double-fisheye sources and their float64 blend included: the reference's bytes (round 4: the chain's transcendentals
    # are the reference's own, bit for bit)
    bad = int((got != want).reshape(got.shape[0], got.shape[1], -1).any(axis=2).sum())
    assert bad == 0, f"{name}: {bad} pixels differ from the reference"
    # the same image as a CUDA tensor stays on the device and gives the same pixels
    if layout in ("RGBA", "L", "RGB"):
        t = torch.from_numpy(img).cuda()
        got_t = run_case(case, t)
        assert isinstance(got_t, torch.Tensor) and got_t.is_cuda and np.array_equal(got_t.cpu().numpy(), got)


@pytest.mark.gpu
def test_thoby_as_user_callables_equals_builtin_thoby():
    """a Lens made of the thoby formulas as Python callables reproduces the built-in lens's bytes."""
    from tests.cases import Case, cam, inscribed, pano

    img = synth_image(32, 64, "RGB", frame=3)
    rot = [(10, 20, 30)]
    a = run_case(Case("a", cam(40, 40, "thobylike", 180, inscribed(40)), pano(32, 64), rot), img)
    b = run_case(Case("b", cam(40, 40, "thoby", 180, inscribed(40)), pano(32, 64), rot), img)
    assert np.array_equal(a, b)
    img = synth_image(48, 48, "RGB", frame=3, circle_mask=1)
    a = run_case(Case("a", pano(32, 64), cam(48, 48, "thobylike", 180, inscribed(48)), mask=1), img)
    b = run_case(Case("b", pano(32, 64), cam(48, 48, "thoby", 180, inscribed(48)), mask=1), img)
    assert np.array_equal(a, b)


@pytest.mark.gpu
def test_grey_double_source_fails_like_numpy_broadcasting():
    dbl = pb.DoubleCameraImage(synth_image(40, 80, "L", frame=1), pb.utils.to_radians(195), pb.equidistant())
    cmap = pb.PanoramaImage(np.zeros((32, 64, 3), np.uint8)).get_coordinate_map()
    with pytest.raises(ValueError, match="broadcast"):
        dbl.process_coordinate_map(cmap)
