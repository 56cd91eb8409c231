"""CPU-only checks of the boundary: the C-ABI library loads and exports every
symbol include/photonbend_hip.h declares; the Python host side reproduces the
reference's host-side scalars (f_distance, rotation matrices) bit for bit and its
error behaviour; the lazy coordinate map keeps its recipe semantics."""

import ctypes
import os
import re

import numpy as np
import pytest

import photonbend_amd as pb
from photonbend_amd import _native as nat
from photonbend_amd.build import LIB_PATH
from tests import helpers as H
from tests.cases import small_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = H.load_small()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "photonbend_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB_PATH), "build the HIP library first (python -m photonbend_amd.build)"
    lib = ctypes.CDLL(LIB_PATH)
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(nat.SIGNATURES) == names, "ctypes binding and header disagree"
    assert nat.load().pb_abi_version() == nat.ABI_VERSION == 5


def test_abi_argument_validation_without_gpu():
    lib = nat.load()
    h = ctypes.c_void_p()
    good = nat.make_proj(nat.KIND_PANO, 4, 8)
    bad = nat.make_proj(7, 4, 8)
    assert lib.pb_plan_create(ctypes.byref(bad), None, 0, ctypes.byref(good), ctypes.byref(h)) == -1
    assert b"kind" in lib.pb_last_error()
    assert lib.pb_plan_create(ctypes.byref(good), None, 9, ctypes.byref(good), ctypes.byref(h)) == -1
    odd = nat.make_proj(nat.KIND_DOUBLE, 4, 9)
    # an odd-width double DESTINATION is an error (the reference's map has 2 * (W // 2) columns: the host passes that);
    # an odd-width double SOURCE is fine (eyes of W // 2 and W - W // 2 columns, projection.py:429-431)
    assert lib.pb_plan_create(ctypes.byref(odd), None, 0, ctypes.byref(good), ctypes.byref(h)) == -1
    assert b"even width" in lib.pb_last_error()
    assert lib.pb_plan_create(ctypes.byref(good), None, 0, ctypes.byref(odd), ctypes.byref(h)) == 0
    lib.pb_plan_destroy(h)
    custom = nat.make_proj(nat.KIND_CAMERA, 8, 8, nat.LENS_CUSTOM, 3.0, 3.5, 2.0)
    assert lib.pb_plan_create(ctypes.byref(good), None, 0, ctypes.byref(custom), ctypes.byref(h)) == -1
    assert b"PB_LENS_CUSTOM" in lib.pb_last_error()
    assert lib.pb_plan_create(ctypes.byref(good), None, 0, ctypes.byref(good), ctypes.byref(h)) == 0
    hh, ww = ctypes.c_int(), ctypes.c_int()
    assert lib.pb_plan_dst_shape(h, ctypes.byref(hh), ctypes.byref(ww)) == 0 and (hh.value, ww.value) == (4, 8)
    lib.pb_plan_destroy(h)


@pytest.mark.parametrize("case", small_cases(), ids=lambda c: c.name)
def test_host_scalars_match_reference_bits(case):
    n = case.name
    dst, src = H.pb_obj(case.dst), H.pb_obj(case.src)
    if f"{n}/dst_f" in SMALL:
        assert H.bits(np.array([dst.f_distance]))[0] == SMALL[f"{n}/dst_f"][0]
    if f"{n}/src_f" in SMALL:
        assert H.bits(np.array([src.f_distance]))[0] == SMALL[f"{n}/src_f"][0]
    if case.rotations:
        R = np.stack([pb.Rotation(*map(pb.utils.to_radians, r)).rotation_matrix for r in case.rotations])
        assert np.array_equal(H.bits(R), SMALL[f"{n}/R"])


def test_lens_host_functions_match_reference_bits():
    g = np.load(H.GOLD + "/lens.npz")
    grid = g["grid"].view(np.float64)
    with np.errstate(all="ignore"):
        for name in ("equidistant", "equisolid", "rectilinear", "stereographic", "orthographic", "thoby"):
            L = getattr(pb, name)()
            assert np.array_equal(H.bits(L.forward_function(np.copy(grid))), g[f"{name}_fwd"]), name
            assert np.array_equal(H.bits(L.reverse_function(np.copy(grid))), g[f"{name}_inv"]), name


def test_error_behaviour_mirrors_reference():
    z = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(ValueError):  # lens.py:91-94 via projection.py:143
        pb.CameraImage(z, pb.utils.to_radians(179), pb.rectilinear())
    with pytest.raises(ValueError):  # lens.py:88-89
        pb.rectilinear().forward_function(-0.1)
    # magnitude kwarg is swallowed by DoubleCameraImage (projection.py:296-316)
    d = pb.DoubleCameraImage(np.zeros((8, 16, 3), np.uint8), 3.3, pb.equidistant(), magnitude=99.0)
    assert d.magnitude == 4.0
    assert pb.CameraImage(z, 3.0, pb.equidistant()).magnitude == 4.0  # default: height / 2.0
    # a Lens of user callables is accepted like in the reference (lens.py:48-64); its map needs the GPU for the mesh
    custom = pb.Lens(lambda t: t * 1.01, lambda r: r / 1.01)
    cam = pb.CameraImage(z, 3.0, custom)
    assert cam.f_distance == 4.0 / (1.5 * 1.01)
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(nat.PbError):  # no CPU path behind the user's back
            cam.get_coordinate_map()


def test_lazy_coordinate_map_recipe():
    dst = pb.CameraImage(np.zeros((6, 10, 3), np.uint8), 3.0, pb.equisolid(), magnitude=2.5)
    m = dst.get_coordinate_map()
    assert m.shape == (6, 10, 3) and m.dtype == np.float64 and m.is_lazy and len(m) == 6
    r = pb.Rotation(0.1, 0.2, 0.3)
    m2 = r.rotate_coordinate_map(m)
    assert m2 is not m and m2.is_lazy and len(m2.rotations) == 1 and len(m.rotations) == 0
    assert np.array_equal(m2.rotations[0], r.rotation_matrix)
    m3 = pb.Rotation(0.0, 0.5, 0.0).rotate_coordinate_map(m2)
    assert len(m3.rotations) == 2


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    src = pb.PanoramaImage(np.zeros((4, 8, 3), np.uint8))
    dst = pb.CameraImage(np.zeros((4, 4, 3), np.uint8), 3.0, pb.equidistant())
    with pytest.raises(nat.PbError):
        src.process_coordinate_map(dst.get_coordinate_map())
    with pytest.raises(nat.PbError):
        np.asarray(dst.get_coordinate_map())


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "photonbend_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", text, flags=re.M), f
