"""BASELINE.json's configs at FULL size on the GPU, against pins captured from the
real reference (tests/golden/full.json: SHA-256 of the int32 index map and of the
uint8 output on the synthetic frame, plus seeded samples).  Equal hashes mean every
one of the 16.8-33.5 M pixels is bit-exact."""

import hashlib

import numpy as np
import pytest
import torch

from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases

pytestmark = pytest.mark.gpu
FULL = H.load_full()


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()


def sample_positions(pin, n_px):
    return np.random.default_rng(pin["sample_seed"]).integers(0, n_px, size=65536)


@pytest.mark.parametrize("case", full_cases(), ids=lambda c: c.name)
def test_full_size_pins(case):
    pin = FULL[case.name]
    plan = H.pb_plan(case)
    Hd, Wd = case.dst[1], case.dst[2]
    pos = sample_positions(pin, Hd * Wd)
    # host-side scalars first: a wrong f_distance would explain everything else
    if "dst_f_bits" in pin:
        assert H.bits(np.array([plan.dst.f_distance]))[0] == pin["dst_f_bits"]
    if "src_f_bits" in pin:
        assert H.bits(np.array([plan.src.f_distance]))[0] == pin["src_f_bits"]
    idx = plan.index_map()
    if case.src[0] == "double":
        flat = idx.reshape(2, -1).cpu().numpy()
        bad_l = int((flat[0][pos[:2048]] != np.array(pin["idx_l_samples"])).sum())
        bad_r = int((flat[1][pos[:2048]] != np.array(pin["idx_r_samples"])).sum())
        assert (bad_l, bad_r) == (0, 0), f"sampled index mismatches: left {bad_l}, right {bad_r} of 2048"
        assert int((flat[0] >= 0).sum()) == pin["valid_left"] and int((flat[1] >= 0).sum()) == pin["valid_right"]
        assert sha(idx[0]) == pin["idx_l_sha256"], "left index map differs somewhere (samples agree)"
        assert sha(idx[1]) == pin["idx_r_sha256"], "right index map differs somewhere (samples agree)"
    else:
        flat = idx.reshape(-1).cpu().numpy()
        bad = int((flat[pos[:2048]] != np.array(pin["idx_samples"])).sum())
        assert bad == 0, f"{bad} of 2048 sampled indices differ"
        assert int((flat >= 0).sum()) == pin["in_bounds_samples"]
        assert sha(idx) == pin["idx_sha256"], "index map differs somewhere (samples and counts agree)"
    if case.src[0] != "double":
        # the same index map, taken from the WINDOWED hot kernel (the kernel bench.py times): source pixels that
        # encode their own linear index come out as the indices the kernel sampled
        from tests.test_hip_mid import index_through_windowed_kernel

        widx = index_through_windowed_kernel(plan, case.src[1], case.src[2])
        assert torch.equal(widx, idx), f"{int((widx != idx).sum())} pixels: windowed kernel and index-map launch disagree"
        assert sha(widx) == pin["idx_sha256"]
        del widx
    del idx
    # the uint8 output on the synthetic frame generated ON THE DEVICE
    _, h, w, *_ = case.src
    frame = nat.synth_frame(h, w, frame=0, seed=0, circle_mask=case.mask)
    assert sha(frame) == pin["frame_sha256"], "device synthetic frame differs from the host formula"
    out = plan.remap(frame)
    got = out.reshape(-1, 3).cpu().numpy()
    assert np.array_equal(got[pos[:2048]].ravel(), np.array(pin["u8_samples"], dtype=np.uint8))
    assert hashlib.sha256(got[pos].tobytes()).hexdigest() == pin["u8_samples_sha256"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == pin["u8_sha256"]


def test_full_size_batch_equals_single_launches():
    """c2 geometry, 3 distinct frames: one 3-frame launch == 3 single launches
    (the index math is shared across the batch, the bytes must not be)."""
    case = [c for c in full_cases() if c.name == "c2"][0]
    plan = H.pb_plan(case)
    frames = torch.stack([nat.synth_frame(4096, 8192, frame=f) for f in range(3)])
    batched = plan.remap(frames)
    for f in range(3):
        assert torch.equal(plan.remap(frames[f]), batched[f])
    assert not torch.equal(batched[0], batched[1])


def test_full_size_identity_pano_roundtrip():
    """pano -> pano with no rotation at equal size is NOT the identity in the
    reference (row end points, quarter-pixel columns - SURVEY a-3); but applying
    it twice equals applying it once on its own output's fixed structure:
    every output pixel is a copy of some source pixel (gather property)."""
    from tests.cases import Case, pano

    case = Case("pp", pano(2048, 4096), pano(2048, 4096))
    plan = H.pb_plan(case)
    frame = nat.synth_frame(2048, 4096, frame=7)
    idx = plan.index_map().reshape(-1).long()
    out = plan.remap(frame).reshape(-1, 3)
    assert int((idx < 0).sum()) == 0
    assert torch.equal(out, frame.reshape(-1, 3)[idx])


@pytest.mark.parametrize("name", sorted(FULL))
def test_materialised_maps_are_the_references_bits_at_full_size(name):
    """f-1 at the BASELINE sizes: pb_coordmap_f64 / pb_rotate_f64 over 8.4-33.5 M pixels - latitude, longitude and invalid-flag planes - hash
    to the reference's maps (full.json: map_sha256; NaNs canonicalised), stage by stage."""
    case = next(c for c in full_cases() if c.name == name)
    want = FULL[name]["map_sha256"]
    k = -1
    for k, m in enumerate(H.pb_map_stages(case)):
        assert list(m.shape) == FULL[name]["map_shape"]
        assert H.canonical_map_sha(m) == want[k], f"{name}: the float64 map of stage {k} differs from the reference's"
        del m
    assert k + 1 == len(want)
