"""Double-fisheye sources on the GPU (DoubleCameraImage.process_coordinate_map, projection.py:408-462):
the per-eye tile plans + weight classes must reproduce the faithful float64 kernel byte for byte -
unrotated (row-table weights), rotated (merge-band tiles fall back to the faithful chain), from fisheye and
double-fisheye destinations, in batches, with strides, and with frames the LDS-DMA path cannot take."""

import pytest
import torch

from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import Case, cam, dbl, inscribed, pano

pytestmark = pytest.mark.gpu

CASES = [
    Case("stitch_180", pano(512, 1024), dbl(486, 972, "equidistant", 180), mask=2),
    Case("stitch_195", pano(512, 1024), dbl(486, 972, "equidistant", 195), mask=2),
    Case("stitch_195_aligned", pano(512, 1024), dbl(480, 960, "equidistant", 195), mask=2),
    Case("stitch_220_raw", pano(384, 768), dbl(400, 800, "equisolid", 220)),
    Case("stitch_195_rot", pano(512, 1024), dbl(486, 972, "equidistant", 195), [(3, 90, -7)], mask=2),
    Case("stitch_200_chain", pano(320, 640), dbl(486, 972, "stereographic", 200), [(30, 45, 10), (-40, 5, 77)], mask=2),
    Case("fisheye_from_double", cam(512, 512, "equidistant", 360, inscribed(512)), dbl(486, 972, "equidistant", 195), [(20, 30, 40)], mask=2),
    Case("fisheye_from_double_norot", cam(500, 500, "equisolid", 200, inscribed(500)), dbl(400, 800, "equidistant", 190)),
    Case("double_from_double", dbl(400, 800, "equisolid", 190), dbl(486, 972, "equidistant", 195), [(0, 15, 0)], mask=2),
    Case("odd_sizes", pano(333, 701), dbl(301, 602, "equidistant", 187), [(1, 2, 3)]),
    Case("thoby", pano(256, 512), dbl(300, 600, "thoby", 185)),
]


def _frames(case, n):
    _, h, w, *_ = case.src
    return torch.stack([nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(n)])


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_double_fast_equals_faithful(case):
    plan = H.pb_plan(case)
    info = plan.info()
    assert info["fast_path"] and info["tiles"] > 0
    # the fast path must actually carry the frame: merge-band tiles may fail under a rotation, never most tiles
    assert info["fix_tiles"] <= info["tiles"] // 2
    frames = _frames(case, 3)
    plan.set_mode(nat.MODE_FAITHFUL)
    want = plan.remap(frames).clone()
    for mode in (nat.MODE_FAST, nat.MODE_AUTO, nat.MODE_FAST_DIRECT):
        plan.set_mode(mode)
        got = torch.empty_like(want).fill_(0xA5)
        plan.remap(frames, got)                      # a batch of 3 in one launch
        assert torch.equal(got, want), mode
        assert torch.equal(plan.remap(frames[1]), want[1]), mode  # a single frame (fused fix list when short)


def test_double_unrotated_uses_row_weights_and_few_fixes():
    # source rows of 960 * 3 bytes are 16-byte multiples: windows can be staged by LDS-DMA (LEAN tiles)
    plan = H.pb_plan(Case("stitch_195_aligned", pano(512, 1024), dbl(480, 960, "equidistant", 195), mask=2))
    info = plan.info()
    assert info["fix_tiles"] == 0 and info["fix_pixels"] < 200  # the truncation quirk at the eye edges is modelled
    assert info["lean_tiles"] > info["tiles"] // 2  # (counted per eye)


def test_double_unaligned_and_strided():
    """Source frames that are not 16-byte aligned cannot be windowed by LDS-DMA: the launch falls back to the
    separable / faithful kernels; strided batches keep their padding untouched."""
    case = CASES[2]
    plan = H.pb_plan(case)
    lib = nat.load()
    _, h, w, *_ = case.src
    H_, W_ = case.dst[1], case.dst[2]
    n_src, n_dst = h * w * 3, H_ * W_ * 3
    frames = _frames(case, 3)
    plan.set_mode(nat.MODE_FAITHFUL)
    want = plan.remap(frames).clone()
    plan.set_mode(nat.MODE_AUTO)
    for s_off, d_off in ((1, 4), (16, 1), (7, 3)):
        sbuf = torch.zeros(n_src + 64, dtype=torch.uint8, device="cuda")
        dbuf = torch.zeros(n_dst + 64, dtype=torch.uint8, device="cuda")
        sbuf[s_off : s_off + n_src] = frames[0].reshape(-1)
        nat.check(lib.pb_remap_u8(plan.handle, sbuf.data_ptr() + s_off, dbuf.data_ptr() + d_off, 1, 0, 0, nat.current_stream()))
        assert torch.equal(dbuf[d_off : d_off + n_dst].reshape(H_, W_, 3), want[0]), (s_off, d_off)
        assert int(dbuf[:d_off].sum()) == 0 and int(dbuf[d_off + n_dst :].sum()) == 0
    for pad in (48, 5):
        ss, ds = n_src + pad, n_dst + pad
        sbuf = torch.zeros(3 * ss + 64, dtype=torch.uint8, device="cuda")
        dbuf = torch.zeros(3 * ds + 64, dtype=torch.uint8, device="cuda")
        for f in range(3):
            sbuf[f * ss : f * ss + n_src] = frames[f].reshape(-1)
        nat.check(lib.pb_remap_u8(plan.handle, sbuf.data_ptr(), dbuf.data_ptr(), 3, ss, ds, nat.current_stream()))
        for f in range(3):
            assert torch.equal(dbuf[f * ds : f * ds + n_dst].reshape(H_, W_, 3), want[f]), (pad, f)
            assert int(dbuf[f * ds + n_dst : (f + 1) * ds].sum()) == 0
