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


# ---- the two-eye hot kernel at FULL size (c5 was pinned by bytes on the masked frame only) -------------
def _self_index_frames(h, w, eye):
    """Two frames whose texels encode their own linear index + 1 (three bytes in frame 0, the fourth in frame 1's red
    channel); eye = 'l' / 'r' keeps one eye and zeroes the other half of the side-by-side frame."""
    lin = torch.arange(1, h * w + 1, dtype=torch.int64, device="cuda").reshape(h, w)
    keep = torch.zeros((h, w), dtype=torch.bool, device="cuda")
    if eye == "l":
        keep[:, : w // 2] = True
    else:
        keep[:, w // 2 :] = True
    lin = torch.where(keep, lin, torch.zeros_like(lin)).reshape(-1)
    lo = torch.stack([lin & 255, (lin >> 8) & 255, (lin >> 16) & 255], dim=1).to(torch.uint8).reshape(h, w, 3)
    hi = torch.stack([(lin >> 24) & 255, torch.zeros_like(lin), torch.zeros_like(lin)], dim=1).to(torch.uint8).reshape(h, w, 3)
    return torch.stack([lo, hi])


def double_taps_through_hot_kernel(plan, h, w):
    """(left, right) int32 (H, W) taps as the two-eye HOT kernel sampled them, valid where that eye's blend factor is exactly
    1.0: with the other eye's half of the frame zeroed the blend (l * 1.0 + 0 * fr).astype(uint8) returns the live eye's texel,
    and a texel that spells its own index gives the index back (0 = the tap was black or out of bounds -> -1)."""
    taps = []
    plan.set_mode(nat.MODE_FAST)
    for eye in ("l", "r"):
        out = plan.remap(_self_index_frames(h, w, eye)).to(torch.int64)
        rec = out[0, ..., 0] | (out[0, ..., 1] << 8) | (out[0, ..., 2] << 16) | (out[1, ..., 0] << 24)
        taps.append((rec - 1).to(torch.int32))
        del out
    plan.set_mode(nat.MODE_AUTO)
    return taps


def _check_taps(plan, h, w):
    Hd, Wd = plan.dst.height, plan.dst.width
    idx, wts = plan.index_map(weights=True)
    idx = idx.reshape(2, Hd, Wd)
    wts = wts.reshape(2, Hd, Wd)
    got = double_taps_through_hot_kernel(plan, h, w)
    n_checked = 0
    for e in (0, 1):
        # a tap counts where its factor is exactly 1 and the OTHER eye's factor is finite (0 * inf = NaN -> 0 in the cast)
        unit = (wts[e] == 1.0) & torch.isfinite(wts[1 - e])
        bad = int(((got[e] != idx[e]) & unit).sum())
        assert bad == 0, f"eye {e}: {bad} taps of the hot kernel differ from pb_index_map_i32"
        n_checked += int(unit.sum())
    return n_checked, idx


@pytest.mark.parametrize("name", ["c5_180", "c5_195"])
def test_full_size_c5_taps_of_the_hot_double_kernel_and_raw_pins(name):
    """The indices of BOTH eyes recovered from the benchmarked two-eye kernel itself (one-eye and two-eye tiles alike) equal
    the index-map launch and the reference's SHA-256; the UNMASKED frame - data outside the circles, where the two samples add
    and wrap - reproduces the reference's bytes; one 4-frame launch equals four single launches and the golden on frame 0."""
    import hashlib

    import numpy as np
    from tests.cases import full_cases

    case = [c for c in full_cases() if c.name == name][0]
    pin = H.load_full()[name]
    plan = H.pb_plan(case)
    _, h, w, *_ = case.src
    n_checked, idx = _check_taps(plan, h, w)
    assert n_checked > 2 * 0.9 * plan.dst.height * plan.dst.width  # the merge band is thin: nearly every tap is checked
    sha = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()
    assert sha(idx[0]) == pin["idx_l_sha256"] and sha(idx[1]) == pin["idx_r_sha256"]
    del idx
    # unmasked frame: the reference's bytes
    raw = nat.synth_frame(h, w, frame=0, seed=0, circle_mask=0)
    assert sha(raw) == pin["raw_frame_sha256"]
    out = plan.remap(raw)
    got = out.reshape(-1, 3).cpu().numpy()
    pos = np.random.default_rng(pin["sample_seed"]).integers(0, got.shape[0], size=65536)
    assert np.array_equal(got[pos[:2048]].ravel(), np.array(pin["raw_u8_samples"], dtype=np.uint8))
    assert int(np.count_nonzero(got)) == pin["raw_nonzero_bytes"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == pin["raw_u8_sha256"]
    del got, out, raw
    # one 4-frame launch == four single launches; frame 0 == the golden
    frames = torch.stack([nat.synth_frame(h, w, frame=f, seed=0, circle_mask=case.mask) for f in range(4)])
    batched = plan.remap(frames)
    for f in range(4):
        assert torch.equal(plan.remap(frames[f]), batched[f]), f
    assert hashlib.sha256(batched[0].cpu().numpy().tobytes()).hexdigest() == pin["u8_sha256"]
    assert not torch.equal(batched[0], batched[1])


@pytest.mark.parametrize("case", [CASES[1], CASES[4], CASES[6], CASES[8]], ids=lambda c: c.name)
def test_taps_of_the_hot_double_kernel_mid_size(case):
    """The same recovery at 0.5 K with rotations (latitude-table weights), fisheye and double-fisheye destinations."""
    plan = H.pb_plan(case)
    _, h, w, *_ = case.src
    if (h * w * 3) % 16:
        pytest.skip("the windowed kernel needs 16-byte aligned frames")
    n_checked, _ = _check_taps(plan, h, w)
    assert n_checked > 0


def test_separable_fallback_is_verified_behind_the_first_launch_and_launches_capture_into_a_graph():
    """Round 4: the exhaustive check of a stitch plan's separable tables no longer runs at plan creation.  The first launch that wants the
    separable kernel (here: PB_MODE_FAST_DIRECT) enqueues the check and runs the float64 kernel; once the check has passed, later
    launches take the separable kernel - every launch returns the same bytes.  No launch allocates or synchronises, so launches capture
    into a graph: inside a capture an unverified plan simply takes the float64 kernel."""
    case = Case("stitch_195_lazy", pano(256, 512), dbl(240, 480, "equidistant", 195), mask=2)
    frames = _frames(case, 2)
    ref_plan = H.pb_plan(case)
    ref_plan.set_mode(nat.MODE_FAITHFUL)
    want = ref_plan.remap(frames).clone()
    # (a) a fresh plan, captured before anything verified its tables: the graph holds the float64 kernel
    plan = H.pb_plan(case)
    plan.set_mode(nat.MODE_FAST_DIRECT)
    out = torch.zeros_like(want)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        plan.remap(frames, out)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    # (b) eager launches: the first enqueues the check, the ones after a synchronisation run the separable kernel
    for rep in range(3):
        got = plan.remap(frames)
        torch.cuda.synchronize()
        assert torch.equal(got, want), rep
    # (c) the default mode, captured and replayed: the windowed two-eye kernel
    plan.set_mode(nat.MODE_AUTO)
    out.zero_()
    graph2 = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(graph2, stream=side):
        plan.remap(frames, out)
    graph2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
