"""Plan behaviour on the GPU: the fast path (tile models + fix list) and the faithful path produce the
same bytes; plan statistics are sane; batches, strides and unaligned frames take the right kernels."""

import numpy as np
import pytest
import torch

from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import Case, cam, case_by_name, dbl, inscribed, pano, small_cases

pytestmark = pytest.mark.gpu


def test_plan_info_and_modes_c2_like():
    case = Case("mid", cam(1024, 1024, "equidistant", 360, inscribed(1024)), pano(1024, 2048))
    plan = H.pb_plan_private(case)
    plan.set_window_budget(12288)  # the statistics below are those of the largest window budget
    info = plan.info()
    assert info["fast_path"] and info["tiles"] == 32 * 32
    assert 0 <= info["fix_tiles"] <= 16 and 0 <= info["fix_pixels"] < 1024 * 1024 // 100
    assert info["lean_tiles"] + info["direct_tiles"] + info["black_tiles"] + info["fix_tiles"] <= info["tiles"]
    assert info["lean_tiles"] > info["tiles"] // 3 and info["black_tiles"] > 0
    # thresholds: invalid <=> lo <= (2x)^2+(2y)^2 < hi; the inscribed circle of diameter 1023 px is valid
    lo, hi = info["thresholds"][:2]
    assert 1023 * 1023 <= lo <= 1023 * 1023 + 4 * 1024 and hi > lo
    frame = nat.synth_frame(1024, 2048, frame=3)
    outs = {}
    for mode in (nat.MODE_FAITHFUL, nat.MODE_FAST, nat.MODE_AUTO, nat.MODE_FAST_DIRECT):
        plan.set_mode(mode)
        assert plan.info()["fast_path"] == (mode != nat.MODE_FAITHFUL)
        outs[mode] = (plan.remap(frame).clone(), plan.index_map().clone())
    for mode in (nat.MODE_FAST, nat.MODE_AUTO, nat.MODE_FAST_DIRECT):
        assert torch.equal(outs[mode][0], outs[nat.MODE_FAITHFUL][0])
        assert torch.equal(outs[mode][1], outs[nat.MODE_FAITHFUL][1])


@pytest.mark.parametrize("name", ["A_photo_odd", "C_alter_eqd_eqs_rot", "B_pano_equisolid_360", "D_pano_chain", "E_double_dst_rot"])
def test_fast_equals_faithful_small(name):
    case = case_by_name(name)
    plan = H.pb_plan(case)
    frame = torch.from_numpy(H.case_frame(case)).cuda()
    plan.set_mode(nat.MODE_FAITHFUL)
    ref = plan.remap(frame).clone()
    plan.set_mode(nat.MODE_FAST)
    assert torch.equal(plan.remap(frame), ref)


def test_unaligned_frames_and_strided_batches():
    """Frame pointers that are not 16-byte aligned take the direct-gather hot kernel + fix kernel; the
    bytes must not change.  Same for a batch addressed through explicit strides."""
    import ctypes

    case = Case("mid", cam(512, 512, "equisolid", 200, inscribed(512)), pano(512, 1024), [(10, 20, 30)])
    plan = H.pb_plan(case)
    n_src, n_dst = 512 * 1024 * 3, 512 * 512 * 3
    frames = [nat.synth_frame(512, 1024, frame=f) for f in range(3)]
    want = [plan.remap(f).clone() for f in frames]
    lib = nat.load()
    # 1) source frame at an odd byte offset, destination at an offset of 4 (store alignment kept) and of 1
    for s_off, d_off in ((1, 4), (16, 1), (7, 3)):
        sbuf = torch.zeros(n_src + 64, dtype=torch.uint8, device="cuda")
        dbuf = torch.zeros(n_dst + 64, dtype=torch.uint8, device="cuda")
        sbuf[s_off : s_off + n_src] = frames[0].reshape(-1)
        nat.check(lib.pb_remap_u8(plan.handle, sbuf.data_ptr() + s_off, dbuf.data_ptr() + d_off, 1, 0, 0, nat.current_stream()))
        assert torch.equal(dbuf[d_off : d_off + n_dst].reshape(512, 512, 3), want[0]), (s_off, d_off)
        assert int(dbuf[:d_off].sum()) == 0 and int(dbuf[d_off + n_dst :].sum()) == 0  # nothing written outside
    # 2) batch of 3 with padded strides (multiples of 16 -> windowed kernel; odd -> direct kernel)
    for pad in (48, 5):
        ss, ds = n_src + pad, n_dst + pad
        sbuf = torch.zeros(3 * ss + 64, dtype=torch.uint8, device="cuda")
        dbuf = torch.zeros(3 * ds + 64, dtype=torch.uint8, device="cuda")
        for f in range(3):
            sbuf[f * ss : f * ss + n_src] = frames[f].reshape(-1)
        nat.check(lib.pb_remap_u8(plan.handle, sbuf.data_ptr(), dbuf.data_ptr(), 3, ss, ds, nat.current_stream()))
        for f in range(3):
            assert torch.equal(dbuf[f * ds : f * ds + n_dst].reshape(512, 512, 3), want[f]), (pad, f)
            assert int(dbuf[f * ds + n_dst : (f + 1) * ds].sum()) == 0


def test_every_small_case_has_consistent_plan_stats():
    for case in small_cases():
        info = H.pb_plan(case).info()
        # double-fisheye sources carry one certified tile table per eye, rotated or not
        assert info["fast_path"] and info["tiles"] > 0 and info["fix_pixels"] >= 0


def test_streaming_batch_overlaps_and_matches():
    """batch.remap_frames (upload DMA of frame k + 1 beside the remap of frame k into its page-locked result) == the per-frame facade, in order."""
    import photonbend_amd as pb
    from photonbend_amd import batch
    from oracle.synth import synth_frame

    fov = pb.utils.to_radians(200)
    dst = pb.CameraImage(np.zeros((200, 200, 3), np.uint8), fov, pb.equisolid(), magnitude=99.5)
    rot = pb.Rotation(0.3, -0.2, 0.5)
    frames = [synth_frame(160, 320, frame=f) for f in range(7)]
    src0 = pb.PanoramaImage(frames[0])
    plan = batch.plan_for(dst, [rot], src0)
    got = list(batch.remap_frames(plan, iter(frames), depth=3))
    assert len(got) == 7
    for f, out in zip(frames, got):
        want = pb.PanoramaImage(f).process_coordinate_map(rot.rotate_coordinate_map(dst.get_coordinate_map()))
        assert np.array_equal(out, want)
    assert list(batch.remap_frames(plan, iter([]))) == []


@pytest.mark.parametrize("name,interpolation", [("c2", "nearest"), ("c3", "nearest"), ("c5_195", "nearest"), ("c3", "bilinear"), ("c5_195", "bilinear")])
def test_streamed_results_written_over_pcie_equal_the_device_results_at_full_size(name, interpolation):
    """Round 6: batch.remap_frames has the remap kernel store straight into the page-locked result ndarray (no device output buffer, no
    download DMA).  At BASELINE size - failed tiles, fix pixels re-stored by their tile's wave, the pair waves of the stitch, both
    samplers - every streamed frame equals the same frame remapped into device memory, byte for byte; frame-sized inputs the library has
    never seen are page-locked in place and uploaded with one DMA each."""
    import torch

    from photonbend_amd import _device, batch
    from tests.cases import full_cases

    case = next(c for c in full_cases() if c.name == name)
    plan = H.pb_plan_private(case)
    _, h, w, *_ = case.src
    dev_frames = [nat.synth_frame(h, w, frame=40 + f, seed=2, circle_mask=case.mask) for f in range(3)]
    host_frames = [f.cpu().numpy().copy() for f in dev_frames]  # fresh ndarrays that own their memory
    before = len(_device.REGISTERED._reg)
    got = list(batch.remap_frames(plan, iter(host_frames), depth=2, interpolation=interpolation))
    assert len(got) == 3 and len(_device.REGISTERED._reg) >= min(before + 3, _device.REGISTERED._max_count) - 1
    for f, out in zip(dev_frames, got):
        want = plan.remap(f, interpolation=interpolation).cpu().numpy()
        assert out.shape == want.shape and np.array_equal(out, want), f"{name} {interpolation}: {int((out != want).any(axis=2).sum())} pixels differ"
    del got, host_frames
    torch.cuda.empty_cache()


CONCURRENT_CASES = [
    Case("k_c2like", cam(1024, 1024, "equidistant", 360, inscribed(1024)), pano(1024, 2048)),
    Case("k_rim", cam(768, 768, "equisolid", 360, inscribed(768)), cam(768, 768, "equidistant", 360, inscribed(768)), [(30, 45, 10)]),
    Case("k_double", pano(768, 1536), dbl(960, 1920, "equidistant", 195), [(3, 90, -7)]),
]


@pytest.mark.parametrize("case", CONCURRENT_CASES, ids=[c.name for c in CONCURRENT_CASES])
def test_one_plan_on_several_streams_at_once(case):
    """Independent frames dealt round-robin to three HIP streams (bench.py: wall_ms_per_frame_by_streams - the next launch's ramp
    runs in the previous one's drain): launches of ONE plan overlap on the device and every output equals the serial one."""
    plan = H.pb_plan(case)
    lib = nat.load()
    _, h, w, *_ = case.src
    n = 12
    frames = torch.stack([nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(n)])
    want = torch.stack([plan.remap(frames[f]) for f in range(n)])
    got = torch.zeros_like(want)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    sb, db = frames[0].numel(), want[0].numel()
    for rep in range(3):
        for f in range(n):
            nat.check(lib.pb_remap_u8(plan.handle, frames.data_ptr() + f * sb, got.data_ptr() + f * db, 1, 0, 0, int(streams[f % 3].cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(got, want)


GRAPH_CASES = [
    Case("g_pano", cam(256, 256, "equidistant", 360, inscribed(256)), pano(256, 512), [(5, 10, 15)]),
    # a rim of failed tiles (360-degree equisolid destination): served from the plan's exact-index tables
    Case("g_rim", cam(256, 256, "equisolid", 360, inscribed(256)), cam(256, 256, "equidistant", 360, inscribed(256)), [(30, 45, 10)]),
    # double-fisheye source under a rotation: per-eye tables + latitude table
    Case("g_double", pano(192, 384), dbl(256, 512, "equidistant", 195), [(3, 90, -7)]),
]


@pytest.mark.parametrize("case", GRAPH_CASES, ids=[c.name for c in GRAPH_CASES])
def test_remap_is_graph_capturable(case):
    """Launch functions neither allocate nor synchronise, and a frame is ONE launch whatever the geometry: a
    burst of pb_remap_u8 calls captures into a HIP graph (torch.cuda.CUDAGraph) and replays with the same bytes."""
    plan = H.pb_plan(case)
    sh, sw = case.src[1], case.src[2]
    dh, dw = case.dst[1], case.dst[2]
    frames = [nat.synth_frame(sh, sw, frame=f) for f in range(4)]
    outs = [torch.zeros((dh, dw, 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
    want = [plan.remap(f).clone() for f in frames]
    lib = nat.load()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for f in range(4):
                nat.check(lib.pb_remap_u8(plan.handle, frames[f].data_ptr(), outs[f].data_ptr(), 1, 0, 0, int(side.cuda_stream)))
    torch.cuda.current_stream().wait_stream(side)
    for o in outs:
        o.zero_()
    g.replay()
    torch.cuda.synchronize()
    for f in range(4):
        assert torch.equal(outs[f], want[f])
    # new pixels in the same buffers, replay again
    for f in range(4):
        frames[f].copy_(nat.synth_frame(sh, sw, frame=10 + f))
    want2 = [plan.remap(f).clone() for f in frames]
    g.replay()
    torch.cuda.synchronize()
    for f in range(4):
        assert torch.equal(outs[f], want2[f])


@pytest.mark.parametrize("case", GRAPH_CASES, ids=[c.name for c in GRAPH_CASES])
def test_bilinear_remap_is_graph_capturable(case):
    """The opt-in bilinear launch is ONE kernel over the plan's own launch table (windows, half windows, coordinate tables: all built at
    plan preparation): pb_remap_bilinear_u8 neither allocates nor synchronises, captures into a HIP graph - a burst of single frames and
    a batch of two - and replays with the same bytes on new pixels."""
    plan = H.pb_plan_private(case)
    sh, sw = case.src[1], case.src[2]
    dh, dw = case.dst[1], case.dst[2]
    frames = torch.stack([nat.synth_frame(sh, sw, frame=f) for f in range(4)])
    outs = torch.zeros((4, dh, dw, 3), dtype=torch.uint8, device="cuda")
    want = plan.remap(frames, interpolation="bilinear").clone()
    lib = nat.load()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for f in range(2):
                nat.check(lib.pb_remap_bilinear_u8(plan.handle, frames[f].data_ptr(), outs[f].data_ptr(), 1, 0, 0, int(side.cuda_stream)))
            nat.check(lib.pb_remap_bilinear_u8(plan.handle, frames[2].data_ptr(), outs[2].data_ptr(), 2, 0, 0, int(side.cuda_stream)))
    torch.cuda.current_stream().wait_stream(side)
    outs.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(outs, want)
    frames.copy_(torch.stack([nat.synth_frame(sh, sw, frame=10 + f) for f in range(4)]))
    want2 = plan.remap(frames, interpolation="bilinear").clone()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(outs, want2)


def test_remap_batch_sharded_single_process():
    """parallel.remap_batch_sharded without a process group = one shard holding every frame, launched in chunks."""
    import photonbend_amd as pb
    from photonbend_amd import parallel

    fov = pb.utils.to_radians(180)
    dst = pb.CameraImage(np.zeros((96, 96, 3), np.uint8), fov, pb.equidistant(), magnitude=47.5)
    rot = pb.Rotation(0.2, 0.1, -0.4)
    srcp = nat.make_proj(nat.KIND_PANO, 64, 128)
    ids, outs = parallel.remap_batch_sharded(dst._proj("dst"), [rot.rotation_matrix], srcp, lambda i: nat.synth_frame(64, 128, frame=i), 11, chunk=4)
    assert ids == list(range(11)) and len(outs) == 11
    for i in (0, 5, 10):
        frame = nat.synth_frame(64, 128, frame=i).cpu().numpy()
        want = pb.PanoramaImage(frame).process_coordinate_map(rot.rotate_coordinate_map(dst.get_coordinate_map()))
        assert np.array_equal(outs[i].cpu().numpy(), want)


BUDGET_CASES = [
    Case("b_pano", cam(768, 768, "equidistant", 360, inscribed(768)), pano(512, 1024), [(10, 20, 30)]),
    Case("b_cam", pano(512, 1024), cam(768, 768, "equidistant", 360, inscribed(768))),
    Case("b_double", pano(512, 1024), dbl(480, 960, "equidistant", 195), [(3, 90, -7)]),
]


@pytest.mark.parametrize("case", BUDGET_CASES, ids=[c.name for c in BUDGET_CASES])
def test_window_budget_only_moves_tiles_between_paths(case):
    """The per-plan LDS window budget (pb_plan_create_ex / pb_plan_set_window_budget) decides which PATH a tile
    takes - LDS window or direct gather - never its pixels: every budget reproduces the faithful bytes, in the
    nearest and (where supported) the bilinear mode; the same plan object is re-budgeted in place."""
    frames = torch.stack([nat.synth_frame(case.src[1], case.src[2], frame=f) for f in range(2)])
    want = None
    b0 = None
    leans = []
    plan = H.pb_plan_private(case, budget=12288)
    for budget in (12288, 8176, 6144, 4224):
        plan.set_window_budget(budget)
        info = plan.info()
        assert info["window_budget"] == budget
        leans.append(info["lean_tiles"])
        if want is None:
            plan.set_mode(nat.MODE_FAITHFUL)
            want = plan.remap(frames).clone()
            plan.set_mode(nat.MODE_AUTO)
        assert torch.equal(plan.remap(frames), want), budget
        assert torch.equal(plan.remap(frames[1]), want[1]), budget
        if case.src[0] != "double":
            got = plan.remap(frames[0], interpolation="bilinear")
            if b0 is None:
                b0 = got.clone()
            assert torch.equal(got, b0), budget
    assert leans == sorted(leans, reverse=True) and leans[0] > leans[-1]  # smaller windows: fewer LEAN tiles


BATCH_CASES = [
    Case("bt_pano", cam(200, 232, "equisolid", 200, 99.5), pano(160, 320), [(10, 20, 30)]),  # ragged grid: 7 x 8 tiles, not a multiple of anything
    Case("bt_cam", pano(192, 384), cam(256, 256, "equidistant", 360, inscribed(256)), mask=1),
    Case("bt_super", cam(512, 512, "equidistant", 360, inscribed(512)), pano(384, 768)),  # 8 x 8 tile groups: the super-tile launch order
    Case("bt_double", pano(160, 320), dbl(192, 384, "equidistant", 195), [(3, 90, -7)], mask=2),
]


@pytest.mark.parametrize("case", BATCH_CASES, ids=[c.name for c in BATCH_CASES])
def test_batches_are_a_grid_dimension_and_change_no_byte(case):
    """pb_remap_u8 with n_frames > 1 is ONE launch whose grid spans the frames (frame-major; double sources: chunks of
    frames): 37 distinct frames in one call, through explicit strides, equal 37 single-frame calls and the faithful
    kernel byte for byte - ragged grids, the super-tile launch order and the padded tail of a frame's workgroups included."""
    import ctypes

    n = 37
    _, h, w, *_ = case.src
    Hd, Wd = case.dst[1], case.dst[2]
    sb, db = h * w * 3, Hd * Wd * 3
    s_stride, d_stride = (sb + 48 + 15) // 16 * 16, db + 32  # frames apart from each other: strides larger than a frame
    src = torch.zeros(n * s_stride, dtype=torch.uint8, device="cuda")
    for f in range(n):
        src[f * s_stride : f * s_stride + sb] = nat.synth_frame(h, w, frame=f, circle_mask=case.mask).reshape(-1)
    plan = H.pb_plan(case)
    assert plan.info()["fast_path"]
    dst = torch.full((n * d_stride,), 0xAB, dtype=torch.uint8, device="cuda")
    nat.check(nat.load().pb_remap_u8(plan.handle, src.data_ptr(), dst.data_ptr(), n, s_stride, d_stride, nat.current_stream()))
    plan.set_mode(nat.MODE_FAITHFUL)
    for f in (0, 1, 17, 35, 36):
        frame = src[f * s_stride : f * s_stride + sb].reshape(h, w, 3)
        want = plan.remap(frame)
        assert torch.equal(dst[f * d_stride : f * d_stride + db].reshape(Hd, Wd, 3), want), f
        assert bool((dst[f * d_stride + db : (f + 1) * d_stride] == 0xAB).all()), "the gap between frames was written"
    plan.set_mode(nat.MODE_AUTO)
    for f in (5, 36):
        frame = src[f * s_stride : f * s_stride + sb].reshape(h, w, 3)
        assert torch.equal(plan.remap(frame), dst[f * d_stride : f * d_stride + db].reshape(Hd, Wd, 3))


@pytest.mark.parametrize("case", BATCH_CASES, ids=[c.name for c in BATCH_CASES])
def test_scattered_frames_in_one_launch_equal_single_launches(case):
    """pb_remap_u8v: a ring of SEPARATELY ALLOCATED frames (pointer tables, no common stride) in one launch per 64 frames - 70 frames
    (two launches: 64 + 6) land byte for byte where 70 single pb_remap_u8 calls put them; a deferred plan, the float64 mode and an
    unaligned source pointer take the frame-by-frame path and give the same bytes; bad arguments are refused."""
    import ctypes as C

    n = 70
    _, h, w, *_ = case.src
    Hd, Wd = case.dst[1], case.dst[2]
    # separately allocated, in a shuffled order, with decoys in between so that neighbouring frames are not at one stride
    rng = np.random.default_rng(5)
    order = rng.permutation(n)
    srcs, decoys = [None] * n, []
    for f in order:
        srcs[f] = nat.synth_frame(h, w, frame=int(f), circle_mask=case.mask)
        decoys.append(torch.empty(int(rng.integers(1, 5)) * 4096 + 16, dtype=torch.uint8, device="cuda"))
    outs = [torch.full((Hd, Wd, 3), 0xAB, dtype=torch.uint8, device="cuda") for _ in range(n)]
    plan = H.pb_plan_private(case)
    assert plan.info()["fast_path"]
    got = plan.remap_each(srcs, outs)
    assert all(g is o for g, o in zip(got, outs))
    for f in range(n):
        assert torch.equal(outs[f], plan.remap(srcs[f])), f
    # fresh outputs when none are given
    fresh = plan.remap_each(srcs[:3])
    assert all(torch.equal(a, b) for a, b in zip(fresh, outs[:3]))
    # the float64 mode and a deferred plan: frame by frame, same bytes
    plan.set_mode(nat.MODE_FAITHFUL)
    assert all(torch.equal(a, b) for a, b in zip(plan.remap_each(srcs[10:13]), outs[10:13]))
    plan.set_mode(nat.MODE_AUTO)
    lazy = H.pb_plan_private(case, defer=True)
    assert all(torch.equal(a, b) for a, b in zip(lazy.remap_each(srcs[20:22]), outs[20:22]))
    # an unaligned source pointer (LDS-DMA cannot address it): still the same bytes
    sb = h * w * 3
    raw = torch.empty(sb + 16, dtype=torch.uint8, device="cuda")
    raw[1 : 1 + sb] = srcs[7].reshape(-1)
    odd = raw[1 : 1 + sb].reshape(h, w, 3)
    assert odd.data_ptr() % 16 != 0
    mixed = plan.remap_each([srcs[6], odd, srcs[8]])
    assert torch.equal(mixed[0], outs[6]) and torch.equal(mixed[1], outs[7]) and torch.equal(mixed[2], outs[8])
    # the launch reads the pointer tables when it is ISSUED and neither allocates nor synchronises: a pb_remap_u8v call captures into a
    # HIP graph; the host tables may be overwritten once the call has returned, and a replay writes the same frames again
    n_g = 5
    g_out = [torch.zeros((Hd, Wd, 3), dtype=torch.uint8, device="cuda") for _ in range(n_g)]
    sp_g = (C.c_void_p * n_g)(*[s.data_ptr() for s in srcs[30 : 30 + n_g]])
    dp_g = (C.c_void_p * n_g)(*[o.data_ptr() for o in g_out])
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            nat.check(nat.load().pb_remap_u8v(plan.handle, sp_g, dp_g, n_g, int(side.cuda_stream)))
    for k in range(n_g):
        sp_g[k] = dp_g[k] = 0  # (the tables were read at capture time)
    torch.cuda.current_stream().wait_stream(side)
    for o in g_out:
        o.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(g_out[k], outs[30 + k]) for k in range(n_g))
    # argument checks
    L = nat.load()
    sp = (C.c_void_p * 2)(srcs[0].data_ptr(), 0)
    dp = (C.c_void_p * 2)(outs[0].data_ptr(), outs[1].data_ptr())
    assert L.pb_remap_u8v(plan.handle, sp, dp, 2, nat.current_stream()) == -1  # PB_ERR_INVALID
    assert L.pb_remap_u8v(plan.handle, sp, dp, -1, nat.current_stream()) == -1  # PB_ERR_INVALID
    assert L.pb_remap_u8v(plan.handle, None, None, 0, nat.current_stream()) == 0
    torch.cuda.synchronize()
    del decoys


def _tiny_cases():
    from tests.test_hip_bilinear import _TINY

    return _TINY


@pytest.mark.parametrize("case", _tiny_cases(), ids=lambda c: c.name)
def test_remap_on_sources_and_destinations_of_a_few_pixels(case):
    """The reference's (nearest) sampler on frames of 1-12 pixels - tiles larger than the source, windows that are the whole frame, a
    1 x 1 destination: byte-identical to the oracle (which is pinned to the reference on the 59 small cases)."""
    from oracle import reference_path as orc

    rng = np.random.default_rng(3)
    frame = rng.integers(0, 256, size=(case.src[1], case.src[2], 3), dtype=np.uint8)
    want = orc.remap(H.orc_proj(case.dst), H.orc_proj(case.src), frame, H.orc_rots(case))
    got = H.pb_plan_private(case).remap(torch.from_numpy(frame).cuda()).cpu().numpy()
    assert np.array_equal(got, want), f"{int((got != want).any(axis=2).sum())} pixels differ"
