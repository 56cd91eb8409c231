"""Host logic of the ndarray path (photonbend_amd/_device.py) on the CPU, against a stand-in for the library's plumbing entry points
(pb_malloc / pb_free / pb_host_alloc / pb_host_free / pb_host_register / pb_host_unregister): which caller arrays get page-locked and
when they are released again, how result blocks are recycled, how device views share one allocation.  No GPU, no compute."""

import ctypes as C
import gc

import numpy as np
import pytest

from photonbend_amd import _device


class FakeLib:
    """Counts calls; 'device' and 'page-locked' memory are plain malloc blocks."""

    def __init__(self):
        self.libc = C.CDLL(None)
        self.libc.malloc.restype = C.c_void_p
        self.libc.malloc.argtypes = [C.c_size_t]
        self.libc.free.argtypes = [C.c_void_p]
        self.live_dev, self.live_host, self.registered = set(), set(), {}
        self.register_calls = self.unregister_calls = self.host_allocs = 0

    def pb_malloc(self, out, n):
        p = self.libc.malloc(max(1, n))
        out._obj.value = p
        self.live_dev.add(p)
        return 0

    def pb_free(self, p):
        self.live_dev.discard(int(p))
        self.libc.free(int(p))
        return 0

    def pb_host_alloc(self, out, n):
        p = self.libc.malloc(max(1, n))
        out._obj.value = p
        self.live_host.add(p)
        self.host_allocs += 1
        return 0

    def pb_host_free(self, p):
        self.live_host.discard(int(p))
        self.libc.free(int(p))
        return 0

    def pb_host_register(self, p, n):
        assert int(p) not in self.registered, "registered twice"
        self.registered[int(p)] = int(n)
        self.register_calls += 1
        return 0

    def pb_host_unregister(self, p):
        assert int(p) in self.registered, "unregistering what was never registered"
        del self.registered[int(p)]
        self.unregister_calls += 1
        return 0


@pytest.fixture
def fake(monkeypatch):
    lib = FakeLib()
    monkeypatch.setattr(_device, "_lib", lambda: lib)
    return lib


def test_a_buffer_is_page_locked_on_its_second_sighting_and_released_with_its_owner(fake):
    reg = _device._Registrations(max_count=3, max_bytes=64 << 20)
    buf = np.zeros((1024, 1024, 3), np.uint8)  # 3 MiB, owns its memory
    assert not reg.is_registered(buf) and fake.register_calls == 0  # first sighting: staged copy
    assert reg.is_registered(buf) and fake.registered == {buf.ctypes.data: buf.nbytes}  # second: registered in place
    assert reg.is_registered(buf[100:200]) and fake.register_calls == 1  # a view inside the registration: direct too
    addr = buf.ctypes.data
    del buf
    gc.collect()
    assert addr not in fake.registered and fake.unregister_calls == 1  # the finaliser ran BEFORE the memory went away


def test_a_frame_sized_buffer_is_page_locked_on_its_first_sighting(fake):
    reg = _device._Registrations(max_count=3, max_bytes=1 << 30)
    frame = np.zeros((3072, 4096, 3), np.uint8)  # 36 MiB: page-locking in place (0.2-0.35 ms per 100 MB, measured) beats the staged copy at once
    assert frame.nbytes >= reg.FIRST_SIGHT_BYTES
    assert reg.is_registered(frame) and fake.registered == {frame.ctypes.data: frame.nbytes} and fake.register_calls == 1
    assert reg.is_registered(frame) and fake.register_calls == 1
    below = np.zeros(reg.FIRST_SIGHT_BYTES - 1, np.uint8)
    assert not reg.is_registered(below) and fake.register_calls == 1  # one byte below the floor: the second-sighting rule
    addr = frame.ctypes.data
    del frame
    gc.collect()
    assert addr not in fake.registered


def test_memory_without_an_owning_ndarray_and_small_arrays_are_never_registered(fake):
    reg = _device._Registrations()
    raw = bytearray(4 << 20)
    view = np.frombuffer(raw, dtype=np.uint8)  # the bytearray owns the memory: nothing to hang a finaliser on
    for _ in range(3):
        assert not reg.is_registered(view)
    small = np.zeros(1000, np.uint8)
    for _ in range(3):
        assert not reg.is_registered(small)
    assert fake.register_calls == 0


def test_registrations_are_evicted_least_recently_used_first(fake):
    reg = _device._Registrations(max_count=2, max_bytes=64 << 20)
    bufs = [np.zeros(2 << 20, np.uint8) for _ in range(3)]
    for b in bufs[:2]:
        reg.is_registered(b)
        assert reg.is_registered(b)
    assert reg.is_registered(bufs[0])  # touch 0: 1 is now the oldest
    reg.is_registered(bufs[2])
    assert reg.is_registered(bufs[2])  # a third buffer: room is made
    reg.settle()  # (an eviction's release runs on the worker thread, which holds the evicted array until it is through)
    assert set(fake.registered) == {bufs[0].ctypes.data, bufs[2].ctypes.data}
    assert not reg.is_registered(bufs[1]) or bufs[1].ctypes.data in fake.registered  # (1 was dropped; it may register again later)


def test_a_registration_that_no_longer_describes_its_owner_is_dropped(fake):
    reg = _device._Registrations()
    buf = np.zeros(2 << 20, np.uint8)
    reg.is_registered(buf)
    assert reg.is_registered(buf)
    base = buf.ctypes.data
    other = np.zeros(4, np.uint8)
    import weakref

    reg._reg[base] = (buf.nbytes, weakref.ref(other))  # as if the block had been freed and handed to another array
    assert reg.is_registered(buf) and fake.unregister_calls == 1 and fake.register_calls == 2  # stale one dropped, registered afresh
    reg._reg[base] = (buf.nbytes // 2, weakref.ref(buf))  # ... or the owner's extent had changed
    assert reg.is_registered(buf) and fake.unregister_calls == 2 and fake.registered == {base: buf.nbytes}


def test_result_blocks_are_recycled_and_never_shared_while_alive(fake):
    pool = _device._PinnedPool(keep_bytes=64 << 20)
    a = pool.ndarray((512, 512, 3), np.uint8)
    a[...] = 7
    b = pool.ndarray((512, 512, 3), np.uint8)
    assert a.ctypes.data != b.ctypes.data and fake.host_allocs == 2 and a.flags.writeable
    pa = a.ctypes.data
    view = a[10:20]
    del a
    gc.collect()
    c = pool.ndarray((512, 512, 3), np.uint8)
    assert c.ctypes.data not in (pa, b.ctypes.data) and int(view[0, 0, 0]) == 7  # a view keeps the block out of the pool
    del view, c
    gc.collect()
    d = pool.ndarray((512, 512, 3), np.uint8)
    assert fake.host_allocs == 3 and d.ctypes.data in fake.live_host  # recycled: no fourth allocation
    assert pool._capacity(0) == 4096 and pool._capacity(1) == 65536 and pool._capacity(65537) == 131072


def test_a_full_pool_frees_instead_of_keeping(fake):
    pool = _device._PinnedPool(keep_bytes=1 << 20)
    a = pool.ndarray((2 << 20,), np.uint8)
    p = a.ctypes.data
    del a
    gc.collect()
    assert p not in fake.live_host  # larger than the pool may keep: freed


def test_device_array_views_share_one_allocation(fake):
    d = _device.DeviceArray((4, 8, 16, 3), np.uint8)
    assert d.nbytes == 4 * 8 * 16 * 3 and len(fake.live_dev) == 1
    f2 = d[2]
    assert f2.shape == (8, 16, 3) and f2.data_ptr() == d.data_ptr() + 2 * 8 * 16 * 3
    assert d[1:3].shape == (2, 8, 16, 3) and d[-1].data_ptr() == d.data_ptr() + 3 * 384
    as_f32 = d.view(np.float32, (4 * 8 * 16 * 3 // 4,))
    assert as_f32.dtype == np.float32 and as_f32.data_ptr() == d.data_ptr()
    with pytest.raises(ValueError):
        d.view(np.float64, (5,))
    with pytest.raises(IndexError):
        d[4]
    cai = d.__cuda_array_interface__
    assert cai["shape"] == (4, 8, 16, 3) and cai["typestr"] == "|u1" and cai["data"] == (d.data_ptr(), False)
    del d, f2
    gc.collect()
    assert len(fake.live_dev) == 1  # the float view still holds the block
    del as_f32
    gc.collect()
    assert not fake.live_dev


# ---- the streaming host pipeline over the same stand-in ----------------------------------------------------------------
class FakePipeLib(FakeLib):
    """Adds synchronous copies, streams and events that only count; a 'launch' is a host function over the fake device memory."""

    def __init__(self):
        super().__init__()
        self.n_streams = self.n_events = 0
        self.log = []

    def pb_memcpy_h2d(self, dst, src, n, stream):
        C.memmove(int(dst), int(src), int(n))
        self.log.append(("h2d", int(stream)))
        return 0

    def pb_memcpy_d2h(self, dst, src, n, stream):
        C.memmove(int(dst), int(src), int(n))
        self.log.append(("d2h", int(stream)))
        return 0

    def pb_stream_create(self, out):
        self.n_streams += 1
        out._obj.value = 0x1000 + self.n_streams
        return 0

    def pb_event_create(self, out):
        self.n_events += 1
        out._obj.value = 0x2000 + self.n_events
        return 0

    def pb_stream_sync(self, s):
        return 0

    def pb_stream_destroy(self, s):
        return 0

    def pb_event_destroy(self, e):
        return 0

    def pb_event_record(self, e, s):
        return 0

    def pb_event_sync(self, e):
        return 0

    def pb_stream_wait_event(self, s, e):
        self.log.append(("wait", int(s)))
        return 0


class FakePlan:
    """Stands where a native Plan would: 'remaps' by flipping the frame upside down, on the fake device memory."""

    class _Dims:
        def __init__(self, h, w):
            self.height, self.width = h, w

    def __init__(self, h, w, lib):
        self.src = self.dst = FakePlan._Dims(h, w)
        self.lib = lib
        self.launches = []

    def launch(self, src_ptr, dst_ptr, n_frames, stream, interpolation="nearest"):
        n = self.src.height * self.src.width * 3
        a = np.frombuffer((C.c_ubyte * n).from_address(src_ptr), np.uint8).reshape(self.src.height, self.src.width, 3)
        out = np.frombuffer((C.c_ubyte * n).from_address(dst_ptr), np.uint8).reshape(a.shape)
        out[...] = a[::-1]
        self.launches.append((int(stream), interpolation))
        self.lib.log.append(("run", int(stream)))


@pytest.fixture
def pipe_env(monkeypatch):
    from photonbend_amd import _hostpipe, _native

    lib = FakePipeLib()
    monkeypatch.setattr(_device, "_lib", lambda: lib)
    monkeypatch.setattr(_native, "load", lambda: lib)
    monkeypatch.setattr(_native, "require_gpu", lambda: None)
    monkeypatch.setattr(_native, "current_device", lambda: 0)
    import contextlib

    monkeypatch.setattr(_native, "on_device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(_device, "PINNED", _device._PinnedPool())
    monkeypatch.setattr(_hostpipe, "PINNED", _device.PINNED)
    monkeypatch.setattr(_hostpipe, "REGISTERED", _device._Registrations())
    monkeypatch.setattr(_hostpipe, "_TLS", __import__("threading").local())
    return lib, _hostpipe


@pytest.mark.parametrize("n_frames,depth", [(0, 3), (1, 3), (3, 3), (7, 3), (5, 2), (4, 1)])
def test_streamed_frames_come_back_in_order_whatever_the_depth(pipe_env, n_frames, depth):
    lib, hp = pipe_env
    rng = np.random.default_rng(n_frames * 10 + depth)
    frames = [rng.integers(0, 256, (24, 40, 3), dtype=np.uint8) for _ in range(n_frames)]
    plan = FakePlan(24, 40, lib)
    outs = list(hp.remap_frames(plan, iter(frames), depth=depth))
    assert len(outs) == n_frames and len(plan.launches) == n_frames
    for f, o in zip(frames, outs):
        assert o.dtype == np.uint8 and np.array_equal(o, f[::-1])
    # two distinct streams: uploads and launches never share one; there is NO download - the launch's destination is the result
    # ndarray itself (page-locked, device-visible: round 6, experiments/r6/pcie_paths.py)
    streams = {kind: {s for k, s in lib.log if k == kind} for kind in ("h2d", "run", "d2h")}
    if n_frames:
        assert len(streams["h2d"]) == 1 and len(streams["run"]) == 1 and streams["h2d"] != streams["run"] and not streams["d2h"]
    del outs
    gc.collect()


def test_streamed_results_outlive_the_generator_and_slots_are_not_overwritten(pipe_env):
    lib, hp = pipe_env
    frames = [np.full((16, 16, 3), k, np.uint8) for k in range(9)]
    plan = FakePlan(16, 16, lib)
    kept = []
    for out in hp.remap_frames(plan, frames, depth=2):
        kept.append(out)  # the caller holds every result while later frames reuse the device slots
    assert [int(o[0, 0, 0]) for o in kept] == list(range(9))
    assert len({o.ctypes.data for o in kept}) == 9  # nine live results, nine distinct page-locked blocks


def test_the_pipelines_device_ring_is_kept_between_calls_and_a_caller_may_stop_early(pipe_env):
    """Round 6: remap_frames checks its rotating device input buffers out of the thread's pipe and returns them (three hipMallocs and
    hipFrees of 100 MB per call otherwise); two pipelines alive at once never share a buffer; a caller that stops iterating leaves
    nothing in flight (the generator's exit waits for both streams) and the ring goes back."""
    lib, hp = pipe_env
    frames = [np.full((16, 16, 3), k, np.uint8) for k in range(6)]
    plan = FakePlan(16, 16, lib)
    assert [int(o[0, 0, 0]) for o in hp.remap_frames(plan, frames, depth=3)] == list(range(6))
    ring = len(lib.live_dev)
    assert ring == 3
    assert [int(o[0, 0, 0]) for o in hp.remap_frames(plan, frames, depth=3)] == list(range(6))
    assert len(lib.live_dev) == ring  # the same three buffers served the second call
    g1, g2 = hp.remap_frames(plan, frames, depth=2), hp.remap_frames(plan, list(reversed(frames)), depth=2)
    got = [(int(next(g1)[0, 0, 0]), int(next(g2)[0, 0, 0])) for _ in range(3)]
    assert got == [(0, 5), (1, 4), (2, 3)]
    assert len(lib.live_dev) == ring + 1  # 2 + 2 buffers checked out, three of them from the idle ring
    g1.close()  # stopped early
    g2.close()
    assert len(lib.live_dev) <= 4  # at most four idle buffers of a size are kept
    assert [int(o[0, 0, 0]) for o in hp.remap_frames(plan, frames, depth=3)] == list(range(6))
    gc.collect()


def test_streaming_rejects_frames_of_the_wrong_shape_or_type(pipe_env):
    lib, hp = pipe_env
    plan = FakePlan(16, 16, lib)
    with pytest.raises(ValueError):
        list(hp.remap_frames(plan, [np.zeros((16, 15, 3), np.uint8)]))
    with pytest.raises(ValueError):
        list(hp.remap_frames(plan, [np.zeros((16, 16, 3), np.float32)]))


def test_single_frame_path_reuses_its_device_buffers_and_uploads_a_refilled_buffer_directly(pipe_env):
    lib, hp = pipe_env
    plan = FakePlan(600, 800, lib)  # 1.4 MB: above the registration floor
    buf = np.zeros((600, 800, 3), np.uint8)
    for k in range(4):
        buf[...] = k + 1
        out = hp.remap_ndarray(plan, buf)
        assert np.array_equal(out, buf[::-1])
    assert len(lib.live_dev) == 2  # one input and one output buffer, kept between calls
    assert lib.register_calls == 1 and buf.ctypes.data in lib.registered  # page-locked in place on its second sighting
    h2d = [k for k, _ in lib.log if k == "h2d"]
    assert len(h2d) == 1 + 3  # first call staged in one chunk, then one direct DMA per call
    with pytest.raises(ValueError):
        hp.pipe_for(0).upload(np.zeros(10, np.uint8), _device.DeviceArray((12,), np.uint8))
