"""NumPy frames <-> device: the host side of the drop-in's REAL path - the reference's contract is ndarray in, fresh
ndarray out (core/__init__.py:66-92), so for a user who just swaps imports a remap is upload + kernel + download.

What each direction costs and how it is kept down (measured on the MI355X box, c2 = 100.7 MB in, 50.3 MB out):
  download  the result ndarray IS page-locked memory from a recycling pool (``_device.PINNED``): the DMA writes it directly, no
            staging copy; the block returns to the pool when the caller drops the array.
  upload    memory the caller owns is pageable.  A frame-sized array (>= 32 MiB) is page-locked in place at its first sighting, a
            smaller one at its second (pb_host_register, tied to the owning object's lifetime: 0.2-0.35 ms for a c2 frame), and
            uploads with ONE DMA straight out of the caller's memory.  Anything else (views, small arrays seen once) is copied
            through page-locked staging buffers in 16 MiB chunks on a few threads, each chunk's DMA running while the next chunk
            is being copied.
  streams   (remap_frames) the two DMA directions do not run at full rate side by side on this box (1.77 ms up + 0.89 ms down take
            2.41 ms on two streams), but an upload DMA and a KERNEL that stores over PCIe do (2.06 ms): the remap kernel of frame k
            writes its output straight into the page-locked result ndarray while frame k + 1 uploads - no device output buffer,
            no download (experiments/r6/pcie_paths.py).
No PyTorch anywhere in this module.
"""

from __future__ import annotations

import threading
from collections import OrderedDict
from typing import Iterable, Iterator

import numpy as np

from . import _native as nat
from ._device import PINNED, REGISTERED, DeviceArray, Event, Stream
from .utils.hostcopy import par_copy

CHUNK = 16 << 20
_TLS = threading.local()


class HostPipe:
    """One thread's streams, staging buffers and cached device buffers on one device."""

    def __init__(self, device: int):
        self.device = device
        self.stream = Stream()
        self._stage = []  # [pinned uint8 ndarray, Event or None]
        self._dev = OrderedDict()  # (slot, nbytes) -> DeviceArray
        self._ring = {}  # nbytes -> idle DeviceArrays of remap_frames
        self._k = 0

    # -- device buffers kept between calls (hipMalloc of a 100 MB frame costs about a millisecond) ------------------------
    def device_buffer(self, slot: str, nbytes: int) -> DeviceArray:
        key = (slot, nbytes)
        buf = self._dev.get(key)
        if buf is None:
            while len(self._dev) >= 8:
                self._dev.popitem(last=False)
            buf = self._dev[key] = DeviceArray((nbytes,), np.uint8)
        self._dev.move_to_end(key)
        return buf

    # -- the streaming pipeline's rotating input buffers: checked out for one remap_frames call, kept for the next (three hipMallocs and
    #    hipFrees of 100 MB were 1.5 ms of every call - a tenth of a millisecond per frame of a 16-frame batch)
    def take_ring(self, nbytes: int, count: int) -> list:
        idle = self._ring.setdefault(nbytes, [])
        out = [idle.pop() for _ in range(min(count, len(idle)))]
        while len(out) < count:
            out.append(DeviceArray((nbytes,), np.uint8))
        return out

    def give_ring(self, nbytes: int, bufs: list) -> None:
        idle = self._ring.setdefault(nbytes, [])
        idle.extend(bufs[: max(0, 4 - len(idle))])  # (at most four per size stay)
        for key in [k for k in self._ring if k != nbytes][1:]:  # (and two sizes)
            del self._ring[key]

    def _staging(self):
        """The next page-locked staging chunk, free of its previous DMA."""
        if len(self._stage) < 3:
            ent = [PINNED.ndarray((CHUNK,), np.uint8), None]
            self._stage.append(ent)
        else:
            ent = self._stage[self._k % 3]
        self._k += 1
        if ent[1] is not None:
            ent[1].sync()
        return ent

    def upload(self, a: np.ndarray, dst: DeviceArray, stream: Stream | None = None) -> bool:
        """Queues the upload of `a`'s bytes into `dst` on `stream`.  Returns False when `a` has been read completely on return
        (staged copy), True when the DMA reads the caller's (page-locked) memory until the stream has passed this point."""
        st = stream or self.stream
        a = np.ascontiguousarray(a)
        flat = a.reshape(-1).view(np.uint8)
        n = flat.nbytes
        if n != dst.nbytes:
            raise ValueError(f"upload of {n} bytes into a device buffer of {dst.nbytes}")
        lib = nat.load()
        if REGISTERED.is_registered(a):
            nat.check(lib.pb_memcpy_h2d(dst.data_ptr(), flat.ctypes.data, n, st.handle))
            return True
        for off in range(0, n, CHUNK):
            m = min(CHUNK, n - off)
            ent = self._staging()
            par_copy(ent[0][:m], flat[off : off + m])
            nat.check(lib.pb_memcpy_h2d(dst.data_ptr() + off, ent[0].ctypes.data, m, st.handle))
            if ent[1] is None:
                ent[1] = Event()
            ent[1].record(st)
        return False

    def download(self, src: DeviceArray, shape, dtype, stream: Stream | None = None) -> np.ndarray:
        """Queues the download of `src` into a fresh page-locked ndarray on `stream`; valid once the stream has been synchronised."""
        st = stream or self.stream
        out = PINNED.ndarray(shape, dtype)
        nat.check(nat.load().pb_memcpy_d2h(out.ctypes.data, src.data_ptr(), out.nbytes, st.handle))
        return out


def pipe_for(device: int | None = None) -> HostPipe:
    dev = nat.current_device() if device is None else int(device)
    pipes = getattr(_TLS, "pipes", None)
    if pipes is None:
        pipes = _TLS.pipes = {}
    p = pipes.get(dev)
    if p is None:
        with nat.on_device(dev):
            p = pipes[dev] = HostPipe(dev)
    return p


def remap_ndarray(plan: nat.Plan, image: np.ndarray, interpolation: str = "nearest", device: int | None = None) -> np.ndarray:
    """One frame: uint8 (h, w, 3) ndarray -> fresh uint8 (H, W, 3) ndarray (upload, ONE kernel launch, download)."""
    nat.require_gpu()
    pipe = pipe_for(device)
    with nat.on_device(pipe.device):
        d_in = pipe.device_buffer("in", image.nbytes)
        d_out = pipe.device_buffer("out", 3 * plan.dst.height * plan.dst.width)
        pipe.upload(image, d_in)
        plan.launch(d_in.data_ptr(), d_out.data_ptr(), 1, pipe.stream.handle, interpolation)
        out = pipe.download(d_out, (plan.dst.height, plan.dst.width, 3), np.uint8)
        pipe.stream.sync()
    return out


def remap_frames(plan: nat.Plan, frames: Iterable[np.ndarray], depth: int = 3, interpolation: str = "nearest") -> Iterator[np.ndarray]:
    """Streams host-resident frames through one plan: while frame k + 1 uploads on the H2D stream, the remap kernel of frame k stores its
    output over PCIe straight into frame k's result ndarray (page-locked, device-visible), through `depth` rotating device input
    buffers.  Yields uint8 (H, W, 3) ndarrays in order (page-locked, recycled when dropped)."""
    nat.require_gpu()
    depth = max(2, int(depth))
    dev = nat.current_device()
    pipe = pipe_for(dev)
    sh = (plan.src.height, plan.src.width, 3)
    dh = (plan.dst.height, plan.dst.width, 3)
    n_in = int(np.prod(sh))
    d_in = pipe.take_ring(n_in, depth)
    s_up, s_run = Stream(), Stream()
    uploaded = [Event() for _ in range(depth)]
    computed = [Event() for _ in range(depth)]
    results = [None] * depth
    sources = [None] * depth  # a sequence's frames whose DMA may still be reading them
    pending: list = []  # slots whose kernel has been queued, oldest first

    def drain_one():
        slot = pending.pop(0)
        computed[slot].sync()  # (the kernel has completed: its stores into the host array are visible, the upload before it is through)
        out, results[slot], sources[slot] = results[slot], None, None
        return out

    ahead = frames if isinstance(frames, (list, tuple)) else None
    k = 0
    try:
        for frame in frames:
            a = np.asarray(frame)
            if a.dtype != np.uint8 or tuple(a.shape) != sh:
                raise ValueError(f"frames must be uint8 {sh}, got {a.dtype} {tuple(a.shape)}")
            slot = k % depth  # (free: the previous turn delivered its result, below)
            direct = pipe.upload(a, d_in[slot], s_up)
            uploaded[slot].record(s_up)
            s_run.wait(uploaded[slot])
            out = results[slot] = PINNED.ndarray(dh, np.uint8)
            plan.launch(d_in[slot].data_ptr(), out.ctypes.data, 1, s_run.handle, interpolation)
            computed[slot].record(s_run)
            pending.append(slot)
            if direct and ahead is not None:
                # frames that already exist (a list, a tuple): nothing waits for this DMA - the next frame's is queued behind it at once;
                # the frame is held until its kernel has run.  The NEXT frame is page-locked meanwhile (0.2-0.35 ms off the critical path).
                sources[slot] = a
                if k + 1 < len(ahead) and isinstance(ahead[k + 1], np.ndarray):
                    REGISTERED.is_registered(ahead[k + 1])
            # everything that does not need the next frame happens HERE, while this frame's DMA runs: the oldest result is handed over (the
            # caller's turn with it included) and frees the slot the next frame takes.  What was left between the end of one upload and the
            # start of the next - the link idle - used to hold all of this: 0.15-0.2 ms of a 2.25 ms frame.
            if len(pending) == depth:
                yield drain_one()
            if direct and ahead is None:
                # an iterator may refill the buffer the DMA is reading as soon as it is asked for the next frame: the DMA must be through
                uploaded[slot].sync()
            k += 1
        while pending:
            yield drain_one()
    finally:
        # (a caller that stops early: nothing of the pipeline may still be reading its frames or writing the results it was not given)
        s_up.sync()
        s_run.sync()
        pipe.give_ring(n_in, d_in)
