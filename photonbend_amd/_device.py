"""Device memory, page-locked host memory and streams over the C ABI alone (pb_malloc / pb_memcpy_* / pb_stream_* /
pb_host_*): what the NumPy-in / NumPy-out path of the package runs on.  No PyTorch here - ``import photonbend_amd`` and
the ndarray workflow of photonbend (core/__init__.py:66-92: ndarray in, fresh ndarray out) need only NumPy and the HIP
library; torch tensors remain an optional way to keep frames on the device.
"""

from __future__ import annotations

import ctypes as C
import threading
import weakref
from collections import OrderedDict

import numpy as np


def _lib():
    from . import _native as nat

    return nat.load()


def _check(rc: int) -> None:
    if rc:
        from . import _native as nat

        nat.check(rc)


class DeviceArray:
    """A contiguous array in device memory (pb_malloc), with just enough of an array's face for the package's own plumbing:
    ``shape``, ``dtype`` (NumPy's), ``data_ptr()``, views, host copies, ``__cuda_array_interface__`` (zero-copy into torch /
    cupy where those are installed).  Frees its memory when the last view goes."""

    __slots__ = ("_ptr", "_owner", "shape", "dtype", "__weakref__")

    def __init__(self, shape, dtype=np.uint8, _ptr=None, _owner=None):
        self.shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        if _ptr is None:
            p = C.c_void_p()
            _check(_lib().pb_malloc(C.byref(p), max(1, self.nbytes)))
            self._ptr = int(p.value)
            self._owner = _Allocation(self._ptr)
        else:
            self._ptr = int(_ptr)
            self._owner = _owner

    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1

    @property
    def nbytes(self) -> int:
        return self.size * self.dtype.itemsize

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def data_ptr(self) -> int:
        return self._ptr

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self._ptr, False), "version": 2, "strides": None}

    def view(self, dtype=None, shape=None) -> "DeviceArray":
        """The same memory under another dtype / shape (the byte count must match)."""
        dt = np.dtype(dtype) if dtype is not None else self.dtype
        shp = tuple(shape) if shape is not None else (self.nbytes // dt.itemsize,)
        if int(np.prod(shp, dtype=np.int64)) * dt.itemsize != self.nbytes:
            raise ValueError("view: the byte counts differ")
        return DeviceArray(shp, dt, _ptr=self._ptr, _owner=self._owner)

    def reshape(self, *shape) -> "DeviceArray":
        shp = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        return self.view(self.dtype, shp)

    def __getitem__(self, i) -> "DeviceArray":
        """Leading-axis integer index or slice (frame k of a batch): a view."""
        if not self.shape:
            raise IndexError("0-d array")
        n = self.shape[0]
        step = (self.nbytes // n) if n else 0
        if isinstance(i, slice):
            a, b, s = i.indices(n)
            if s != 1:
                raise IndexError("only contiguous slices")
            return DeviceArray((max(0, b - a),) + self.shape[1:], self.dtype, _ptr=self._ptr + a * step, _owner=self._owner)
        k = int(i)
        if k < 0:
            k += n
        if not 0 <= k < n:
            raise IndexError(i)
        return DeviceArray(self.shape[1:], self.dtype, _ptr=self._ptr + k * step, _owner=self._owner)

    def copy_from_host(self, a: np.ndarray, stream: int = 0, sync: bool = True) -> "DeviceArray":
        a = np.ascontiguousarray(a)
        if a.nbytes != self.nbytes:
            raise ValueError(f"host array of {a.nbytes} bytes into a device array of {self.nbytes}")
        _check(_lib().pb_memcpy_h2d(self._ptr, a.ctypes.data, a.nbytes, stream))
        if sync:
            _check(_lib().pb_stream_sync(stream))
        return self

    def numpy(self, stream: int = 0) -> np.ndarray:
        """A fresh ndarray with the array's bytes (synchronous)."""
        out = np.empty(self.shape, self.dtype)
        _check(_lib().pb_memcpy_d2h(out.ctypes.data, self._ptr, self.nbytes, stream))
        _check(_lib().pb_stream_sync(stream))
        return out

    def cpu(self) -> "DeviceArray":  # (so that `x.cpu().numpy()` reads the same for a torch tensor and for this)
        return self

    def fill(self, value: int, stream: int = 0) -> None:
        _check(_lib().pb_memset(self._ptr, int(value) & 0xFF, self.nbytes, stream))

    def __repr__(self):
        return f"<DeviceArray {self.shape} {self.dtype} at 0x{self._ptr:x}>"


class _Allocation:
    """Owner of one pb_malloc block; views share it."""

    __slots__ = ("ptr", "__weakref__")

    def __init__(self, ptr: int):
        self.ptr = ptr

    def __del__(self):
        p, self.ptr = self.ptr, 0
        if p:
            try:
                _lib().pb_free(p)
            except Exception:  # interpreter shutdown
                pass


def from_host(a: np.ndarray, stream: int = 0) -> DeviceArray:
    a = np.ascontiguousarray(a)
    return DeviceArray(a.shape, a.dtype).copy_from_host(a, stream)


# ---- page-locked host memory ------------------------------------------------------------------------------------------
class _PinnedPool:
    """hipHostMalloc'd blocks by size, recycled: a result array handed to the caller is page-locked memory the download DMA wrote
    directly (no staging copy), and returns to the pool when the caller drops it."""

    def __init__(self, keep_bytes: int = 1 << 30):
        self._free = {}  # capacity -> [ptr, ...]
        self._kept = 0
        self._keep_bytes = keep_bytes
        self._lock = threading.Lock()

    @staticmethod
    def _capacity(nbytes: int) -> int:
        return max(4096, (nbytes + 0xFFFF) & ~0xFFFF)  # 64 KiB granules: frames of one size share blocks exactly

    def get(self, nbytes: int):
        cap = self._capacity(nbytes)
        with self._lock:
            lst = self._free.get(cap)
            if lst:
                self._kept -= cap
                return lst.pop(), cap
        p = C.c_void_p()
        _check(_lib().pb_host_alloc(C.byref(p), cap))
        return int(p.value), cap

    def put(self, ptr: int, cap: int) -> None:
        with self._lock:
            if self._kept + cap <= self._keep_bytes:
                self._free.setdefault(cap, []).append(ptr)
                self._kept += cap
                return
        try:
            _lib().pb_host_free(ptr)
        except Exception:
            pass

    def ndarray(self, shape, dtype) -> np.ndarray:
        """A writable ndarray in page-locked memory; the block goes back to the pool when the array (and every view of it) is gone."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        ptr, cap = self.get(n)
        buf = (C.c_ubyte * cap).from_address(ptr)
        weakref.finalize(buf, self.put, ptr, cap)
        return np.frombuffer(buf, dtype=dt, count=n // dt.itemsize).reshape(shape)


PINNED = _PinnedPool()


class _Registrations:
    """Page-locking of the CALLER's arrays (pb_host_register): a frame-sized array (>= FIRST_SIGHT_BYTES) at its FIRST sighting, a smaller
    one when its memory is seen a second time (a capture buffer the caller refills) - from then on its upload is ONE DMA straight out of
    the caller's memory.  (Round 6, measured on the MI355X box: page-locking a fresh 100.7 MB ndarray in place costs 0.21-0.35 ms and
    releasing it 0.02 ms, against 0.7-1.5 ms more for the chunked copy through staging buffers - experiments/r6/pcie_paths.py.)  The
    registration is tied to the object that owns the memory (unregistered before that object frees it) and re-checked against
    the owner's address on every use; at most `max_count` buffers / `max_bytes` stay registered (least recently used first out).
    Memory we cannot tie to an owning ndarray (a view of something else) is never registered: it takes the staged copy."""

    FIRST_SIGHT_BYTES = 32 << 20

    def __init__(self, max_count: int = 48, max_bytes: int = 4 << 30):  # (an eviction's release contends with the pipeline's own HIP calls - 0.4-0.5 ms
        # stalls of event.record / stream.wait measured while one ran: a batch of a few dozen never-seen frames should not evict; the bytes bound holds)
        self._seen = OrderedDict()  # (addr, nbytes) -> sightings
        self._reg = OrderedDict()   # addr -> (nbytes, weakref to owner)
        self._late = {}             # addr -> owner: evicted, its release is queued on the worker thread
        self._lock = threading.Lock()
        self._max_count, self._max_bytes = max_count, max_bytes

    @staticmethod
    def _owner(a: np.ndarray):
        o = a
        while isinstance(o.base, np.ndarray):
            o = o.base
        return o if (o.base is None and o.flags.owndata) else None

    def _drop(self, addr: int, later: bool = False) -> None:
        """Ends a registration.  later: an LRU eviction - the release (0.5-0.7 ms for a c2 frame when it follows DMAs, measured) runs on a
        worker thread, which holds the owning array until it is through; a finaliser (the owner is going away) releases at once."""
        with self._lock:
            ent = self._reg.pop(addr, None)
        if ent is None:
            return
        own = ent[1]() if later else None
        if own is not None:
            with self._lock:
                self._late[addr] = own  # (kept alive, and found by its finaliser-in-waiting, until the worker is through)
            _WORKER.submit(self._release_later, addr)
            return
        self._release(addr)

    @staticmethod
    def _release(addr: int) -> None:
        try:
            _lib().pb_host_unregister(addr)
        except Exception:
            pass

    def _release_later(self, addr: int) -> None:
        self._release(addr)
        with self._lock:
            own = self._late.pop(addr, None)
        del own  # (outside the lock: dropping the last reference runs the owner's finaliser, which takes it)

    def settle(self) -> None:
        """Waits for the releases queued by evictions (tests; before a buffer that was evicted is registered again)."""
        _WORKER.submit(lambda: None).result()

    def is_registered(self, a: np.ndarray) -> bool:
        """True when `a`'s bytes lie inside a live registration (then a DMA may read them directly)."""
        addr, n = a.ctypes.data, a.nbytes
        own = self._owner(a)
        if own is None:
            return False
        base, size = own.ctypes.data, own.nbytes
        with self._lock:
            ent = self._reg.get(base)
            if ent is not None and ent[0] == size and ent[1]() is own and base <= addr and addr + n <= base + size:
                self._reg.move_to_end(base)
                return True
            key = (base, size)
            self._seen[key] = self._seen.get(key, 0) + 1
            self._seen.move_to_end(key)
            while len(self._seen) > 64:
                self._seen.popitem(last=False)
            again = self._seen[key] >= 2 or size >= self.FIRST_SIGHT_BYTES
        if ent is not None:  # same address, another object or size: the old registration is stale
            self._drop(base)
        if not again or size < (1 << 20):
            return False
        # a frame-sized buffer, or the second sighting of a smaller one: register it (make room first)
        while True:
            with self._lock:
                total = sum(v[0] for v in self._reg.values())
                victim = next(iter(self._reg)) if (len(self._reg) >= self._max_count or total + size > self._max_bytes) and self._reg else None
            if victim is None:
                break
            self._drop(victim, later=True)
        with self._lock:
            pending = base in self._late
        if pending:  # this very buffer was evicted a moment ago: its release must be through before it is registered again
            self.settle()
        try:
            _check(_lib().pb_host_register(base, size))
        except Exception:
            return False
        with self._lock:
            self._reg[base] = (size, weakref.ref(own))
        weakref.finalize(own, self._drop, base)
        return True


_WORKER = __import__("concurrent.futures").futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="pb-host-release")
REGISTERED = _Registrations()


class Stream:
    def __init__(self):
        p = C.c_void_p()
        _check(_lib().pb_stream_create(C.byref(p)))
        self.handle = int(p.value)

    def sync(self) -> None:
        _check(_lib().pb_stream_sync(self.handle))

    def wait(self, event: "Event") -> None:
        _check(_lib().pb_stream_wait_event(self.handle, event.handle))

    def __del__(self):
        h, self.handle = getattr(self, "handle", 0), 0
        if h:
            try:
                _lib().pb_stream_destroy(h)
            except Exception:
                pass


class Event:
    def __init__(self):
        p = C.c_void_p()
        _check(_lib().pb_event_create(C.byref(p)))
        self.handle = int(p.value)

    def record(self, stream: Stream) -> None:
        _check(_lib().pb_event_record(self.handle, stream.handle))

    def sync(self) -> None:
        _check(_lib().pb_event_sync(self.handle))

    def __del__(self):
        h, self.handle = getattr(self, "handle", 0), 0
        if h:
            try:
                _lib().pb_event_destroy(h)
            except Exception:
                pass
