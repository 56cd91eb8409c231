"""ctypes binding of include/photonbend_hip.h (the C ABI of the HIP library).

There is no fallback: if ``libphotonbend_hip.so`` is missing or a call fails,
this module raises.

PyTorch is OPTIONAL.  The reference needs numpy / Pillow / click (pyproject.toml:9-14) and so does this package: device memory,
streams and page-locked host memory come from the library itself (``_device.py`` over pb_malloc / pb_memcpy_* / pb_stream_* /
pb_host_*).  Where torch IS installed it is imported FIRST, on purpose - it ships its own ``libamdhip64.so`` (SONAME
libamdhip64.so.7); loading it before our library makes both share one HIP runtime, so torch device pointers and streams can be
handed straight to the kernels - and CUDA tensors are accepted wherever a device array is (frames that stay on the device).
"""

from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading

import numpy as np

try:  # must precede the CDLL below - see module docstring
    import torch
except ImportError:  # the NumPy workflow needs no torch
    torch = None

from ._device import DeviceArray
from . import build as _build


def host_math_flavour() -> str:
    """Which code THIS host's NumPy runs for np.arcsin / arccos / arctan / tan - and so which bits the reference would print here:
    "svml" (NumPy's AVX-512 kernels: x86-64 Linux with AVX512_SKX) or "libm" (everything else: the platform's libm).  PB_MATH_FLAVOUR
    overrides.  The library is loaded in the same flavour (include/photonbend_hip.h, "MATH FLAVOURS")."""
    env = os.environ.get("PB_MATH_FLAVOUR", "").lower()
    if env in ("svml", "libm"):
        return env
    try:
        try:
            from numpy._core._multiarray_umath import __cpu_features__ as feats
        except ImportError:
            from numpy.core._multiarray_umath import __cpu_features__ as feats
    except Exception:
        return "svml"
    return "svml" if feats.get("AVX512_SKX") else "libm"


MATH_FLAVOUR = host_math_flavour()
# PB_LIB_PATH: another build of the same sources (A/B experiments, the diagnostic build); else the product library of the host's flavour
LIB_PATH = os.environ.get("PB_LIB_PATH") or (_build.LIBM_LIB_PATH if MATH_FLAVOUR == "libm" else _build.LIB_PATH)

ABI_VERSION = 5
PB_MAX_ROTATIONS = 8
PLAN_DEFER, PLAN_TUNE, PLAN_MATH_SVML, PLAN_MATH_LIBM, PLAN_NO_BILINEAR, PLAN_BILINEAR = 1, 2, 4, 8, 16, 32
MODE_AUTO, MODE_FAITHFUL, MODE_FAST, MODE_FAST_DIRECT = 0, 1, 2, 3
KIND_CAMERA, KIND_DOUBLE, KIND_PANO = 0, 1, 2
LENS_IDS = {
    "equidistant": 0,
    "equisolid": 1,
    "rectilinear": 2,
    "stereographic": 3,
    "orthographic": 4,
    "thoby": 5,
}
LENS_CUSTOM = 6  # a Lens of user callables: the host evaluates it (pb_index_from_map_i32's distance planes)


class PbError(RuntimeError):
    pass


class pb_proj(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("lens", C.c_int32),
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("fov", C.c_double),
        ("magnitude", C.c_double),
        ("f_distance", C.c_double),
    ]

    def key(self):
        return (self.kind, self.lens, self.height, self.width, self.fov, self.magnitude, self.f_distance)


# name -> (restype, argtypes); every symbol of include/photonbend_hip.h
_VP = C.c_void_p
SIGNATURES = {
    "pb_abi_version": (C.c_int, []),
    "pb_math_flavour": (C.c_int, []),
    "pb_last_error": (C.c_char_p, []),
    "pb_init": (C.c_int, [C.c_int]),
    "pb_shutdown": (C.c_int, []),
    "pb_device_name": (C.c_int, [C.c_char_p, C.c_size_t]),
    "pb_plan_create": (C.c_int, [C.POINTER(pb_proj), C.POINTER(C.c_double), C.c_int, C.POINTER(pb_proj), C.POINTER(_VP)]),
    "pb_plan_create_ex": (C.c_int, [C.POINTER(pb_proj), C.POINTER(C.c_double), C.c_int, C.POINTER(pb_proj), C.c_uint, C.c_int, C.POINTER(_VP)]),
    "pb_plan_prepare": (C.c_int, [_VP, C.c_uint, C.c_int]),
    "pb_plan_set_window_budget": (C.c_int, [_VP, C.c_int]),
    "pb_plan_timing": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pb_plan_serialize": (C.c_int, [_VP, _VP, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pb_plan_deserialize": (C.c_int, [_VP, C.c_size_t, C.POINTER(_VP)]),
    "pb_plan_destroy": (None, [_VP]),
    "pb_plan_set_mode": (C.c_int, [_VP, C.c_int]),
    "pb_plan_info": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "pb_plan_dst_shape": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pb_plan_src_shape": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pb_plan_window_budget": (C.c_int, [_VP]),
    "pb_plan_bilinear_float64_tiles": (C.c_int, [_VP]),
    "pb_plan_bilinear_tile_mix": (C.c_int, [_VP, C.POINTER(C.c_longlong)]),
    "pb_plan_bilinear_launch_shape": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pb_plan_matches": (C.c_int, [_VP, C.POINTER(pb_proj), C.POINTER(C.c_double), C.c_int, C.POINTER(pb_proj)]),
    "pb_remap_u8": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_size_t, C.c_size_t, _VP]),
    "pb_remap_u8v": (C.c_int, [_VP, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, _VP]),
    "pb_remap_bilinear_u8": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_size_t, C.c_size_t, _VP]),
    "pb_index_map_i32": (C.c_int, [_VP, _VP, _VP, _VP]),
    "pb_coordmap_f64": (C.c_int, [C.POINTER(pb_proj), _VP, _VP]),
    "pb_rotate_f64": (C.c_int, [C.POINTER(C.c_double), _VP, _VP, C.c_int, C.c_int, _VP]),
    "pb_sample_map_u8": (C.c_int, [C.POINTER(pb_proj), _VP, C.c_int, C.c_int, _VP, _VP, _VP]),
    "pb_index_from_map_i32": (C.c_int, [C.POINTER(pb_proj), _VP, C.c_int, C.c_int, _VP, _VP, _VP, _VP, _VP]),
    "pb_sample_map_bilinear_px": (C.c_int, [C.POINTER(pb_proj), _VP, C.c_int, C.c_int, _VP, _VP, _VP, _VP, C.c_int, C.c_int, _VP]),
    "pb_sample_map_bilinear_u8": (C.c_int, [C.POINTER(pb_proj), _VP, C.c_int, C.c_int, _VP, _VP, _VP]),
    "pb_gather_px": (C.c_int, [_VP, _VP, _VP, C.c_size_t, C.c_int, _VP]),
    "pb_gather_blend_u8": (C.c_int, [_VP, _VP, _VP, _VP, C.c_size_t, C.c_int, C.c_int, _VP]),
    "pb_map_projection_u8": (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, _VP]),
    "pb_synth_frame_u8": (C.c_int, [_VP, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, _VP]),
    "pb_malloc": (C.c_int, [C.POINTER(_VP), C.c_size_t]),
    "pb_free": (C.c_int, [_VP]),
    "pb_memcpy_h2d": (C.c_int, [_VP, _VP, C.c_size_t, _VP]),
    "pb_memcpy_d2h": (C.c_int, [_VP, _VP, C.c_size_t, _VP]),
    "pb_memset": (C.c_int, [_VP, C.c_int, C.c_size_t, _VP]),
    "pb_stream_create": (C.c_int, [C.POINTER(_VP)]),
    "pb_stream_destroy": (C.c_int, [_VP]),
    "pb_stream_sync": (C.c_int, [_VP]),
    "pb_event_create": (C.c_int, [C.POINTER(_VP)]),
    "pb_event_destroy": (C.c_int, [_VP]),
    "pb_event_record": (C.c_int, [_VP, _VP]),
    "pb_event_sync": (C.c_int, [_VP]),
    "pb_event_elapsed_ms": (C.c_int, [_VP, _VP, C.POINTER(C.c_float)]),
    "pb_stream_copy": (C.c_int, [_VP, _VP, C.c_size_t, _VP]),
    "pb_comm_unique_id": (C.c_int, [_VP]),
    "pb_comm_init": (C.c_int, [C.c_int, C.c_int, _VP, C.POINTER(_VP)]),
    "pb_comm_destroy": (C.c_int, [_VP]),
    "pb_comm_rank": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pb_bcast_params": (C.c_int, [_VP, C.POINTER(pb_proj), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(pb_proj), C.c_int, _VP]),
    "pb_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pb_remap_batch_sharded": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), _VP]),
    "pb_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pb_set_device": (C.c_int, [C.c_int]),
    "pb_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "pb_device_sync": (C.c_int, []),
    "pb_stream_wait_event": (C.c_int, [_VP, _VP]),
    "pb_host_alloc": (C.c_int, [C.POINTER(_VP), C.c_size_t]),
    "pb_host_free": (C.c_int, [_VP]),
    "pb_host_register": (C.c_int, [_VP, C.c_size_t]),
    "pb_host_unregister": (C.c_int, [_VP]),
}

_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """dlopen the in-tree HIP library and bind every declared symbol."""
    global _lib, MATH_FLAVOUR
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PbError(
                f"{LIB_PATH} is missing - build it with `python -m photonbend_amd.build{' --libm' if MATH_FLAVOUR == 'libm' else ''}` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if lib.pb_abi_version() != ABI_VERSION:
            raise PbError(f"ABI version mismatch: library says {lib.pb_abi_version()}, binding expects {ABI_VERSION}")
        # the flavour a process runs in is the LIBRARY's: the plan cache's key and the fixtures the tests pick follow it.  A product library
        # must be the host's flavour; a library named by PB_LIB_PATH (A/B builds, the diagnostic build) says itself which one it is - and
        # must agree with an explicit PB_MATH_FLAVOUR.
        theirs = "libm" if lib.pb_math_flavour() == 1 else "svml"
        if theirs != MATH_FLAVOUR:
            if not os.environ.get("PB_LIB_PATH") or os.environ.get("PB_MATH_FLAVOUR", "").lower() in ("svml", "libm"):
                raise PbError(f"{LIB_PATH} is the {theirs} flavour of the library, this process wants {MATH_FLAVOUR}")
            MATH_FLAVOUR = theirs
        _lib = lib
    return _lib


def check(status: int) -> None:
    if status != 0:
        msg = load().pb_last_error()
        raise PbError(f"photonbend_hip error {status}: {msg.decode() if msg else '?'}")


def is_tensor(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def is_device_array(x) -> bool:
    """A CUDA tensor or a DeviceArray: something the kernels can be pointed at."""
    return isinstance(x, DeviceArray) or (is_tensor(x) and x.is_cuda)


def current_stream() -> int:
    """The stream launches go to: torch's current stream where torch is installed (its tensors are ordered on it), the
    default stream otherwise."""
    return int(torch.cuda.current_stream().cuda_stream) if (torch is not None and torch.cuda.is_available()) else 0


def device_count() -> int:
    n = C.c_int(0)
    check(load().pb_device_count(C.byref(n)))
    return int(n.value)


def current_device() -> int:
    if torch is not None and torch.cuda.is_available():
        return int(torch.cuda.current_device())
    d = C.c_int(0)
    check(load().pb_get_device(C.byref(d)))
    return int(d.value)


@contextlib.contextmanager
def on_device(index):
    """Runs the block with device `index` current (None: whatever is current)."""
    if index is None:
        yield
    elif torch is not None and torch.cuda.is_available():
        with torch.cuda.device(int(index)):
            yield
    else:
        prev = current_device()
        if prev != int(index):
            check(load().pb_set_device(int(index)))
        try:
            yield
        finally:
            if prev != int(index):
                check(load().pb_set_device(prev))


def device_index_of(x):
    """Device ordinal a device array lives on (DeviceArrays: the current device, where they were allocated)."""
    if is_tensor(x):
        return x.device.index if x.device.index is not None else current_device()
    return None


_gpu_ok = None


def require_gpu() -> None:
    global _gpu_ok
    if _gpu_ok is None:
        _gpu_ok = (torch.cuda.is_available() if torch is not None else False) or device_count() > 0
    if not _gpu_ok:
        raise PbError("no HIP device is visible; photonbend_amd has no CPU path")


def torch_dtype(dt):
    return torch.from_numpy(np.empty(0, np.dtype(dt))).dtype


def empty(shape, dtype, like=None, device=None):
    """Uninitialised device array: a CUDA tensor where torch is installed (on `like`'s device / `device` / the current one),
    else a DeviceArray.  `like` a DeviceArray forces a DeviceArray."""
    if isinstance(like, DeviceArray) or torch is None:
        return DeviceArray(shape, dtype)
    dev = like.device if is_tensor(like) else (device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    return torch.empty(tuple(shape), dtype=torch_dtype(dtype), device=dev)


def to_host(x) -> np.ndarray:
    """Device array -> fresh ndarray (synchronous)."""
    return x.cpu().numpy() if is_tensor(x) else x.numpy()


def to_device(a: np.ndarray, device=None):
    """ndarray -> device array of the default kind (CUDA tensor with torch, DeviceArray without); synchronous."""
    a = np.ascontiguousarray(a)
    if torch is None:
        return DeviceArray(a.shape, a.dtype).copy_from_host(a)
    if not a.flags.writeable:
        a = a.copy()  # (torch refuses read-only arrays)
    return torch.from_numpy(a).to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))


def _on(x):
    return on_device(device_index_of(x))


class _LaunchGate:
    """Launches of one plan (any number at a time) against the one rebuild of its tables a plan made for the reference's sampler may see -
    the opt-in bilinear mode's first use, on whichever thread of the host's plan cache that happens.  The C ABI's rule for
    pb_plan_prepare on a prepared plan is the caller's: no launch of the plan in flight or started during the call."""

    __slots__ = ("_cv", "_inside", "_closed")

    def __init__(self):
        self._cv = threading.Condition(threading.Lock())
        self._inside, self._closed = 0, False

    def enter(self) -> None:
        with self._cv:
            while self._closed:
                self._cv.wait()
            self._inside += 1

    def leave(self) -> None:
        with self._cv:
            self._inside -= 1
            if self._closed and not self._inside:
                self._cv.notify_all()

    def close(self) -> None:
        with self._cv:
            self._closed = True
            while self._inside:
                self._cv.wait()

    def open(self) -> None:
        with self._cv:
            self._closed = False
            self._cv.notify_all()


class Plan:
    """Owner of one pb_plan (dst projection, rotations, src projection)."""

    def __init__(self, dst: pb_proj, rotations, src: pb_proj, *, defer: bool = False, tune: bool = False, budget: int = 0, bilinear: bool = False):
        """``defer``: no device work now - launches run the faithful kernel until ``prepare()``;
        ``tune``: pick the LDS window budget by timing (opt-in, allocates scratch frames);
        ``budget``: explicit window budget in bytes (0 = library default);
        ``bilinear``: build the opt-in bilinear mode's tables now (0.2-0.3 ms of a c2 plan's 0.85) instead of at the first bilinear use
        (``ensure_bilinear``, called by ``remap`` / ``launch`` / the mode's diagnostics)."""
        lib = load()
        rots = np.ascontiguousarray(np.asarray(list(rotations), dtype=np.float64).reshape(-1, 9))
        if rots.shape[0] > PB_MAX_ROTATIONS:
            raise PbError(f"at most {PB_MAX_ROTATIONS} chained rotations are supported")
        self._h = _VP()
        rp = rots.ctypes.data_as(C.POINTER(C.c_double)) if rots.shape[0] else None
        flags = (PLAN_DEFER if defer else 0) | (PLAN_TUNE if tune else 0) | (0 if bilinear else PLAN_NO_BILINEAR)
        check(lib.pb_plan_create_ex(C.byref(dst), rp, rots.shape[0], C.byref(src), flags, int(budget), C.byref(self._h)))
        self.dst, self.src, self.n_rot = dst, src, rots.shape[0]
        self.double_src = src.kind == KIND_DOUBLE
        self._bilinear, self._deferred, self._bil_lock = bool(bilinear), bool(defer), threading.Lock()
        self._gate = None if (bilinear and not defer) else _LaunchGate()  # (None: the tables are final)

    @classmethod
    def _adopt(cls, handle, dst: pb_proj, src: pb_proj, n_rot: int) -> "Plan":
        self = cls.__new__(cls)
        self._h = handle
        self.dst, self.src, self.n_rot = dst, src, n_rot
        self.double_src = src.kind == KIND_DOUBLE
        self._bilinear, self._deferred, self._bil_lock = True, False, threading.Lock()  # (a restored plan rebuilds the mode's tables with everything else)
        self._gate = None
        return self

    def _rebuild(self, flags: int, budget: int) -> None:
        """pb_plan_prepare with this plan's launches held off: those inside the library leave first, the device drains (the tables the
        call rewrites may be in use by launches already queued, on any stream), new ones wait at the gate."""
        g = self._gate
        if g is not None:
            g.close()
        try:
            lib = load()
            if g is not None and not self._deferred:  # (a prepared plan: its tables may be in use by launches already queued, on any stream)
                check(lib.pb_device_sync())
            check(lib.pb_plan_prepare(self._h, flags, int(budget)))
        finally:
            if g is not None:
                g.open()

    def prepare(self, tune: bool = False, budget: int = 0) -> None:
        """Builds the fast path of a deferred plan on the current device (or re-applies ``budget``)."""
        with self._bil_lock:
            self._rebuild((PLAN_TUNE if tune else 0) | (PLAN_BILINEAR if self._bilinear else 0), budget)
            self._deferred = False
            if self._bilinear:
                self._gate = None

    def ensure_bilinear(self) -> None:
        """The opt-in bilinear mode's tables, built once when the mode is first used (synchronous).  Launches of this plan made through
        this object on other threads wait while the tables are built; launches through the raw ``handle`` follow the C ABI's rule
        (none in flight, as for ``set_window_budget``).  A deferred plan stays deferred - its launches run the float64 kernels of
        either mode - and remembers the wish for ``prepare()``."""
        if self._bilinear:
            return
        with self._bil_lock:
            if not self._bilinear:
                if not self._deferred:
                    self._rebuild(PLAN_BILINEAR, 0)
                    self._gate = None
                self._bilinear = True

    def set_window_budget(self, budget: int) -> None:
        check(load().pb_plan_set_window_budget(self._h, int(budget)))

    def timing(self) -> dict:
        a, b = C.c_double(), C.c_double()
        check(load().pb_plan_timing(self._h, C.byref(a), C.byref(b)))
        return {"prepare_ms": a.value, "tune_ms": b.value}

    def serialize(self) -> bytes:
        """The prepared plan as a blob (``Plan.deserialize`` restores it without re-certifying)."""
        n = C.c_size_t()
        check(load().pb_plan_serialize(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        check(load().pb_plan_serialize(self._h, buf, n.value, C.byref(n)))
        return buf.raw[: n.value]

    @classmethod
    def deserialize(cls, blob: bytes, dst: pb_proj, rotations, src: pb_proj) -> "Plan":
        """Restores a serialized plan and checks that it was made for exactly this request (projections and rotation
        bits): a cache file name is not proof of what the file holds."""
        rots = np.ascontiguousarray(np.asarray(list(rotations), dtype=np.float64).reshape(-1, 9))
        h = _VP()
        raw = bytes(blob)
        check(load().pb_plan_deserialize(raw, len(raw), C.byref(h)))
        plan = cls._adopt(h, dst, src, rots.shape[0])  # (owns the handle: destroyed with `plan` if the check below raises)
        rp = rots.ctypes.data_as(C.POINTER(C.c_double)) if rots.shape[0] else None
        same = load().pb_plan_matches(h, C.byref(dst), rp, rots.shape[0], C.byref(src))
        if same < 0:
            check(same)
        if same != 1:
            raise PbError("the serialized plan belongs to another geometry")
        return plan

    @property
    def handle(self):
        return self._h

    def set_mode(self, mode: int) -> None:
        """MODE_AUTO (certified fast tiles, else faithful), MODE_FAITHFUL, MODE_FAST."""
        check(load().pb_plan_set_mode(self._h, int(mode)))

    def info(self) -> dict:
        fast = C.c_int()
        st = (C.c_longlong * 7)()
        thr = (C.c_longlong * 4)()
        check(load().pb_plan_info(self._h, C.byref(fast), st, thr))
        return {
            "fast_path": bool(fast.value),
            "tiles": int(st[0]),
            "fix_tiles": int(st[1]),
            "fix_pixels": int(st[2]),
            "model_diff_pixels": int(st[3]),
            "lean_tiles": int(st[4]),
            "black_tiles": int(st[5]),
            "direct_tiles": int(st[6]),
            "thresholds": [int(t) for t in thr],
            "window_budget": int(load().pb_plan_window_budget(self._h)),
            "bilinear_float64_tiles": int(load().pb_plan_bilinear_float64_tiles(self._h)),
        }

    def bilinear_launch_shape(self) -> dict:
        """The bilinear launch's workgroup LDS (bytes) and workgroups per frame - diagnostic."""
        self.ensure_bilinear()
        lds, wgs = C.c_int(), C.c_int()
        check(load().pb_plan_bilinear_launch_shape(self._h, C.byref(lds), C.byref(wgs)))
        return {"lds_bytes": lds.value, "workgroups": wgs.value}

    def bilinear_tile_mix(self) -> dict:
        """How the opt-in bilinear mode serves the plan's tiles (diagnostic, synchronous)."""
        self.ensure_bilinear()
        m = (C.c_longlong * 8)()
        check(load().pb_plan_bilinear_tile_mix(self._h, m))
        return dict(zip(("window", "direct", "table", "black", "td3", "entries", "half_windows", "table_plain"), (int(x) for x in m)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.pb_plan_destroy(h)

    # -- launches (device arrays in, device arrays out: CUDA tensors or DeviceArrays; on the current stream unless told otherwise)
    def launch(self, src_ptr: int, dst_ptr: int, n_frames: int = 1, stream: int | None = None, interpolation: str = "nearest",
               src_stride: int = 0, dst_stride: int = 0) -> None:
        """The raw call: n_frames frames at src_ptr / dst_ptr (device addresses, strides in bytes, 0 = packed) on `stream`."""
        if interpolation != "nearest":
            self.ensure_bilinear()
        fn = load().pb_remap_u8 if interpolation == "nearest" else load().pb_remap_bilinear_u8
        self._gated(fn, self._h, src_ptr, dst_ptr, int(n_frames), int(src_stride), int(dst_stride), current_stream() if stream is None else stream)

    def _gated(self, fn, *args) -> None:
        """One library call that launches this plan's kernels: through the gate while the plan's tables may still be rebuilt."""
        g = self._gate
        if g is None:
            check(fn(*args))
            return
        g.enter()
        try:
            check(fn(*args))
        finally:
            g.leave()

    def remap(self, src, out=None, interpolation: str = "nearest"):
        """src: uint8 device array (h, w, 3) or (N, h, w, 3) -> (H, W, 3) / (N, H, W, 3), of src's kind.
        interpolation: "nearest" (the reference's truncating sample) or the opt-in "bilinear"."""
        if interpolation not in ("nearest", "bilinear"):
            raise ValueError("interpolation must be 'nearest' or 'bilinear'")
        require_gpu()
        if not is_device_array(src):
            raise PbError(f"source frames must be uint8 device arrays (CUDA tensors or DeviceArrays), got {type(src).__name__}")
        tens = is_tensor(src)
        batched = len(src.shape) == 4
        shp = tuple(src.shape)
        n = shp[0] if batched else 1
        u8 = (src.dtype == torch.uint8) if tens else (src.dtype == np.uint8)
        if not u8 or shp[-3:] != (self.src.height, self.src.width, 3) or len(shp) not in (3, 4):
            raise PbError(f"source frames must be uint8 cuda (N, {self.src.height}, {self.src.width}, 3), got {shp} {src.dtype}")
        s = src.contiguous() if tens else src
        oshape = (n, self.dst.height, self.dst.width, 3)
        if out is None:
            o = empty(oshape, np.uint8, like=s)
        else:
            o = out
            if tuple(o.shape) not in (oshape, oshape[1:] if not batched else oshape):
                raise PbError("out must be a contiguous uint8 cuda tensor of the destination shape")
            if is_tensor(o) != tens or (tens and (o.dtype != torch.uint8 or not o.is_contiguous())) or (not tens and o.dtype != np.uint8):
                raise PbError("out must be a contiguous uint8 cuda tensor of the destination shape")
            if tens and (not o.is_cuda or o.device != s.device):
                raise PbError(f"out must live on the source's device ({s.device}), got {o.device}")
        with _on(s):
            self.launch(s.data_ptr(), o.data_ptr(), n, None, interpolation)
        if out is not None:
            return out
        return o if batched else o[0]

    def remap_each(self, srcs, outs=None, stream: int | None = None):
        """A batch of SEPARATELY ALLOCATED frames (a ring of buffers) in one launch (``pb_remap_u8v``): srcs / outs are
        sequences of uint8 device arrays (h, w, 3) / (H, W, 3); returns the list of outputs (allocated when ``outs`` is None).
        Same bytes as one ``remap`` per frame, at the batch rate."""
        require_gpu()
        srcs = list(srcs)
        n = len(srcs)
        if n == 0:
            return []
        for s in srcs:
            u8 = is_device_array(s) and ((s.dtype == torch.uint8) if is_tensor(s) else (s.dtype == np.uint8))
            if not u8 or tuple(s.shape) != (self.src.height, self.src.width, 3) or (is_tensor(s) and not s.is_contiguous()):
                raise PbError(f"source frames must be contiguous uint8 device arrays ({self.src.height}, {self.src.width}, 3)")
        if outs is None:
            outs = [empty((self.dst.height, self.dst.width, 3), np.uint8, like=srcs[0]) for _ in range(n)]
        else:
            outs = list(outs)
            if len(outs) != n:
                raise PbError("outs must hold one array per source frame")
            for o in outs:
                u8 = is_device_array(o) and ((o.dtype == torch.uint8) if is_tensor(o) else (o.dtype == np.uint8))
                if not u8 or tuple(o.shape) != (self.dst.height, self.dst.width, 3) or (is_tensor(o) and not o.is_contiguous()):
                    raise PbError(f"outputs must be contiguous uint8 device arrays ({self.dst.height}, {self.dst.width}, 3)")
        sp = (C.c_void_p * n)(*[int(s.data_ptr()) for s in srcs])
        dp = (C.c_void_p * n)(*[int(o.data_ptr()) for o in outs])
        with _on(srcs[0]):
            self._gated(load().pb_remap_u8v, self._h, sp, dp, n, current_stream() if stream is None else stream)
        return outs

    def index_map(self, weights: bool = False, device=None):
        """int32 (H, W) index map, or for a double source (2, H, W) [+ float64 (2, H, W) weights]."""
        require_gpu()
        H, W = self.dst.height, self.dst.width
        shape = (2, H, W) if self.double_src else (H, W)
        idx = empty(shape, np.int32, device=device)
        w = empty((2, H, W), np.float64, like=idx) if (weights and self.double_src) else None
        with _on(idx):
            self._gated(load().pb_index_map_i32, self._h, idx.data_ptr(), w.data_ptr() if w is not None else None, current_stream())
        return (idx, w) if weights else idx


def make_proj(kind: int, height: int, width: int, lens: int = 0, fov: float = 0.0, magnitude: float = 0.0, f_distance: float = 0.0) -> pb_proj:
    return pb_proj(int(kind), int(lens), int(height), int(width), float(fov), float(magnitude), float(f_distance))


def coordmap(dst: pb_proj, device=None):
    require_gpu()
    out = empty((dst.height, dst.width, 3), np.float64, device=device)
    with _on(out):
        check(load().pb_coordmap_f64(C.byref(dst), out.data_ptr(), current_stream()))
    return out


def rotate(matrix: np.ndarray, cmap):
    """cmap (H, W, 3) float64 device array, contiguous; invalid lat/lon are zeroed in it."""
    require_gpu()
    m = np.ascontiguousarray(matrix, dtype=np.float64).reshape(9)
    out = empty(tuple(cmap.shape), np.float64, like=cmap)
    with _on(cmap):
        check(load().pb_rotate_f64(m.ctypes.data_as(C.POINTER(C.c_double)), cmap.data_ptr(), out.data_ptr(), cmap.shape[0], cmap.shape[1], current_stream()))
    return out


def sample_map(src: pb_proj, cmap, image):
    require_gpu()
    out = empty((cmap.shape[0], cmap.shape[1], 3), np.uint8, like=cmap)
    with _on(cmap):
        check(load().pb_sample_map_u8(C.byref(src), cmap.data_ptr(), cmap.shape[0], cmap.shape[1], image.data_ptr(), out.data_ptr(), current_stream()))
    return out


def index_from_map(src: pb_proj, cmap, dist_l=None, dist_r=None):
    """cmap (H, W, 3) float64 device array -> (int32 indices (H, W) or (2, H, W) for a double source, float64 weights (2, H, W) or None).
    Zeroes invalid lat/lon in cmap for a panorama source, like the reference."""
    require_gpu()
    H, W = cmap.shape[0], cmap.shape[1]
    double = src.kind == KIND_DOUBLE
    idx = empty((2, H, W) if double else (H, W), np.int32, like=cmap)
    w = empty((2, H, W), np.float64, like=cmap) if double else None
    for d in (dist_l, dist_r):
        if d is None:
            continue
        if is_tensor(d):
            ok = d.is_cuda and d.dtype == torch.float64 and d.is_contiguous() and d.numel() == H * W and (not is_tensor(cmap) or d.device == cmap.device)
        else:
            ok = isinstance(d, DeviceArray) and d.dtype == np.float64 and d.size == H * W
        if not ok:
            raise PbError("distance planes must be contiguous float64 CUDA tensors of the map's size on the map's device")
    with _on(cmap):
        check(load().pb_index_from_map_i32(C.byref(src), cmap.data_ptr(), H, W, dist_l.data_ptr() if dist_l is not None else None,
                                           dist_r.data_ptr() if dist_r is not None else None, idx.data_ptr(), w.data_ptr() if w is not None else None, current_stream()))
    return idx, w


def sample_map_bilinear(src: pb_proj, cmap, img, channels: int, dt: np.dtype, dist_l=None, dist_r=None):
    """The opt-in bilinear mode from a materialised float64 map (H, W, 3) on the device: img = the image's samples as a device array
    (h, w, channels) of dtype dt (uint8 / uint16) -> (H, W, channels), uint8 for a double-fisheye source (pb_sample_map_bilinear_px)."""
    require_gpu()
    H, W = int(cmap.shape[0]), int(cmap.shape[1])
    dt = np.dtype(dt)
    out_dt = np.dtype(np.uint8) if src.kind == KIND_DOUBLE else dt
    out = empty((H, W, channels), out_dt, like=cmap)
    with _on(cmap):
        check(load().pb_sample_map_bilinear_px(C.byref(src), cmap.data_ptr(), H, W, dist_l.data_ptr() if dist_l is not None else None,
                                               dist_r.data_ptr() if dist_r is not None else None, img.data_ptr(), out.data_ptr(), int(channels),
                                               dt.itemsize, current_stream()))
    return out


def gather_px(idx, img_bytes):
    """idx int32 (H, W), img_bytes uint8 (h, w, bpp) -> uint8 (H, W, bpp): src[idx] or zeros where idx < 0."""
    require_gpu()
    bpp = img_bytes.shape[2]
    n = int(idx.shape[0]) * int(idx.shape[1])
    out = empty((idx.shape[0], idx.shape[1], bpp), np.uint8, like=idx)
    with _on(idx):
        check(load().pb_gather_px(idx.data_ptr(), img_bytes.data_ptr(), out.data_ptr(), n, bpp, current_stream()))
    return out


def gather_blend(idx2, w2, img_bytes, channels: int, sample_bytes: int):
    """The double-fisheye blend of arbitrary-width images: uint8 (H * W * channels)."""
    require_gpu()
    n = int(idx2.shape[1]) * int(idx2.shape[2])
    out = empty((n * channels,), np.uint8, like=idx2)
    with _on(idx2):
        check(load().pb_gather_blend_u8(idx2.data_ptr(), w2.data_ptr(), img_bytes.data_ptr(), out.data_ptr(), n, channels, sample_bytes, current_stream()))
    return out


def map_projection(cmap):
    """cmap (H, W, 3) float64 device array, contiguous -> uint8 (H, W, 3); invalid lat/lon are zeroed in cmap."""
    require_gpu()
    out = empty((cmap.shape[0], cmap.shape[1], 3), np.uint8, like=cmap)
    ws = empty((3,), np.int64, like=cmap)
    with _on(cmap):
        check(load().pb_map_projection_u8(cmap.data_ptr(), cmap.shape[0], cmap.shape[1], out.data_ptr(), ws.data_ptr(), current_stream()))
    if int(to_host(ws)[2]) == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")  # np.min of no valid pixel
    return out


def synth_frame(height: int, width: int, frame: int = 0, seed: int = 0, circle_mask: int = 0, device=None, out=None):
    require_gpu()
    if out is None:
        out = empty((height, width, 3), np.uint8, device=device)
    with _on(out):
        check(load().pb_synth_frame_u8(out.data_ptr(), height, width, frame & 0xFFFFFFFF, seed & 0xFFFFFFFF, circle_mask, current_stream()))
    return out


def device_name() -> str:
    buf = C.create_string_buffer(256)
    check(load().pb_device_name(buf, 256))
    return buf.value.decode()
