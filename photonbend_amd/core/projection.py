"""Projection images - the drop-in for photonbend.core.projection
(projection.py:40-547): ``CameraImage``, ``DoubleCameraImage``, ``PanoramaImage``
behind the ``ProjectionImage`` protocol, same constructor signatures, same public
attributes, same error behaviour.

What differs is where the work happens.  ``get_coordinate_map()`` returns a lazy
``CoordinateMap`` recipe; ``process_coordinate_map(recipe)`` runs ONE fused HIP
kernel (pb_remap_u8) in which each output pixel's inverse projection, rotations,
forward projection and sample happen in registers.  A real ndarray map (a recipe
that was looked at or edited, or a hand-made map) is sampled by the
materialised-map kernel (pb_sample_map_u8).  There is no NumPy path.

``image`` may be a NumPy uint8 array (H, W, 3) - uploaded per call, result
returned as a fresh ndarray like the reference (``_hostpipe.py``: no PyTorch
involved) - or, where PyTorch is installed, a uint8 CUDA tensor, which stays on
the device and yields a CUDA tensor.
"""

from __future__ import annotations

import contextlib
import hashlib
import os
import threading
from abc import abstractmethod
from collections import OrderedDict
from typing import Protocol, Union

import numpy as np

from .. import _hostpipe
from .. import _native as nat
from ._coordmap import CoordinateMap
from .lens import Lens, lens_id

_PLAN_CACHE: "OrderedDict" = OrderedDict()  # key -> [plan, uses, prepared, pending preparation]; least recently used first
_PLAN_CACHE_MAX = 64
_PLAN_LOCK = threading.RLock()


def _plan_key(dst: nat.pb_proj, rotations, src: nat.pb_proj, dev_index: int):
    return (dev_index, dst.key(), tuple(np.asarray(r, dtype=np.float64).tobytes() for r in rotations), src.key())


def _disk_cache_path(key) -> Union[str, None]:
    """Opt-in persistence of prepared plans (PB_PLAN_CACHE_DIR): a process that has seen a geometry before - the
    CLI run again on another image - uploads the certified tables instead of rebuilding them."""
    root = os.environ.get("PB_PLAN_CACHE_DIR")
    if not root:
        return None
    lib_stamp = str(os.path.getmtime(nat.LIB_PATH)) if os.path.exists(nat.LIB_PATH) else "?"
    digest = hashlib.sha256(repr((key[1:], lib_stamp, nat.ABI_VERSION, nat.MATH_FLAVOUR)).encode()).hexdigest()[:32]
    return os.path.join(root, f"plan_{digest}.pbplan")


def _prepare(plan: nat.Plan, key, rotations) -> nat.Plan:
    """A PREPARED plan for the geometry of `plan` (a deferred plan): from the disk cache when there is one, else built and
    certified now.  Always a NEW object - the deferred one may be in use by other threads (its launches run the faithful
    kernel) and is never mutated; the caller swaps the cache entry."""
    path = _disk_cache_path(key)
    if path and os.path.exists(path):
        try:
            with open(path, "rb") as f:
                return nat.Plan.deserialize(f.read(), plan.dst, rotations, plan.src)
        except (OSError, nat.PbError):
            pass  # stale or foreign blob: rebuild
    fresh = nat.Plan(plan.dst, rotations, plan.src)
    if path:
        try:
            os.makedirs(os.path.dirname(path), mode=0o700, exist_ok=True)  # plans drive unguarded device loads: keep the directory private
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "wb") as f:
                f.write(fresh.serialize())
            os.replace(tmp, path)
        except (OSError, nat.PbError):
            pass
    return fresh


def _plan_for(dst: nat.pb_proj, rotations, src: nat.pb_proj, device=None, eager: bool = True) -> nat.Plan:
    """The cached plan of a geometry on `device` (default: the current device).

    eager=True returns a PREPARED plan (per-tile models built and certified: tens of frames' worth of GPU time,
    once).  eager=False is what the one-image facade calls use: the FIRST remap of a geometry runs the faithful
    kernel from a deferred plan (no preparation at all - the reference CLI's case costs one faithful launch), the
    second use of the same geometry prepares the fast path.  PB_PLAN_EAGER=1 makes every use eager."""
    rotations = list(rotations)
    if len(rotations) > nat.PB_MAX_ROTATIONS:
        raise nat.PbError(f"at most {nat.PB_MAX_ROTATIONS} rotations fit one fused plan")
    have_gpu = _have_gpu()
    if device is None:
        dev_index = nat.current_device() if have_gpu else 0
    elif isinstance(device, int):
        dev_index = device
    elif isinstance(device, str):  # "cuda" / "cuda:N"
        tail = device.partition(":")[2]
        dev_index = int(tail) if tail else (nat.current_device() if have_gpu else 0)
    else:  # a torch.device
        idx = getattr(device, "index", None)
        dev_index = idx if idx is not None else (nat.current_device() if have_gpu else 0)
    eager = eager or os.environ.get("PB_PLAN_EAGER") == "1"
    key = _plan_key(dst, rotations, src, dev_index)
    ctx = (lambda: nat.on_device(dev_index)) if have_gpu else contextlib.nullcontext
    # _PLAN_LOCK guards the DICTIONARY only (round 5).  Preparing a plan - 0.4-1.1 ms warm, 90-150 ms for a process's first - runs
    # outside it, owned by the one thread that found the entry unprepared (entry[3] = its _Pending); a one-thread-per-GPU host
    # prepares its eight plans side by side, and threads asking for OTHER geometries never wait.
    mine = pending = None
    with _PLAN_LOCK:
        entry = _PLAN_CACHE.get(key)
        if entry is None:
            with ctx():
                plan = nat.Plan(dst, rotations, src, defer=True)  # (a parameter block: no device work)
            entry = _PLAN_CACHE[key] = [plan, 0, False, None]
            while len(_PLAN_CACHE) > _PLAN_CACHE_MAX:
                _PLAN_CACHE.popitem(last=False)  # evict ONE entry, the least recently used
        entry[1] += 1
        if not entry[2] and (eager or entry[1] >= 2) and have_gpu:
            if entry[3] is None:
                mine = entry[3] = _Pending()
            else:
                pending = entry[3]
        _PLAN_CACHE.move_to_end(key)
        plan = entry[0]
    if mine is not None:
        try:
            with ctx():
                fresh = _prepare(plan, key, rotations)
        except BaseException as exc:
            with _PLAN_LOCK:
                entry[3] = None  # the next caller may try again
            mine.error = exc
            mine.done.set()
            raise
        with _PLAN_LOCK:
            entry[0], entry[2], entry[3] = fresh, True, None
        mine.done.set()
        return fresh
    if pending is not None and eager:
        # another thread is preparing this very geometry: an eager caller wants the prepared plan (a facade call does not wait - the
        # deferred plan it holds remaps with the float64 kernel, same bytes)
        pending.done.wait()
        if pending.error is not None:
            raise nat.PbError(f"plan preparation failed in another thread: {pending.error}")
        with _PLAN_LOCK:
            return entry[0]
    return plan


class _Pending:
    """A plan preparation in flight: the owner sets `done` (and `error` when it failed)."""

    def __init__(self):
        self.done = threading.Event()
        self.error = None


def _have_gpu() -> bool:
    try:
        nat.require_gpu()
        return True
    except nat.PbError:
        return False


def _shape_hw(image) -> tuple:
    shp = tuple(image.shape)
    if len(shp) < 2:
        raise ValueError("an image needs at least (height, width)")
    return int(shp[0]), int(shp[1])


def _to_host(t) -> np.ndarray:
    """Device array -> fresh ndarray (the reference returns freshly allocated arrays)."""
    if nat.is_tensor(t):
        pipe = _hostpipe.pipe_for(nat.device_index_of(t))
        out = _hostpipe.PINNED.ndarray(tuple(t.shape), nat.to_host(t[:0]).dtype) if t.numel() >= (1 << 20) else None
        if out is None:
            return t.cpu().numpy()
        t = t.contiguous()
        with nat.on_device(pipe.device):
            nat.torch.cuda.current_stream().synchronize()  # the tensor's producer runs on torch's stream
            nat.check(nat.load().pb_memcpy_d2h(out.ctypes.data, t.data_ptr(), out.nbytes, pipe.stream.handle))
            pipe.stream.sync()
        return out
    return t.numpy()


def _upload(a: np.ndarray, device=None):
    """ndarray -> device array on `device` (default: current); synchronous."""
    with nat.on_device(getattr(device, "index", device) if device is not None else None):
        return nat.to_device(a, device if nat.torch is not None else None)


def _device_image(image, height: int, width: int):
    """uint8 device array (h, w, 3) of the pixels behind ``.image`` (a tensor stays on ITS device)."""
    nat.require_gpu()
    if nat.is_tensor(image):
        t = image
        if t.dtype != nat.torch.uint8:
            raise TypeError("image tensors must be uint8")
        if not t.is_cuda:
            t = t.cuda()
        if tuple(t.shape) != (height, width, 3):
            raise ValueError(f"image must have shape ({height}, {width}, 3), got {tuple(t.shape)}")
        return t.contiguous()
    if isinstance(image, nat.DeviceArray):
        if image.dtype != np.uint8 or tuple(image.shape) != (height, width, 3):
            raise ValueError(f"image must be uint8 ({height}, {width}, 3), got {image.dtype} {tuple(image.shape)}")
        return image
    a = _checked_rgb8(image, height, width)
    return _upload(a)


def _checked_rgb8(image, height: int, width: int) -> np.ndarray:
    a = np.asarray(image)
    if a.dtype != np.uint8:
        raise TypeError(f"images are uint8 (H, W, 3) RGB arrays (core/__init__.py:31-36), got {a.dtype}")
    if tuple(a.shape) != (height, width, 3):
        raise ValueError(f"image must have shape ({height}, {width}, 3), got {tuple(a.shape)}")
    return a


def _check_map_tensor(cmap, device=None) -> None:
    """A tensor coordinate map must be what the kernels index: contiguous float64 (H, W, 3) on the image's device."""
    if nat.is_tensor(cmap):
        if not (cmap.is_cuda and cmap.dtype == nat.torch.float64 and cmap.is_contiguous()):
            raise TypeError("tensor coordinate maps must be contiguous float64 CUDA tensors")
        shp = tuple(cmap.shape)
        if device is not None and cmap.device != device:
            raise ValueError(f"coordinate map on {cmap.device} but the image on {device}")
    else:
        if cmap.dtype != np.float64:
            raise TypeError("device coordinate maps must be float64")
        shp = tuple(cmap.shape)
    if len(shp) != 3 or shp[2] != 3:
        raise ValueError(f"a coordinate map has shape (H, W, 3), got {shp}")


class ProjectionImage(Protocol):
    """The protocol every projection image follows (projection.py:40-66)."""

    image: np.ndarray

    @abstractmethod
    def get_coordinate_map(self):
        ...

    @abstractmethod
    def process_coordinate_map(self, coordinate_map):
        ...


def _image_info(image):
    """(height, width, trailing shape, numpy dtype) of an image array or tensor.  The reference fancy-indexes whatever
    array it is given (projection.py:234-243, :545-546): grey (H, W), RGB, RGBA (H, W, 4), 8- or 16-bit samples."""
    shp = tuple(int(v) for v in image.shape)
    if len(shp) < 2:
        raise ValueError("an image needs at least (height, width)")
    if nat.is_tensor(image):
        dt = nat.torch.empty(0, dtype=image.dtype).numpy().dtype
    elif isinstance(image, nat.DeviceArray):
        dt = image.dtype
    else:
        dt = np.asarray(image).dtype
    return shp[0], shp[1], shp[2:], np.dtype(dt)


def _device_bytes(image, device=None):
    """The image's pixels as a contiguous uint8 device array (h, w, bytes per pixel)."""
    nat.require_gpu()
    h, w, tail, dt = _image_info(image)
    bpp = int(np.prod(tail, dtype=np.int64)) * dt.itemsize
    if nat.is_tensor(image):
        t = image if image.is_cuda else image.cuda()
        return t.contiguous().view(nat.torch.uint8).reshape(h, w, bpp)
    if isinstance(image, nat.DeviceArray):
        return image.view(np.uint8, (h, w, bpp))
    a = np.ascontiguousarray(image)
    return _upload(a.view(np.uint8).reshape(h, w, bpp), device)


def _typed(out, tail, dt: np.dtype, H: int, W: int):
    """uint8 device bytes (H, W, bpp) as the image's sample type and trailing shape."""
    if nat.is_tensor(out):
        return out.view(nat.torch_dtype(dt)).reshape((H, W) + tuple(tail))
    return out.view(dt, (H, W) + tuple(tail))


class _GpuProjection:
    """Shared GPU plumbing of the three projection classes."""

    image: np.ndarray  # (or a uint8 CUDA tensor / DeviceArray: the pixels then stay on the device)

    def _proj(self, role: str = "src") -> nat.pb_proj:  # pragma: no cover - overridden
        raise NotImplementedError

    def get_coordinate_map(self) -> CoordinateMap:
        """This image's coordinate map, as a lazy recipe (see ``CoordinateMap``).  A destination whose lens is made
        of user callables gets a materialised map: ``reverse_function`` runs on the host over the exact radius mesh
        (lens.py:48-64, projection.py:186-189), everything around it on the GPU."""
        proj = self._proj("dst")
        if proj.kind != nat.KIND_PANO and proj.lens == nat.LENS_CUSTOM:
            return CoordinateMap.from_array(proj, self._custom_coordinate_map(proj))
        return CoordinateMap(proj)

    # -- the sampling half, for everything the fused uint8 RGB kernel does not take ------------------------------
    def _source_distances(self, lat: np.ndarray):  # pragma: no cover - overridden by the camera classes
        raise NotImplementedError

    def _distance_planes(self, src: nat.pb_proj, dev_map):
        """forward_lens(latitude) * f_distance per pixel for a source Lens of user callables (None, None for a built-in lens)."""
        dl = dr = None
        if src.kind != nat.KIND_PANO and src.lens == nat.LENS_CUSTOM:
            # forward_function is host Python by definition: latitude plane down, distances up (projection.py:251)
            lat = np.ascontiguousarray(nat.to_host(dev_map)[..., 0])
            planes = self._source_distances(lat)
            dev = dev_map.device if nat.is_tensor(dev_map) else None
            dl = _upload(np.ascontiguousarray(planes[0], dtype=np.float64), dev)
            if src.kind == nat.KIND_DOUBLE:
                dr = _upload(np.ascontiguousarray(planes[1], dtype=np.float64), dev)
        return dl, dr

    def _index_from_map(self, src: nat.pb_proj, dev_map):
        """int32 source indices (and float64 weights for a double source) of a materialised map on the device."""
        dl, dr = self._distance_planes(src, dev_map)
        return nat.index_from_map(src, dev_map, dl, dr)

    def _gather(self, src: nat.pb_proj, idx, weights, img_bytes, tail, dt: np.dtype):
        """index map -> output pixels, any channel count / sample width; returns a CUDA tensor of the output dtype."""
        H, W = (idx.shape[-2], idx.shape[-1])
        if src.kind == nat.KIND_DOUBLE:
            if len(tail) != 1:
                # the reference multiplies (H, W) samples by an (H, W, 1) factor map: NumPy cannot broadcast that
                raise ValueError(f"operands could not be broadcast together with shapes ({H},{W}) ({H},{W},1)")
            if dt not in (np.dtype(np.uint8), np.dtype(np.uint16)):
                raise NotImplementedError(f"the double-fisheye blend takes uint8 or uint16 images, got {dt}")
            out = nat.gather_blend(idx, weights, img_bytes, tail[0], dt.itemsize)  # uint8, like .astype(np.uint8)
            return out.reshape(H, W, tail[0])
        return _typed(nat.gather_px(idx, img_bytes), tail, dt, H, W)

    def process_coordinate_map(self, coordinate_map, interpolation: str = "nearest"):
        """Maps this image's pixels through ``coordinate_map`` and returns the new image
        (projection.py:197-245, :408-462, :515-547).

        uint8 (H, W, 3) images with built-in lenses take the fused kernel (one launch, no map in memory); any other
        image the reference accepts - grey (H, W), RGBA, 16-bit samples - and sources whose lens is made of user
        callables go through the integer index map and a gather (same indices, same bytes as the reference).

        ``interpolation="bilinear"`` is an opt-in extension with no reference counterpart (the reference truncates to the nearest
        pixel).  A lazy map + a uint8 RGB image + built-in lenses take the tile kernels (one launch); a materialised or edited map, a
        grey / RGBA / 16-bit image or a Lens of user callables take the mode's definition per pixel from the map
        (pb_sample_map_bilinear_px) - same definition, float64 arithmetic."""
        src = self._proj("src")
        h, w, tail, dt = _image_info(self.image)
        rgb8 = tail == (3,) and dt == np.dtype(np.uint8)
        custom_src = src.kind != nat.KIND_PANO and src.lens == nat.LENS_CUSTOM
        lazy = isinstance(coordinate_map, CoordinateMap) and coordinate_map.is_lazy
        bilinear = interpolation != "nearest"
        if bilinear:
            if interpolation != "bilinear":
                raise ValueError("interpolation must be 'nearest' or 'bilinear'")
            if dt not in (np.dtype(np.uint8), np.dtype(np.uint16)):
                raise NotImplementedError(f"bilinear sampling takes 8- or 16-bit unsigned samples, got {dt}")
            if src.kind == nat.KIND_DOUBLE and len(tail) != 1:
                # (the reference's blend cannot broadcast (H, W) samples against its (H, W, 1) factor maps either)
                H_, W_ = tuple(coordinate_map.shape[:2])
                raise ValueError(f"operands could not be broadcast together with shapes ({H_},{W_}) ({H_},{W_},1)")
        on_device = nat.is_device_array(self.image)  # the pixels live on the device: so does the result
        fused = rgb8 and not custom_src
        rotations = coordinate_map.rotations if lazy else ()
        if bilinear and lazy and len(rotations) > nat.PB_MAX_ROTATIONS:
            # The reference applies any number of -r rotations one after the other (scripts/commands/make_photo.py:128-131).  A chain
            # longer than one fused plan takes leaves the plan for the materialised-map kernels, which only truncate; in THIS mode
            # (our own definition, no reference bits to keep) the chain folds into one matrix product R_k ... R_1 instead.
            folded = np.eye(3)
            for r in rotations:
                folded = np.asarray(r, dtype=np.float64).reshape(3, 3) @ folded
            rotations = [folded]
        too_many = lazy and len(rotations) > nat.PB_MAX_ROTATIONS
        if fused and not on_device and lazy and not too_many:
            # THE path of a user who swapped imports: ndarray in, fresh ndarray out - upload, ONE fused launch, download, on the
            # package's own device buffers, stream and page-locked memory (_hostpipe.py; no PyTorch involved)
            nat.require_gpu()
            a = _checked_rgb8(self.image, h, w)
            plan = _plan_for(coordinate_map.dst_proj, rotations, src, eager=interpolation != "nearest")
            out = _hostpipe.remap_ndarray(plan, a, interpolation)
            if src.kind == nat.KIND_PANO:
                coordinate_map.note_invalid_zeroed()  # projection.py:534-536
            return out
        img = _device_image(self.image, h, w) if fused else _device_bytes(self.image)
        dev = img.device if nat.is_tensor(img) else None
        if lazy and not too_many and not custom_src and (fused or not bilinear):
            # bilinear taps come from the tile models: that mode needs the prepared plan from the first use on
            plan = _plan_for(coordinate_map.dst_proj, rotations, src, device=dev, eager=interpolation != "nearest")
            with nat.on_device(nat.device_index_of(img)):
                if fused:
                    out = plan.remap(img, interpolation=interpolation)
                else:
                    idx, wts = plan.index_map(weights=True, device=dev) if src.kind == nat.KIND_DOUBLE else (plan.index_map(device=dev), None)
                    out = self._gather(src, idx, wts, img, tail, dt)
            if src.kind == nat.KIND_PANO:
                coordinate_map.note_invalid_zeroed()  # projection.py:534-536
            return out if on_device else _to_host(out)
        # a materialised map: the caller's tensor / ndarray, or a recipe that has to become one (more rotations than one
        # fused plan takes: the reference accepts any number of -r options; a source lens evaluated on the host)
        host = None
        with nat.on_device(nat.device_index_of(img)):
            if nat.is_device_array(coordinate_map):
                _check_map_tensor(coordinate_map, dev)
                dmap = coordinate_map
            elif lazy:
                dmap = coordinate_map.device_tensor()
                if src.kind == nat.KIND_PANO:
                    coordinate_map.note_invalid_zeroed()
            else:
                host = coordinate_map.materialize() if isinstance(coordinate_map, CoordinateMap) else coordinate_map
                if not (isinstance(host, np.ndarray) and host.dtype == np.float64 and host.ndim == 3 and host.shape[2] == 3):
                    raise TypeError("coordinate_map must be a float64 array of shape (H, W, 3)")
                dmap = _upload(host, dev)
            if bilinear:
                # the mode's definition per pixel from the map (pb_sample_map_bilinear_px): a materialised or edited map, any image
                # layout, a source Lens of user callables - everything the tile kernels do not take
                dl, dr = self._distance_planes(src, dmap)
                channels = int(np.prod(tail, dtype=np.int64))
                out = nat.sample_map_bilinear(src, dmap, img, channels, dt, dl, dr)
                out_dt = np.dtype(np.uint8) if src.kind == nat.KIND_DOUBLE else dt
                H_, W_ = int(dmap.shape[0]), int(dmap.shape[1])
                out = out.reshape((H_, W_) + tuple(tail)) if nat.is_tensor(out) else out.view(out_dt, (H_, W_) + tuple(tail))
            elif fused:
                out = nat.sample_map(src, dmap, img)
            else:
                idx, wts = self._index_from_map(src, dmap)
                out = self._gather(src, idx, wts, img, tail, dt)
            if host is not None and src.kind == nat.KIND_PANO:
                host[...] = nat.to_host(dmap)  # the in-place zeroing of invalid pixels
            return out if on_device else _to_host(out)


def _role_lens_id(lens: Lens, role: str) -> int:
    """pb_lens id of the function the role uses: a destination inverts (reverse_function), a source projects
    (forward_function).  A user callable gets PB_LENS_CUSTOM: the host evaluates it (lens.py:48-64)."""
    lid = lens_id(lens.reverse_function if role == "dst" else lens.forward_function)
    return nat.LENS_CUSTOM if lid is None else lid


class CameraImage(_GpuProjection):
    """A single-fisheye / rectilinear camera image (projection.py:69-274).

    Attributes: image, fov, forward_lens, reverse_lens, magnitude, f_distance."""

    def __init__(self, image_arr, fov: float, lens: Lens, magnitude: Union[None, float] = None):
        self.image = image_arr
        self.fov = fov
        self.forward_lens = lens.forward_function
        self.reverse_lens = lens.reverse_function
        self._lens = lens
        height, _ = _shape_hw(image_arr)
        self.magnitude: float = (height / 2.0) if (magnitude is None) else magnitude
        self.f_distance = self._compute_f_distance()

    def _compute_f_distance(self) -> float:
        """Pixels per focal length: magnitude / forward(fov / 2) (projection.py:123-144);
        raises what the lens raises (rectilinear beyond 178 degrees)."""
        return self.magnitude / self.forward_lens(self.fov / 2)

    def _proj(self, role: str = "src") -> nat.pb_proj:
        h, w = _shape_hw(self.image)
        return nat.make_proj(nat.KIND_CAMERA, h, w, _role_lens_id(self._lens, role), self.fov, self.magnitude, self.f_distance)

    def _custom_coordinate_map(self, proj: nat.pb_proj) -> np.ndarray:
        """get_coordinate_map for a user reverse_function (projection.py:147-194).  The radius mesh
        sqrt(x^2 + y^2) / f_distance is IEEE-exact arithmetic: it comes from the device as the latitude plane of
        the equidistant lens (whose inverse is the identity), the longitudes with it; the callable and the
        validity rule run here."""
        eq = nat.make_proj(nat.KIND_CAMERA, proj.height, proj.width, nat.LENS_IDS["equidistant"], proj.fov, proj.magnitude, proj.f_distance)
        m = nat.to_host(nat.coordmap(eq))
        lat = np.asarray(self.reverse_lens(m[:, :, 0].copy()), dtype=np.float64)
        m[:, :, 0] = lat
        with np.errstate(invalid="ignore"):
            m[:, :, 2] = (lat > self.fov / 2).astype(np.float64)  # projection.py:160 (NaN compares False: valid)
        return m

    def _source_distances(self, lat: np.ndarray):
        return (self.forward_lens(lat) * self.f_distance,)  # projection.py:251


class DoubleCameraImage(_GpuProjection):
    """Two side-by-side fisheyes from a 360-degree camera (projection.py:277-462).

    Attributes: image, sensor_fov, lens, forward_lens, reverse_lens, magnitude,
    f_distance.  Extra keyword arguments (the CLI passes ``magnitude=``) are
    accepted and ignored, like the reference (projection.py:296-316)."""

    def __init__(self, image_arr, sensor_fov: float, lens: Lens, **kwargs):
        self.image = image_arr
        self.sensor_fov = sensor_fov
        self.lens = lens
        self.forward_lens = lens.forward_function
        self.reverse_lens = lens.reverse_function
        height, _ = _shape_hw(image_arr)
        self.magnitude = height / 2.0
        self.f_distance = self._compute_f_distance()

    def _compute_f_distance(self) -> float:
        return self.magnitude / self.forward_lens(self.sensor_fov / 2)

    def _proj(self, role: str = "src") -> nat.pb_proj:
        h, w = _shape_hw(self.image)
        if role == "dst":
            w = 2 * (w // 2)  # the reference's map of an odd-width double frame has 2 * (W // 2) columns (projection.py:389-397)
        return nat.make_proj(nat.KIND_DOUBLE, h, w, _role_lens_id(self.lens, role), self.sensor_fov, self.magnitude, self.f_distance)

    def _custom_coordinate_map(self, proj: nat.pb_proj) -> np.ndarray:
        """get_coordinate_map for a user reverse_function (projection.py:341-406): the two eyes share one radius mesh
        (the right eye's x axis is the left one negated), taken from the device like CameraImage's."""
        half = proj.width // 2
        eq = nat.LENS_IDS["equidistant"]
        dist = nat.to_host(nat.coordmap(nat.make_proj(nat.KIND_CAMERA, proj.height, half, eq, proj.fov, proj.magnitude, proj.f_distance)))[:, :, 0]
        m = nat.to_host(nat.coordmap(nat.make_proj(nat.KIND_DOUBLE, proj.height, proj.width, eq, proj.fov, proj.magnitude, proj.f_distance)))
        lat = np.asarray(self.reverse_lens(np.concatenate([dist, dist], axis=1)), dtype=np.float64)
        lat[:, half:] *= -1
        lat[:, half:] += np.pi
        with np.errstate(invalid="ignore"):
            invalid = lat > self.sensor_fov / 2.0
            invalid[:, half:] = lat[:, half:] < np.pi - (self.sensor_fov / 2.0)
        m[:, :, 0] = lat
        m[:, :, 2] = invalid.astype(np.float64)
        return m

    def _source_distances(self, lat: np.ndarray):
        lat_r = lat.copy()
        lat_r *= -1
        lat_r += np.pi  # projection.py:425-427
        return self.forward_lens(lat.copy()) * self.f_distance, self.forward_lens(lat_r) * self.f_distance


class PanoramaImage(_GpuProjection):
    """An equirectangular panorama (projection.py:465-547)."""

    def __init__(self, image_arr) -> None:
        self.image = image_arr

    def _proj(self, role: str = "src") -> nat.pb_proj:
        h, w = _shape_hw(self.image)
        return nat.make_proj(nat.KIND_PANO, h, w)


def map_projection(coordinate_map):
    """Coordinate map -> RGB colour map for eyeballing a projection (projection.py:550-599): latitude in
    red (stretched over the valid pixels), longitude in green, the invalid flag in blue.  Runs on the GPU
    (pb_map_projection_u8); like the reference it zeroes lat/lon of invalid pixels in the map it is given."""
    if nat.is_device_array(coordinate_map):
        _check_map_tensor(coordinate_map)
        return nat.map_projection(coordinate_map)
    host = coordinate_map.materialize() if isinstance(coordinate_map, CoordinateMap) else coordinate_map
    if not (isinstance(host, np.ndarray) and host.dtype == np.float64 and host.ndim == 3 and host.shape[2] == 3):
        raise TypeError("coordinate_map must be a float64 array of shape (H, W, 3)")
    nat.require_gpu()
    dev = _upload(host)
    out = nat.map_projection(dev)
    host[...] = nat.to_host(dev)
    return _to_host(out)
