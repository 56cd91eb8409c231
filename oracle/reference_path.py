"""TEST INFRASTRUCTURE ONLY - NumPy restatement of photonbend.core's remap path.

This is the parity oracle and the reported CPU baseline ("port").  It is NOT
shipped, NOT imported by ``photonbend_amd`` and NOT a fallback.  It restates,
stage by stage, what the reference computes (all citations are
``/root/reference/photonbend/...`` file:line) using the same NumPy ufuncs the
reference reaches, because which libm/SVML/BLAS routine NumPy dispatches to
decides the last bit of every float64 and therefore the truncated integer
source index.  It is pinned bit-for-bit against fixtures generated from the
real reference (``oracle/make_goldens.py`` -> ``tests/golden/``).

Vocabulary follows the reference: *coordinate map* = float64 (H, W, 3) holding
(latitude, longitude, invalid-flag) per pixel (core/__init__.py:42-49).
"""

from __future__ import annotations

import warnings
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

LENSES = ("equidistant", "equisolid", "rectilinear", "stereographic", "orthographic", "thoby")
KINDS = ("camera", "double", "pano")

INT64_MIN = np.iinfo(np.int64).min


def to_radians(degrees: float) -> float:
    """utils/__init__.py:27-37 - divide first, then multiply by pi."""
    return degrees / 180 * np.pi


# --------------------------------------------------------------------------
# a-1  lens functions (core/lens.py:68-335)
# --------------------------------------------------------------------------
def lens_forward(lens, theta):
    """theta (rad) -> distance from the centre in focal-length units.  `lens`: a built-in's name, or a (forward, reverse) pair of
    callables - a Lens of user functions (lens.py:48-64)."""
    if isinstance(lens, tuple):
        return lens[0](theta)
    if lens == "equidistant":  # lens.py:169-187
        return theta
    if lens == "equisolid":  # lens.py:224-243
        return 2 * np.sin(theta / 2.0)
    if lens == "stereographic":  # lens.py:127-145
        return 2.0 * np.tan(theta / 2.0)
    if lens == "orthographic":  # lens.py:266-285
        return np.sin(theta)
    if lens == "thoby":  # lens.py:313-335
        return 1.47 * np.sin(0.713 * theta)
    if lens == "rectilinear":  # lens.py:76-103
        if isinstance(theta, float):
            if theta < 0:
                raise ValueError("The angle theta cannot be negative")
            if theta > to_radians(89):
                raise ValueError("The Rectilinear lens can't handle FoV larger than 179 degrees")
            return np.tan(theta)
        bad = np.logical_or(theta < 0, theta > to_radians(89))
        out = np.tan(theta)
        out[bad] = np.nan
        return out
    raise KeyError(lens)


def lens_inverse(lens, r):
    """distance in focal-length units -> incidence angle (rad)."""
    if isinstance(lens, tuple):
        return lens[1](r)
    if lens == "equidistant":  # lens.py:148-165
        return r
    if lens == "equisolid":  # lens.py:191-220 (NaN -> 0.0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            theta = 2.0 * np.arcsin(r / 2.0)
        nan = np.isnan(theta)
        if isinstance(theta, float):
            return 0.0 if nan else theta
        theta[nan] = 0.0
        return theta
    if lens == "stereographic":  # lens.py:105-124
        return 2.0 * np.arctan(r / 2.0)
    if lens == "orthographic":  # lens.py:247-262 (NaN kept)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return np.arcsin(r)
    if lens == "thoby":  # lens.py:290-307
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return np.arcsin(r / 1.47) / 0.713
    if lens == "rectilinear":  # lens.py:68-73
        return np.arctan(r)
    raise KeyError(lens)


# --------------------------------------------------------------------------
# projection description
# --------------------------------------------------------------------------
@dataclass
class Proj:
    """One end of a remap: what photonbend's CameraImage / DoubleCameraImage /
    PanoramaImage objects hold besides the pixels."""

    kind: str  # "camera" | "double" | "pano"
    height: int
    width: int
    lens: str = "equidistant"
    fov: float = 0.0  # radians; per-sensor fov for "double"
    magnitude: Optional[float] = None
    f_distance: float = field(default=0.0, init=False)

    def __post_init__(self):
        if self.kind == "camera":
            # projection.py:118-121, :123-144
            if self.magnitude is None:
                self.magnitude = self.height / 2.0
            self.f_distance = self.magnitude / lens_forward(self.lens, self.fov / 2)
        elif self.kind == "double":
            # projection.py:315-316, :336-339 (magnitude kwarg is swallowed)
            self.magnitude = self.height / 2.0
            self.f_distance = self.magnitude / lens_forward(self.lens, self.fov / 2)
        elif self.kind != "pano":
            raise KeyError(self.kind)


def _atan2_via_clog(x, y):
    """np.log(make_complex(x, y)).imag - _shared.py:25-55, projection.py:193."""
    z = np.empty(np.broadcast(x, y).shape, dtype=np.complex128)
    z.real = x
    z.imag = y
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return np.log(z).imag


# --------------------------------------------------------------------------
# a-2 / a-3 / a-6  destination inverse projection -> coordinate map
# --------------------------------------------------------------------------
def coordinate_map(p: Proj) -> np.ndarray:
    H, W = p.height, p.width
    if p.kind == "pano":
        # projection.py:487-513: quarter-pixel longitude offset, row END points
        q = np.pi / W / 2
        lon = np.linspace(-np.pi + q, np.pi - q, num=W)
        lat = np.linspace(0, np.pi, num=H)
        out = np.zeros((H, W, 3), np.float64)
        out[:, :, 0] = lat[:, None]
        out[:, :, 1] = lon[None, :]
        return out

    if p.kind == "camera":
        # projection.py:171-194
        x = np.linspace(-W / 2 + 0.5, W / 2 - 0.5, num=W)[None, :]
        y = np.linspace(H / 2 - 0.5, -H / 2 + 0.5, num=H)[:, None]
        d = np.sqrt(x**2 + y**2) / p.f_distance
        lat = lens_inverse(p.lens, d)
        lon = _atan2_via_clog(x, y)
        invalid = lat > p.fov / 2  # projection.py:160
    else:
        # projection.py:370-406, :354-360
        half = W // 2
        hx = np.linspace(-half / 2 + 0.5, half / 2 - 0.5, num=half)
        x = np.concatenate([hx, hx * (-1)], 0)[None, :]
        y = np.linspace(H / 2 - 0.5, -H / 2 + 0.5, num=H)[:, None]
        d = np.sqrt(x**2 + y**2) / p.f_distance
        lat = lens_inverse(p.lens, d)
        lat[:, half:] *= -1
        lat[:, half:] += np.pi
        lon = _atan2_via_clog(x, y)
        invalid = lat > p.fov / 2.0
        invalid[:, half:] = lat[:, half:] < np.pi - (p.fov / 2.0)
    out = np.empty(lat.shape + (3,), np.float64)
    out[:, :, 0] = lat
    out[:, :, 1] = lon
    out[:, :, 2] = invalid
    return out


# --------------------------------------------------------------------------
# a-7 / a-8  rotation
# --------------------------------------------------------------------------
def rotation_matrix(pitch: float, yaw: float, roll: float) -> np.ndarray:
    """Rotation(pitch, yaw, roll).rotation_matrix - rotation.py:27-62 evaluated
    at the NEGATED angles (rotation.py:100)."""
    p, y, r = -pitch, -yaw, -roll
    cp, sp = np.cos(p), np.sin(p)
    cy, sy = np.cos(y), np.sin(y)
    cr, sr = np.cos(r), np.sin(r)
    P = np.array((1, 0, 0, 0, cp, sp, 0, -sp, cp)).reshape((3, 3))
    Y = np.array((cy, 0, -sy, 0, 1, 0, sy, 0, cy)).reshape((3, 3))
    R = np.array((cr, sr, 0, -sr, cr, 0, 0, 0, 1)).reshape((3, 3))
    return P @ Y @ R


def rotate_map(R: np.ndarray, cmap: np.ndarray) -> np.ndarray:
    """Rotation.rotate_coordinate_map - rotation.py:102-176.  Like the
    reference it zeroes lat/lon of invalid pixels IN the caller's array."""
    polar = cmap[:, :, :2]
    invalid = cmap[:, :, 2] != 0.0
    polar[invalid] = 0
    lat = polar[:, :, 0]
    lon = polar[:, :, 1]
    y = np.cos(lat)
    xz = np.exp(lon * 1j) * np.sin(lat)
    v = np.empty(lat.shape + (3, 1), np.float64)
    v[:, :, 0, 0] = xz.real
    v[:, :, 1, 0] = y
    v[:, :, 2, 0] = xz.imag
    # the accumulation order inside the 3x3 product is whatever the BLAS
    # behind np.matmul does (fma(R[i,2],z, fma(R[i,0],x, R[i,1]*y)))
    w = np.matmul(R, v, axes=[(-2, -1), (-2, -1), (-2, -1)])
    w = w.reshape(w.shape[:-1])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        new_lat = np.arccos(w[:, :, 1])
    new_lon = _atan2_via_clog(w[:, :, 0], w[:, :, 2])
    out = np.empty_like(cmap)
    out[:, :, 0] = new_lat
    out[:, :, 1] = new_lon
    out[:, :, :2][invalid] = 0
    out[:, :, 2] = invalid
    return out


# --------------------------------------------------------------------------
# a-4 / a-5 / a-6  source forward projection -> integer source position
# --------------------------------------------------------------------------
def _to_int(a):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with np.errstate(all="ignore"):
            return a.astype(int)


def pano_positions(h: int, w: int, cmap: np.ndarray):
    """projection.py:533-545 - returns (row, col, invalid, pre_r, pre_c).
    Zeroes invalid lat/lon in the caller's map like the reference."""
    invalid = cmap[:, :, 2] != 0.0
    polar = cmap[:, :, :2]
    polar[invalid] = 0
    wseg = np.pi / (w / 2)
    hseg = np.pi / h
    pre_r = polar[:, :, 0] / hseg
    pre_c = polar[:, :, 1] / wseg + (w / 2)
    return _to_int(pre_r) % h, _to_int(pre_c) % w, invalid, pre_r, pre_c


def camera_positions(p: Proj, h: int, w: int, lat, lon):
    """_make_cartesian_map - projection.py:247-260 - (py, px, pre_y, pre_x) for
    a fisheye of size h x w with p's lens and f_distance."""
    cy, cx = h / 2 - 0.5, w / 2 - 0.5
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with np.errstate(all="ignore"):
            dist = lens_forward(p.lens, lat) * p.f_distance
            z = np.exp(lon * 1j) * dist
            pre_y = (z.imag * (-1)) + cy
            pre_x = z.real + cx
    return _to_int(pre_y), _to_int(pre_x), pre_y, pre_x


def camera_index(p: Proj, h: int, w: int, cmap: np.ndarray):
    """CameraImage.process_coordinate_map minus the gather - projection.py:214-243.
    Returns (py, px, black) with py/px zeroed where out of range."""
    invalid = cmap[:, :, 2] != 0.0
    py, px, pre_y, pre_x = camera_positions(p, h, w, cmap[:, :, 0], cmap[:, :, 1])
    bad_y = np.logical_or(py >= h, py < 0)
    bad_x = np.logical_or(px >= w, px < 0)
    py[bad_y] = 0
    px[bad_x] = 0
    black = np.logical_or(np.logical_or(bad_y, bad_x), invalid)
    return py, px, black, pre_y, pre_x


def _double_sides(p: Proj):
    """The two CameraImage objects DoubleCameraImage.process_coordinate_map
    builds on the image halves - projection.py:429-434 (default magnitude)."""
    w2 = p.width // 2
    left = Proj("camera", p.height, w2, p.lens, p.fov)
    right = Proj("camera", p.height, p.width - w2, p.lens, p.fov)
    return left, right, w2


def double_weights(p: Proj, lat):
    """Blend ramps - projection.py:414-457.  lat is the LEFT latitude."""
    ref = (p.fov / 2) - (np.pi / 2)
    mn = np.pi / 2 - ref
    mx = np.pi / 2 + ref
    rng = 2.0 * ref
    safety = to_radians(0.5)
    lat_r = lat * (-1) + np.pi
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with np.errstate(all="ignore"):
            for side_lat in (lat, lat_r):
                band = np.logical_and(side_lat >= mn, side_lat <= (mx + safety))
                fac = (side_lat - mx) / rng * -1
                fac[np.logical_not(band)] = 1.0
                out.append(fac)
    return out[0], out[1], lat_r


def source_index(p: Proj, cmap: np.ndarray):
    """The integer coordinate map: linear index into the (h, w) source pixel
    grid, or -1 where the output pixel is black.  For "double" returns the
    tuple (index_left, index_right, weight_left, weight_right, invalid) with
    indices into the FULL side-by-side frame."""
    h, w = p.height, p.width
    if p.kind == "pano":
        r, c, invalid, _, _ = pano_positions(h, w, cmap)
        idx = (r * w + c).astype(np.int64)
        idx[invalid] = -1
        return idx.astype(np.int32)
    if p.kind == "camera":
        py, px, black, _, _ = camera_index(p, h, w, cmap)
        idx = py * w + px
        idx[black] = -1
        return idx.astype(np.int32)
    left, right, w2 = _double_sides(p)
    invalid = cmap[:, :, 2] != 0.0
    lat = cmap[:, :, 0]
    fl, fr, lat_r = double_weights(p, lat)
    rmap = np.copy(cmap)
    rmap[:, :, 0] = lat_r
    pyl, pxl, bl, _, _ = camera_index(left, h, w2, cmap)
    pyr, pxr, br, _, _ = camera_index(right, h, w - w2, rmap)
    il = pyl * w + pxl
    il[bl] = -1
    # the right half is mirrored before sampling (projection.py:430-431)
    ir = pyr * w + (w2 + ((w - w2) - 1 - pxr))
    ir[br] = -1
    return il.astype(np.int32), ir.astype(np.int32), fl, fr, invalid


def sample(p: Proj, image: np.ndarray, cmap: np.ndarray) -> np.ndarray:
    """src.process_coordinate_map(cmap) -> uint8 (H, W, 3)."""
    h, w = p.height, p.width
    assert image.shape[:2] == (h, w)
    if p.kind == "pano":
        r, c, invalid, _, _ = pano_positions(h, w, cmap)  # projection.py:545-546
        out = image[r, c]
        out[invalid] = 0
        return out
    if p.kind == "camera":
        py, px, black, _, _ = camera_index(p, h, w, cmap)  # projection.py:234-243
        out = image[py, px]
        out[black] = 0
        return out
    # double - projection.py:408-462
    left, right, w2 = _double_sides(p)
    invalid = cmap[:, :, 2] != 0.0
    fl, fr, lat_r = double_weights(p, cmap[:, :, 0])
    rmap = np.copy(cmap)
    rmap[:, :, 0] = lat_r
    limg = image[:, :w2]
    rimg = np.copy(image[:, w2:])[:, ::-1]
    lmap = sample(left, limg, cmap)
    rmapd = sample(right, rimg, rmap)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with np.errstate(all="ignore"):
            li = lmap.astype(np.float64) * fl[:, :, None]
            ri = rmapd.astype(np.float64) * fr[:, :, None]
            final = (li + ri).astype(np.uint8)
    final[invalid] = 0
    return final


def remap(dst: Proj, src: Proj, image: np.ndarray, rotations=()) -> np.ndarray:
    """The canonical three-stage sequence - core/__init__.py:66-92.
    ``rotations`` is a sequence of (pitch, yaw, roll) in radians."""
    cmap = coordinate_map(dst)
    for rot in rotations:
        cmap = rotate_map(rotation_matrix(*rot), cmap)
    return sample(src, image, cmap)


def remap_index(dst: Proj, src: Proj, rotations=()):
    cmap = coordinate_map(dst)
    for rot in rotations:
        cmap = rotate_map(rotation_matrix(*rot), cmap)
    return source_index(src, cmap)


def pretrunc(dst: Proj, src: Proj, rotations=()) -> Tuple[np.ndarray, ...]:
    """Pre-truncation float64 source coordinates (row-like, col-like) - used to
    build the *fragile mask*: pixels whose coordinate sits within a few ulps of
    an integer, where a last-bit difference in a transcendental flips the
    truncated index (hard part 3)."""
    cmap = coordinate_map(dst)
    for rot in rotations:
        cmap = rotate_map(rotation_matrix(*rot), cmap)
    if src.kind == "pano":
        _, _, _, a, b = pano_positions(src.height, src.width, cmap)
        return (a, b)
    if src.kind == "camera":
        _, _, a, b = camera_positions(src, src.height, src.width, cmap[:, :, 0], cmap[:, :, 1])
        return (a, b)
    left, right, w2 = _double_sides(src)
    lat_r = cmap[:, :, 0] * (-1) + np.pi
    _, _, a, b = camera_positions(left, src.height, w2, cmap[:, :, 0], cmap[:, :, 1])
    _, _, c, d = camera_positions(right, src.height, src.width - w2, lat_r, cmap[:, :, 1])
    return (a, b, c, d)


def fragile_mask(coords, rel=2.0**-40) -> np.ndarray:
    """True where any pre-truncation coordinate lies within ``rel`` (relative
    to its magnitude, floor 1.0) of an integer, or is not finite."""
    m = None
    with np.errstate(all="ignore"):
        for a in coords:
            near = np.abs(a - np.rint(a)) <= rel * np.maximum(np.abs(a), 1.0)
            near |= ~np.isfinite(a)
            m = near if m is None else (m | near)
    return m


def map_projection(cmap: np.ndarray) -> np.ndarray:
    """projection.py:550-599 - coordinate map -> colour map (red: latitude stretched to 0..255 over the
    valid pixels, green: longitude * 255 / 2pi, blue: invalid * 255).  Zeroes invalid lat/lon in ``cmap``."""
    invalid = cmap[:, :, 2] != 0.0
    valid = np.logical_not(invalid)
    polar = cmap[:, :, :2]
    polar[invalid] = 0
    dist = polar[:, :, 0]
    mn, mx = np.min(dist[valid]), np.max(dist[valid])
    factor = 255.0 / (mx - mn)
    nd = dist.copy()
    nd[valid] -= mn
    nd[valid] *= factor
    with np.errstate(all="ignore"):
        red = np.round(nd).astype(np.uint8)
        green = np.round((255.0 / (np.pi * 2)) * polar[:, :, 1]).astype(np.uint8)
    blue = (invalid.astype(np.uint8) * 255).astype(np.uint8)
    return np.stack([red, green, blue], axis=2)


def _bilinear_camera(p: Proj, h: int, w: int, image: np.ndarray, lat, lon, invalid):
    """One fisheye (or one eye of a double frame) sampled bilinearly -> (uint8 values, live mask)."""
    with np.errstate(all="ignore"):
        _, _, fy, fx = camera_positions(p, h, w, lat, lon)
        live = ~invalid & np.isfinite(fy) & np.isfinite(fx) & (fy >= 0) & (fy < h) & (fx >= 0) & (fx < w)
        fy = np.where(live, fy, 0.5)
        fx = np.where(live, fx, 0.5)
        sy, sx = fy - 0.5, fx - 0.5
        r0 = np.floor(sy).astype(np.int64)
        c0 = np.floor(sx).astype(np.int64)
        ty, tx = (sy - r0)[..., None], (sx - c0)[..., None]
        r1, c1 = r0 + 1, c0 + 1
        r0, r1 = np.clip(r0, 0, h - 1), np.clip(r1, 0, h - 1)
        c0, c1 = np.clip(c0, 0, w - 1), np.clip(c1, 0, w - 1)
        img = image.astype(np.float64)
        top = img[r0, c0] + tx * (img[r0, c1] - img[r0, c0])
        bot = img[r1, c0] + tx * (img[r1, c1] - img[r1, c0])
        val = np.clip(np.rint(top + ty * (bot - top)), 0, np.iinfo(image.dtype).max).astype(image.dtype)
    val[~live] = 0
    return val, live


def remap_bilinear(dst: Proj, src: Proj, image: np.ndarray, rotations=(), cmap: np.ndarray = None) -> np.ndarray:
    """OUR definition of the opt-in bilinear mode - the reference has no such behaviour, so this
    function is "parity unpinned": it pins the HIP kernels to a written-down definition, not to the reference.

    The continuous source coordinate is the reference's pre-truncation coordinate f (pixel k covers [k, k+1),
    centre k + 0.5); s = f - 0.5, i0 = floor(s), t = s - i0; taps clamped to the image (panorama columns wrap),
    float64 weights, round half to even.  Black where the nearest mode is black.  A double-fisheye source is the
    reference's blend (projection.py:439-460) of the two eyes' bilinear uint8 samples, each eye sampled like a camera
    source on its half of the frame (the right eye on the mirrored half).

    Round 5: `cmap` - a materialised (possibly edited) coordinate map used instead of dst's and the rotations; `image` any layout the
    reference's fancy indexing accepts - (h, w) or (h, w, C), uint8 or uint16: values round half to even and clip to the sample type's
    range; a double-fisheye source returns uint8 whatever the samples, like the reference's blend."""
    if cmap is None:
        cmap = coordinate_map(dst)
        for rot in rotations:
            cmap = rotate_map(rotation_matrix(*rot), cmap)
    if image.ndim == 2:
        if src.kind == "double":
            raise ValueError("operands could not be broadcast together")
        return remap_bilinear(dst, src, image[:, :, None], rotations, cmap)[:, :, 0]
    invalid = cmap[:, :, 2] != 0.0
    h, w = src.height, src.width
    if src.kind == "double":
        left, right, w2 = _double_sides(src)
        lat = cmap[:, :, 0]
        fl, fr, lat_r = double_weights(src, lat)
        l, _ = _bilinear_camera(left, h, w2, image[:, :w2], lat, cmap[:, :, 1], invalid)
        r, _ = _bilinear_camera(right, h, w - w2, np.copy(image[:, w2:])[:, ::-1], lat_r, cmap[:, :, 1], invalid)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                out = (l.astype(np.float64) * fl[..., None] + r.astype(np.float64) * fr[..., None]).astype(np.uint8)
        out[invalid] = 0
        return out
    if src.kind == "camera":
        val, _ = _bilinear_camera(src, h, w, image, cmap[:, :, 0], cmap[:, :, 1], invalid)
        return val
    with np.errstate(all="ignore"):
        _, _, _, fy, fx = pano_positions(h, w, cmap)
        live = ~invalid & np.isfinite(fy) & np.isfinite(fx)
        fy = np.where(live, fy, 0.5)
        fx = np.where(live, fx, 0.5)
        sy, sx = fy - 0.5, fx - 0.5
        r0 = np.floor(sy).astype(np.int64)
        c0 = np.floor(sx).astype(np.int64)
        ty, tx = (sy - r0)[..., None], (sx - c0)[..., None]
        r1, c1 = r0 + 1, c0 + 1
        r0, r1 = np.clip(r0, 0, h - 1), np.clip(r1, 0, h - 1)
        c0, c1 = np.clip(c0 % w, 0, w - 1), np.clip(c1 % w, 0, w - 1)
        img = image.astype(np.float64)
        top = img[r0, c0] + tx * (img[r0, c1] - img[r0, c0])
        bot = img[r1, c0] + tx * (img[r1, c1] - img[r1, c0])
        val = np.clip(np.rint(top + ty * (bot - top)), 0, np.iinfo(image.dtype).max).astype(image.dtype)
    val[~live] = 0
    return val
