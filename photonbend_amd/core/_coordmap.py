"""The lazy coordinate map.

The reference materialises a float64 (H, W, 3) coordinate map between its three
stages (core/__init__.py:42-49, 24 bytes per pixel).  Here
``get_coordinate_map()`` returns a ``CoordinateMap``: a *recipe* (destination
projection + the rotations applied so far).  ``process_coordinate_map`` turns a
recipe into one fused kernel launch; nothing pixel-sized exists in between.

The object still satisfies code that treats the map as an ndarray: ``shape``,
``dtype``, ``np.asarray(m)``, ``m[...]`` and ``m[...] = v`` materialise it on the
GPU (pb_coordmap_f64 / pb_rotate_f64) and from then on the ndarray is the truth -
an edited map is sampled through the materialised-map kernel
(pb_sample_map_u8), never through a stale recipe.
"""

from __future__ import annotations

import numpy as np

from .. import _native as nat


class CoordinateMap:
    __array_priority__ = 100

    def __init__(self, dst_proj: nat.pb_proj, rotations=(), device=None):
        self._dst = dst_proj
        self._rotations = [np.array(r, dtype=np.float64).reshape(3, 3) for r in rotations]
        self._device = device
        self._array = None
        self._zero_invalid = False  # a later stage zeroed invalid lat/lon "in place"

    @classmethod
    def from_array(cls, dst_proj: nat.pb_proj, array: np.ndarray) -> "CoordinateMap":
        """A map that is an ndarray from the start (a destination whose lens the host evaluated)."""
        m = cls(dst_proj)
        m._array = np.ascontiguousarray(array, dtype=np.float64)
        return m

    # -- recipe ---------------------------------------------------------------------
    @property
    def is_lazy(self) -> bool:
        return self._array is None

    @property
    def dst_proj(self) -> nat.pb_proj:
        return self._dst

    @property
    def rotations(self):
        return list(self._rotations)

    def rotated(self, matrix) -> "CoordinateMap":
        return CoordinateMap(self._dst, self._rotations + [matrix], self._device)

    def note_invalid_zeroed(self) -> None:
        """Rotation.rotate_coordinate_map and PanoramaImage.process_coordinate_map
        zero lat/lon of invalid pixels in the CALLER's map (rotation.py:119-125,
        projection.py:534-536); a recipe remembers that for when it is looked at."""
        self._zero_invalid = True

    # -- ndarray face ------------------------------------------------------------------
    @property
    def shape(self):
        return (self._dst.height, self._dst.width, 3)

    @property
    def dtype(self):
        return np.dtype(np.float64)

    @property
    def ndim(self):
        return 3

    def __len__(self):
        return self._dst.height

    def device_tensor(self):
        """The map as a float64 device array (H, W, 3), computed on the GPU."""
        t = nat.coordmap(self._dst, self._device)
        for R in self._rotations:
            t = nat.rotate(R, t)
        if self._zero_invalid:
            # lat / lon of invalid pixels become 0 IN PLACE - exactly the side effect pb_rotate_f64 has on its input map
            # (rotation.py:119-125); its output is dropped
            nat.rotate(np.eye(3), t)
        return t

    def materialize(self) -> np.ndarray:
        if self._array is None:
            self._array = nat.to_host(self.device_tensor())
        return self._array

    def __array__(self, dtype=None, copy=None):
        a = self.materialize()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __getitem__(self, key):
        return self.materialize()[key]

    def __setitem__(self, key, value):
        self.materialize()[key] = value

    def copy(self):
        if self._array is not None:
            return self._array.copy()
        c = CoordinateMap(self._dst, self._rotations, self._device)
        c._zero_invalid = self._zero_invalid
        return c

    def __repr__(self):
        state = "lazy" if self.is_lazy else "materialised"
        return f"<CoordinateMap {self.shape} {state}, {len(self._rotations)} rotation(s)>"
