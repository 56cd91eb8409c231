"""3-DoF rotation of coordinate maps - the drop-in for photonbend.core.rotation
(rotation.py:27-176).  The 3x3 matrix is host arithmetic (nine scalars, once);
applying it to a map is GPU work: appended to a lazy CoordinateMap (and executed
inside the fused remap kernel) or run by pb_rotate_f64 on a materialised map.
"""

from __future__ import annotations

import numpy as np

from .. import _native as nat
from ._coordmap import CoordinateMap


def _calculate_rotation_matrix(pitch: float, yaw: float, roll: float) -> np.ndarray:
    """R = P(pitch) @ Y(yaw) @ Rl(roll), rotation.py:27-62."""
    cp, sp = np.cos(pitch), np.sin(pitch)
    cy, sy = np.cos(yaw), np.sin(yaw)
    cr, sr = np.cos(roll), np.sin(roll)
    about_x = np.array([[1, 0, 0], [0, cp, sp], [0, -sp, cp]], dtype=np.float64)
    about_y = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]], dtype=np.float64)
    about_z = np.array([[cr, sr, 0], [-sr, cr, 0], [0, 0, 1]], dtype=np.float64)
    return about_x @ about_y @ about_z


class Rotation:
    """``Rotation(pitch, yaw, roll)`` in radians; the matrix is evaluated at the
    negated angles (rotation.py:92-100)."""

    def __init__(self, pitch: float, yaw: float, roll: float) -> None:
        self.rotation_matrix = _calculate_rotation_matrix(-pitch, -yaw, -roll)

    def rotate_coordinate_map(self, coordinate_map):
        """Returns the rotated map (rotation.py:102-176).  Like the reference it
        zeroes lat/lon of invalid pixels in the map it was given."""
        if isinstance(coordinate_map, CoordinateMap) and coordinate_map.is_lazy:
            coordinate_map.note_invalid_zeroed()
            return coordinate_map.rotated(self.rotation_matrix)
        if nat.is_device_array(coordinate_map):
            if nat.is_tensor(coordinate_map) and not (coordinate_map.is_cuda and coordinate_map.dtype == nat.torch.float64 and coordinate_map.is_contiguous()):
                raise TypeError("tensor coordinate maps must be contiguous float64 CUDA tensors")
            if not nat.is_tensor(coordinate_map) and coordinate_map.dtype != np.float64:
                raise TypeError("device coordinate maps must be float64")
            if len(coordinate_map.shape) != 3 or coordinate_map.shape[2] != 3:
                raise ValueError(f"a coordinate map has shape (H, W, 3), got {tuple(coordinate_map.shape)}")
            with nat.on_device(nat.device_index_of(coordinate_map)):
                return nat.rotate(self.rotation_matrix, coordinate_map)
        host = coordinate_map.materialize() if isinstance(coordinate_map, CoordinateMap) else coordinate_map
        if not (isinstance(host, np.ndarray) and host.dtype == np.float64 and host.ndim == 3 and host.shape[2] == 3):
            raise TypeError("coordinate_map must be a float64 array of shape (H, W, 3)")
        nat.require_gpu()
        dev = nat.to_device(host)
        out = nat.rotate(self.rotation_matrix, dev)
        host[...] = nat.to_host(dev)  # the in-place zeroing of invalid pixels
        return nat.to_host(out)
