"""Streaming many host-resident frames through one plan.

The remap itself takes tens of microseconds per frame; for frames that live in host memory the wall time is
PCIe (an 8192x4096 source is 1.8 ms of H2D, its 4096x4096 result 0.9 ms of D2H).  ``remap_frames`` keeps both
directions of the link busy: while frame k+1 uploads (one DMA out of the caller's page-locked array), the remap
kernel of frame k stores its output over PCIe straight into frame k's result ndarray, through ``depth`` rotating
device input buffers.  Outputs are yielded in order as fresh (page-locked, recycled) ndarrays.
"""

from __future__ import annotations

from typing import Iterable, Iterator

import numpy as np

from . import _hostpipe
from . import _native as nat
from .core.projection import _plan_for


def plan_for(dst_image, rotations, src_image) -> nat.Plan:
    """The plan of ``src_image.process_coordinate_map(rotations(dst_image.get_coordinate_map()))``.
    ``rotations``: sequence of ``Rotation`` objects (or 3x3 matrices), applied in order."""
    mats = [getattr(r, "rotation_matrix", r) for r in rotations]
    return _plan_for(dst_image._proj("dst"), mats, src_image._proj("src"))


def remap_frames(plan: nat.Plan, frames: Iterable[np.ndarray], depth: int = 3, interpolation: str = "nearest") -> Iterator[np.ndarray]:
    """Remaps an iterable of uint8 (h, w, 3) ndarrays with ``plan``; yields uint8 (H, W, 3) ndarrays in order
    (``_hostpipe.remap_frames``: upload stream + launch stream, page-locked results the kernel writes directly, no PyTorch)."""
    return _hostpipe.remap_frames(plan, frames, depth, interpolation)
