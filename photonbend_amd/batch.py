"""Streaming many host-resident frames through one plan.

The remap itself takes tens of microseconds per frame; for frames that live in host memory the wall time is
PCIe (an 8192x4096 source is 1.9 ms of H2D, its 4096x4096 result 0.9 ms of D2H).  ``remap_frames`` keeps the
three engines busy at once: while frame k is being remapped on the compute stream, frame k+1 is uploading on
the H2D stream and frame k-1 is downloading on the D2H stream, through ``depth`` rotating pinned staging
buffers.  Outputs are yielded in order as fresh ndarrays.
"""

from __future__ import annotations

from typing import Iterable, Iterator

import numpy as np
import torch

from . import _native as nat
from .core.projection import _plan_for


def plan_for(dst_image, rotations, src_image) -> nat.Plan:
    """The plan of ``src_image.process_coordinate_map(rotations(dst_image.get_coordinate_map()))``.
    ``rotations``: sequence of ``Rotation`` objects (or 3x3 matrices), applied in order."""
    mats = [getattr(r, "rotation_matrix", r) for r in rotations]
    return _plan_for(dst_image._proj("dst"), mats, src_image._proj("src"))


from .utils.hostcopy import par_copy as _par_copy


def remap_frames(plan: nat.Plan, frames: Iterable[np.ndarray], depth: int = 3) -> Iterator[np.ndarray]:
    """Remaps an iterable of uint8 (h, w, 3) ndarrays with ``plan``; yields uint8 (H, W, 3) ndarrays in order."""
    nat.require_gpu()
    depth = max(2, int(depth))
    dev = torch.device("cuda", torch.cuda.current_device())
    sh = (plan.src.height, plan.src.width, 3)
    dh = (plan.dst.height, plan.dst.width, 3)
    h_in = [torch.empty(sh, dtype=torch.uint8).pin_memory() for _ in range(depth)]
    h_out = [torch.empty(dh, dtype=torch.uint8).pin_memory() for _ in range(depth)]
    d_in = [torch.empty(sh, dtype=torch.uint8, device=dev) for _ in range(depth)]
    d_out = [torch.empty(dh, dtype=torch.uint8, device=dev) for _ in range(depth)]
    s_up, s_run, s_down = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    uploaded = [torch.cuda.Event() for _ in range(depth)]
    computed = [torch.cuda.Event() for _ in range(depth)]
    downloaded = [torch.cuda.Event() for _ in range(depth)]
    pending: list = []  # slots whose download has been queued, oldest first
    lib = nat.load()

    def drain_one():
        slot = pending.pop(0)
        downloaded[slot].synchronize()
        out = np.empty(dh, np.uint8)
        _par_copy(out, h_out[slot].numpy())
        return out

    k = 0
    for frame in frames:
        a = np.asarray(frame)
        if a.dtype != np.uint8 or tuple(a.shape) != sh:
            raise ValueError(f"frames must be uint8 {sh}, got {a.dtype} {tuple(a.shape)}")
        slot = k % depth
        if len(pending) == depth:  # the slot about to be reused still holds an undelivered result
            yield drain_one()
        _par_copy(h_in[slot].numpy(), a)  # the slot's previous upload finished before its result was delivered
        with torch.cuda.stream(s_up):
            d_in[slot].copy_(h_in[slot], non_blocking=True)
            uploaded[slot].record(s_up)
        with torch.cuda.stream(s_run):
            s_run.wait_event(uploaded[slot])
            nat.check(lib.pb_remap_u8(plan.handle, d_in[slot].data_ptr(), d_out[slot].data_ptr(), 1, 0, 0, int(s_run.cuda_stream)))
            computed[slot].record(s_run)
        with torch.cuda.stream(s_down):
            s_down.wait_event(computed[slot])
            h_out[slot].copy_(d_out[slot], non_blocking=True)
            downloaded[slot].record(s_down)
        pending.append(slot)
        k += 1
    while pending:
        yield drain_one()
