"""Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl"
is RCCL on ROCm, over xGMI inside a node).

Frames are independent, so a batch shards contiguously over the ranks and no
pixel ever crosses a link.  The ONLY data-path collective is a broadcast of the
parameter block - destination projection, rotation matrices, source projection,
about 200-800 bytes - from rank 0, so every device provably computes with
identical bits (f_distance and the rotation matrices are host-side float64
results).  On CPU the same code runs over gloo (tests).
"""

from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import _native as nat

try:  # the multi-process host logic rides on torch.distributed (nccl = RCCL on ROCm; gloo on CPU); a host without PyTorch binds the
    import torch  # same five steps through the C ABI instead (pb_comm_* / pb_bcast_params / pb_remap_batch_sharded)
    import torch.distributed as dist
except ImportError:
    torch = None

    class _NoDist:
        @staticmethod
        def is_available():
            return False

        @staticmethod
        def is_initialized():
            return False

    dist = _NoDist()

_MAGIC = 0x50424E44  # "PBND"
_HEADER = 2
_PROJ_FIELDS = 7
BLOCK_LEN = _HEADER + 2 * _PROJ_FIELDS + 9 * nat.PB_MAX_ROTATIONS  # fixed size: receivers need no length exchange


def _proj_to_list(p: nat.pb_proj) -> List[float]:
    return [float(p.kind), float(p.lens), float(p.height), float(p.width), p.fov, p.magnitude, p.f_distance]


def _proj_from_list(v: Sequence[float]) -> nat.pb_proj:
    return nat.make_proj(int(v[0]), int(v[2]), int(v[3]), int(v[1]), float(v[4]), float(v[5]), float(v[6]))


def pack_params(dst: nat.pb_proj, rotations, src: nat.pb_proj) -> np.ndarray:
    """float64[BLOCK_LEN]; integers are exact in float64 and the float64 fields
    keep their bits, so a broadcast reproduces the sender's plan exactly."""
    rots = np.asarray(list(rotations), dtype=np.float64).reshape(-1, 9)
    if rots.shape[0] > nat.PB_MAX_ROTATIONS:
        raise nat.PbError(f"at most {nat.PB_MAX_ROTATIONS} chained rotations are supported")
    block = np.zeros(BLOCK_LEN, dtype=np.float64)
    block[0] = _MAGIC
    block[1] = rots.shape[0]
    block[_HEADER : _HEADER + _PROJ_FIELDS] = _proj_to_list(dst)
    block[_HEADER + _PROJ_FIELDS : _HEADER + 2 * _PROJ_FIELDS] = _proj_to_list(src)
    block[_HEADER + 2 * _PROJ_FIELDS : _HEADER + 2 * _PROJ_FIELDS + rots.size] = rots.ravel()
    return block


def unpack_params(block: np.ndarray) -> Tuple[nat.pb_proj, List[np.ndarray], nat.pb_proj]:
    block = np.asarray(block, dtype=np.float64)
    if block.shape != (BLOCK_LEN,) or int(block[0]) != _MAGIC:
        raise nat.PbError("corrupt parameter block")
    n = int(block[1])
    if not 0 <= n <= nat.PB_MAX_ROTATIONS:
        raise nat.PbError("corrupt parameter block (rotation count)")
    dst = _proj_from_list(block[_HEADER : _HEADER + _PROJ_FIELDS])
    src = _proj_from_list(block[_HEADER + _PROJ_FIELDS : _HEADER + 2 * _PROJ_FIELDS])
    base = _HEADER + 2 * _PROJ_FIELDS
    rots = [block[base + 9 * k : base + 9 * (k + 1)].reshape(3, 3).copy() for k in range(n)]
    return dst, rots, src


def broadcast_params(block, device=None, src: int = 0) -> np.ndarray:
    """Broadcast the parameter block from rank ``src`` (RCCL when the process
    group is nccl and ``device`` is a GPU; gloo on CPU).  Without an initialised
    process group this is the identity (single process); with one - a group of
    ONE rank included - the collective really runs, so a single-GPU box exercises
    the same RCCL call an 8-GPU node makes."""
    if not (dist.is_available() and dist.is_initialized()):
        if block is None:
            raise nat.PbError("no parameter block on a single-process run")
        return np.asarray(block, dtype=np.float64)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(block, dtype=np.float64)).to(dev)
    else:
        t = torch.empty(BLOCK_LEN, dtype=torch.float64, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def shard_range(n_items: int, world: int, rank: int) -> range:
    """Contiguous, balanced split of ``n_items`` frames: the first
    ``n_items % world`` ranks take one extra."""
    if world < 1 or not 0 <= rank < world or n_items < 0:
        raise ValueError("bad shard request")
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def remap_batch_sharded(dst_proj, rotations, src_proj, load_frame, n_frames: int, device=None, chunk: int = 8):
    """Remap this rank's contiguous share of a batch of ``n_frames`` frames.

    ``load_frame(i)`` returns frame i as a uint8 CUDA tensor (h, w, 3).  Rank 0's
    parameters win (broadcast); returns (frame indices, list of output tensors).
    Frames are launched ``chunk`` at a time: one pb_remap_u8 call = ONE kernel
    launch whose grid spans the chunk's frames (the launch ramp and drain are paid
    once per chunk; the index math is simply repeated per frame)."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    block = pack_params(dst_proj, rotations, src_proj) if rank == 0 else None
    d, rots, s = unpack_params(broadcast_params(block, device=device))
    plan = nat.Plan(d, rots, s)
    mine = shard_range(n_frames, world, rank)
    outs = []
    ids = list(mine)
    for a in range(0, len(ids), chunk):
        part = ids[a : a + chunk]
        batch = torch.stack([load_frame(i) for i in part])
        outs.extend(plan.remap(batch).unbind(0))
    return ids, outs
