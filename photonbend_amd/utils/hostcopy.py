"""Host-side frame copies for the NumPy-in / NumPy-out boundary.

A single-threaded memcpy of a 100 MB frame costs several times its PCIe transfer, so frames move
between user arrays and the page-locked staging buffers in row blocks on a few threads (NumPy releases
the GIL while copying), and in chunks, so that one chunk crosses PCIe while the next is being copied."""

from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np

_POOL = ThreadPoolExecutor(max_workers=8)


def par_copy(dst: np.ndarray, src: np.ndarray, parts: int = 8) -> None:
    """``dst[...] = src`` split along axis 0 over a few threads."""
    n = dst.shape[0]
    if n < 4 * parts or dst.nbytes < (4 << 20):
        dst[...] = src
        return
    step = (n + parts - 1) // parts
    futs = [_POOL.submit(np.copyto, dst[i : i + step], src[i : i + step]) for i in range(0, n, step)]
    for f in futs:
        f.result()


def row_chunks(rows: int, nbytes: int, target: int = 32 << 20):
    """(start, stop) row ranges of about ``target`` bytes each (one range for small frames)."""
    k = max(1, min(8, nbytes // target))
    step = (rows + k - 1) // k
    return [(i, min(i + step, rows)) for i in range(0, rows, step)]
