"""TEST INFRASTRUCTURE ONLY - deterministic synthetic frames.

v(f, r, c, ch) = low8(mix32((f*0x9E3779B1) ^ (r*0x85EBCA6B) ^ (c*0xC2B2AE35)
                              ^ (ch*0x27D4EB2F) ^ seed)),
mix32 = the murmur3 32-bit finaliser, all arithmetic uint32.  The HIP library
generates the same frames on the device (``pb_synth_frame_u8``), so no RNG
library is involved on either side.
"""

import numpy as np

_M = np.uint32(0xFFFFFFFF)


def _mix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32, copy=True)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def synth_frame(height: int, width: int, frame: int = 0, seed: int = 0, circle_mask: int = 0) -> np.ndarray:
    """uint8 (height, width, 3).  circle_mask: 0 = none; 1 = black outside the
    inscribed circle (single fisheye); 2 = black outside the two side-by-side
    inscribed circles of a double-fisheye frame."""
    with np.errstate(over="ignore"):
        r = (np.arange(height, dtype=np.uint32) * np.uint32(0x85EBCA6B))[:, None, None]
        c = (np.arange(width, dtype=np.uint32) * np.uint32(0xC2B2AE35))[None, :, None]
        ch = (np.arange(3, dtype=np.uint32) * np.uint32(0x27D4EB2F))[None, None, :]
        f = np.uint32((frame * 0x9E3779B1) & 0xFFFFFFFF) ^ np.uint32(seed & 0xFFFFFFFF)
        img = (_mix32(r ^ c ^ ch ^ f) & np.uint32(0xFF)).astype(np.uint8)
    if circle_mask:
        img *= circle_mask_u8(height, width, circle_mask)[:, :, None]
    return img


def circle_mask_u8(height: int, width: int, mode: int) -> np.ndarray:
    """1 inside the inscribed circle(s), 0 outside.  Integer arithmetic:
    (2*y+1-h)^2 + (2*x+1-wc)^2 <= d^2 with d = min(h, wc), per circle."""
    ys = 2 * np.arange(height, dtype=np.int64) + 1 - height
    if mode == 1:
        xs = 2 * np.arange(width, dtype=np.int64) + 1 - width
        d = min(height, width)
    else:
        half = width // 2
        xl = np.arange(width, dtype=np.int64) % half
        xs = 2 * xl + 1 - half
        d = min(height, half)
    return ((ys[:, None] ** 2 + xs[None, :] ** 2) <= d * d).astype(np.uint8)


def synth_image(height: int, width: int, layout: str = "RGB", frame: int = 0, circle_mask: int = 0) -> np.ndarray:
    """Synthetic images in the layouts Pillow hands to the reference unconverted: "RGB" uint8 (H, W, 3), "RGBA" uint8
    (H, W, 4), "L" uint8 (H, W), "I;16" uint16 (H, W), "RGB16" uint16 (H, W, 3) - built from synth_frame planes."""
    a = synth_frame(height, width, frame=frame, circle_mask=circle_mask)
    b = synth_frame(height, width, frame=frame + 1000, circle_mask=circle_mask)
    if layout == "RGB":
        return a
    if layout == "RGBA":
        return np.ascontiguousarray(np.concatenate([a, b[:, :, :1]], axis=2))
    if layout == "L":
        return np.ascontiguousarray(a[:, :, 0])
    if layout == "I;16":
        return a[:, :, 0].astype(np.uint16) | (b[:, :, 0].astype(np.uint16) << 8)
    if layout == "RGB16":
        return a.astype(np.uint16) | (b.astype(np.uint16) << 8)
    raise KeyError(layout)
