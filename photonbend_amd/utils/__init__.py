"""Angle helpers and the panorama -> photo size rule, with the reference's operation order
(photonbend/utils/__init__.py:27-118).  Host scalars only; nothing here touches a pixel."""

import math
from typing import Callable, Tuple

__all__ = ["to_radians", "to_degrees", "calculate_size_panorama_to_photo"]


def to_radians(degrees: float) -> float:
    """degrees -> radians as ``degrees / 180 * pi`` (divide first; utils/__init__.py:37)."""
    return degrees / 180 * math.pi


def to_degrees(radians: float) -> float:
    """radians -> degrees as ``radians / pi * 180.0`` (utils/__init__.py:50)."""
    return radians / math.pi * 180.0


def _lens_ratio(lens_function: Callable[[float], float]) -> float:
    # radius of the 360-degree circle over the radius of the 180-degree circle (utils/__init__.py:58-60, :72-74)
    return lens_function(math.pi) / lens_function(math.pi / 2)


def calculate_size_panorama_to_photo(
    panorama_size: Tuple[int, int],
    lens_function: Callable[[float], float],
    preserve_vertical_resolution: bool = False,
) -> Tuple[int, int]:
    """(width, height) of the inscribed photo that keeps a panorama's pixel detail (utils/__init__.py:81-118).

    Horizontal rule (:53-64): the panorama's equator is ``width`` pixels long, so the 180-degree circle of the
    photo gets the diameter ``width / pi`` and the full circle that times the lens's 360 / 180 radius ratio,
    rounded up.  With ``preserve_vertical_resolution`` the vertical rule (:67-78) - ``height`` over the smaller of
    the ratio and one minus it - wins when it asks for more.  A panorama that is not 2:1 trips the same assertion."""
    width, height = panorama_size
    assert width == 2 * height, "Equirectangular panoramas should have width and height in a 2:1 proportion"
    ratio = _lens_ratio(lens_function)
    side = int(math.ceil(width / math.pi * ratio))
    if preserve_vertical_resolution:
        small_side_factor = 1.0 / (1.0 - ratio if ratio > 0.5 else ratio)
        v_side = abs(int(math.ceil(height * small_side_factor)))
        if v_side > side:
            side = v_side
    return (side, side)
