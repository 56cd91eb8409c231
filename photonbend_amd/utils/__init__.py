"""Angle helpers with the reference's operation order (photonbend/utils/__init__.py:27-50)."""

import math

__all__ = ["to_radians", "to_degrees"]


def to_radians(degrees: float) -> float:
    """degrees -> radians as ``degrees / 180 * pi`` (divide first; utils/__init__.py:37)."""
    return degrees / 180 * math.pi


def to_degrees(radians: float) -> float:
    """radians -> degrees as ``radians / pi * 180.0`` (utils/__init__.py:50)."""
    return radians / math.pi * 180.0
