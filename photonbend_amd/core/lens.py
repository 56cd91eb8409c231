"""Lens models - the drop-in for photonbend.core.lens (lens.py:48-64, :341-412).

A ``Lens`` is a pair of callables (forward: incidence angle -> distance in
focal-length units, reverse: the inverse).  The built-in factories return
callables that (a) work on the host for scalars/arrays exactly like the
reference's (they are needed there: f_distance = magnitude / forward(fov / 2) is
a host-side scalar, projection.py:141-144) and (b) carry a ``pb_lens`` id, which
is what the HIP kernels dispatch on.  A Lens built from user callables has no
id: its functions are evaluated by the host on planes the device supplies (the
exact radius mesh of a destination, the latitude plane of a source) and the
device does everything else - index map, gather, blend (PB_LENS_CUSTOM,
pb_index_from_map_i32's distance planes; projection.py of this package).
"""

from __future__ import annotations

import warnings
from dataclasses import dataclass
from typing import Callable

import numpy as np

from ..utils import to_radians
from .._native import LENS_IDS


@dataclass
class Lens:
    forward_function: Callable
    reverse_function: Callable


def lens_id(lens_or_fn) -> int | None:
    """pb_lens id of a built-in lens (or one of its functions), else None."""
    if isinstance(lens_or_fn, Lens):
        a = getattr(lens_or_fn.forward_function, "pb_lens_id", None)
        b = getattr(lens_or_fn.reverse_function, "pb_lens_id", None)
        return a if (a is not None and a == b) else None
    return getattr(lens_or_fn, "pb_lens_id", None)


def _tag(name):
    def deco(fn):
        fn.pb_lens_id = LENS_IDS[name]
        fn.pb_lens_name = name
        return fn

    return deco


# -- equidistant (lens.py:148-187) ----------------------------------------------
@_tag("equidistant")
def _equidistant(theta):
    return theta


@_tag("equidistant")
def _equidistant_inverse(r):
    return r


# -- equisolid (lens.py:191-243) -------------------------------------------------
@_tag("equisolid")
def _equisolid(theta):
    return 2 * np.sin(theta / 2.0)


@_tag("equisolid")
def _equisolid_inverse(r):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        theta = 2.0 * np.arcsin(r / 2.0)
    if isinstance(theta, float):
        return 0.0 if np.isnan(theta) else theta
    theta[np.isnan(theta)] = 0.0  # outside the image circle -> 0.0, lens.py:219
    return theta


# -- stereographic (lens.py:105-145) ---------------------------------------------
@_tag("stereographic")
def _stereographic(theta):
    return 2.0 * np.tan(theta / 2.0)


@_tag("stereographic")
def _stereographic_inverse(r):
    return 2.0 * np.arctan(r / 2.0)


# -- orthographic (lens.py:247-285) -----------------------------------------------
@_tag("orthographic")
def _orthographic(theta):
    return np.sin(theta)


@_tag("orthographic")
def _orthographic_inverse(r):
    return np.arcsin(r)


# -- rectilinear (lens.py:68-103) --------------------------------------------------
@_tag("rectilinear")
def _rectilinear(theta):
    limit = to_radians(89)
    if isinstance(theta, float):
        if theta < 0:
            raise ValueError("The angle theta cannot be negative")
        if theta > limit:
            raise ValueError("The Rectilinear lens can't handle FoV larger than 179 degrees")
        return np.tan(theta)
    out = np.tan(theta)
    out[np.logical_or(theta < 0, theta > limit)] = np.nan
    return out


@_tag("rectilinear")
def _rectilinear_inverse(r):
    return np.arctan(r)


# -- thoby (lens.py:290-335) --------------------------------------------------------
@_tag("thoby")
def _thoby(theta):
    return 1.47 * np.sin(0.713 * theta)


@_tag("thoby")
def _thoby_inverse(r):
    return np.arcsin(r / 1.47) / 0.713


def rectilinear() -> Lens:
    r"""$f(\theta) = \tan\theta$ (lens.py:341-348)."""
    return Lens(_rectilinear, _rectilinear_inverse)


def equisolid() -> Lens:
    r"""$f(\theta) = 2\sin(\theta/2)$ (lens.py:351-358)."""
    return Lens(_equisolid, _equisolid_inverse)


def equidistant() -> Lens:
    r"""$f(\theta) = \theta$ (lens.py:361-370)."""
    return Lens(_equidistant, _equidistant_inverse)


def orthographic() -> Lens:
    r"""$f(\theta) = \sin\theta$ (lens.py:373-380)."""
    return Lens(_orthographic, _orthographic_inverse)


def stereographic() -> Lens:
    r"""$f(\theta) = 2\tan(\theta/2)$ (lens.py:383-390)."""
    return Lens(_stereographic, _stereographic_inverse)


def thoby() -> Lens:
    r"""$f(\theta) = 1.47\sin(0.713\,\theta)$ (lens.py:393-401)."""
    return Lens(_thoby, _thoby_inverse)


__all__ = ["Lens", "equisolid", "equidistant", "rectilinear", "stereographic", "orthographic", "thoby"]
