"""photonbend_amd - MI355X (gfx950) native per-pixel lens remapper behind the
``photonbend.core`` API.  Hand-written HIP kernels in ``csrc/`` behind the C ABI
of ``include/photonbend_hip.h``; this package is the Python host side."""

__version__ = "0.1.0"

from . import core, utils  # noqa: F401
from .core.lens import Lens, equidistant, equisolid, orthographic, rectilinear, stereographic, thoby  # noqa: F401
from .core.projection import CameraImage, DoubleCameraImage, PanoramaImage, ProjectionImage, map_projection  # noqa: F401
from .core.rotation import Rotation  # noqa: F401
from .core._coordmap import CoordinateMap  # noqa: F401
