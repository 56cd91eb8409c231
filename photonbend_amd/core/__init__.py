"""photonbend_amd.core - the MI355X-native drop-in for ``photonbend.core``.

Terminology is the reference's (core/__init__.py:31-64): an *image* is a uint8
(H, W, 3) RGB array; a *coordinate map* holds (latitude, longitude, invalid) per
pixel; a *ProjectionImage* offers ``get_coordinate_map()`` and
``process_coordinate_map(map)``.  Typical use is unchanged (core/__init__.py:66-92):

    source = PanoramaImage(pano_arr)
    destiny = CameraImage(np.zeros((4096, 4096, 3), np.uint8), fov, equidistant(), magnitude=2047.5)
    cmap = destiny.get_coordinate_map()
    cmap = Rotation(pitch, yaw, roll).rotate_coordinate_map(cmap)   # optional, repeatable
    photo = source.process_coordinate_map(cmap)
"""

from . import lens, projection, rotation  # noqa: F401
from ._coordmap import CoordinateMap  # noqa: F401
