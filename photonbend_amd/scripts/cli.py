"""The three photonbend commands on top of the GPU core - SURVEY 8 f-2.

Same command names, options and rules as the reference CLI (photonbend/scripts/main.py:28-35,
commands/make_photo.py:52-141, alter_photo.py:51-162, make_pano.py:54-149, commands/__init__.py:53-191):
image type -> class and magnitude, output size, fov validation, any number of ``-r pitch yaw roll``
applied in order, the .jpg/.jpeg/.png suffix rule and the overwrite prompt.  Everything here is host
plumbing (Pillow decode/encode dominates its wall time); the remap is one pb_remap_u8 call.
"""

from __future__ import annotations

import math
import sys
from pathlib import Path
from typing import Optional, Sequence, Tuple

import click
import numpy as np
from PIL import Image

from .. import core
from ..core.lens import equidistant, equisolid, orthographic, rectilinear, stereographic
from ..core.projection import CameraImage, DoubleCameraImage, PanoramaImage
from ..core.rotation import Rotation
from ..utils import to_radians

LENSES = {
    "equidistant": equidistant,
    "equisolid": equisolid,
    "orthographic": orthographic,
    "rectilinear": rectilinear,
    "stereographic": stereographic,
}
TYPES = ("inscribed", "double", "cropped", "full")

TYPE_HELP = """

    \b
    The choices are:
    - inscribed: The valid data is on a inscribed circle.
    - double: The valid data is on two inscribed side-by-side circles.
    - cropped: The valid data is on a inscribed circle, top-and-bottom cropped.
    - full: The whole area of the image is valid data.
    """
DOUBLE_FOV_NOTE = "\n\n    IMPORTANT: FoV for double images are the value for one of the sensors and > 180."
ROTATION_HELP = "The rotation that should be applied to the camera: <pitch yaw roll> in degrees. Repeatable."


# ---- rules ------------------------------------------------------------------------------------
def checked_output(path: Path) -> Path:
    """Suffix rule and overwrite prompt (commands/__init__.py:53-70)."""
    out = Path(path)
    if out.suffix.lower() not in (".jpg", ".jpeg", ".png"):
        print("The desired output image should be a JPG or PNG file.")
        print("Provide an output filename ending in either JPG, JPEG or PNG (case insensitive)")
        print("Exiting!")
        sys.exit(1)
    if out.exists():
        answer = ""
        while answer not in ("y", "n"):
            answer = input("File already exists. Overwrite? (y/n) ")
        if answer == "n":
            print("Exiting!")
            sys.exit(0)
    return out


def open_image(path: Path) -> np.ndarray:
    """The decoded pixels exactly as Pillow hands them over (commands/__init__.py:135-143): RGB, RGBA, grey
    ("L" -> (H, W)), 16-bit - nothing is converted, the remap gathers whatever the array holds, like the reference."""
    try:
        with Image.open(path) as im:
            return np.asarray(im)
    except IOError:
        print("Error: Input image could not be opened!")
        print("Exiting!")
        sys.exit(1)


def magnitude_for(image_type: str, shape: Sequence[int]) -> float:
    """Pixels from the centre to where the full fov is reached (commands/__init__.py:91-109)."""
    if len(shape) > 3:
        raise ValueError("Can't calculate magnitude of images with more than 3 dimensions")
    height, width, _ = shape  # like the reference (:96): a grey (H, W) array does not unpack - ValueError
    if image_type == "double":
        return height / 2 - 0.5
    if image_type == "full":
        return math.sqrt((width / 2.0 - 0.5) ** 2 + (height / 2.0 - 0.5) ** 2)
    return width / 2 - 0.5  # inscribed, cropped


def radians_fov(fov: float, image_type: str) -> float:
    """fov rules (commands/__init__.py:171-177)."""
    if image_type == "double" and fov < 180:
        raise ValueError("The fov of a double image can't be smaller than 180 degrees.")
    if fov > 360:
        raise ValueError("The fov of an image can't be higher than 360 degrees.")
    return to_radians(fov)


def camera_shape(image_type: str, source: np.ndarray, height: Optional[int]) -> Tuple[int, int, int]:
    """Shape of a fisheye destination (commands/__init__.py:180-191)."""
    h = source.shape[0] if height is None else height
    return (h, 2 * h, 3) if image_type == "double" else (h, h, 3)


def camera_object(image_type: str, pixels: np.ndarray, fov: float, lens: str, magnitude: float):
    """CameraImage or DoubleCameraImage for the type (commands/__init__.py:84-88)."""
    cls = DoubleCameraImage if image_type == "double" else CameraImage
    return cls(pixels, fov, LENSES[lens](), magnitude=magnitude)


def run_chain(source, destiny, rotations, out: Path) -> None:
    """dst.get_coordinate_map() -> rotations in order -> src.process_coordinate_map() -> save."""
    cmap = destiny.get_coordinate_map()
    for rot in rotations:
        cmap = Rotation(*map(to_radians, rot)).rotate_coordinate_map(cmap)
    mapped = source.process_coordinate_map(cmap)
    try:
        Image.fromarray(np.ascontiguousarray(mapped)).save(out)
    except IOError:
        print("Could not save to the specified location!")
        print("Exiting!")
        sys.exit(1)


# ---- commands -----------------------------------------------------------------------------------
_lens_choice = click.Choice(list(LENSES))
_type_choice = click.Choice(list(TYPES))


def _common(fn):
    fn = click.option("-s", "--size", type=click.INT, default=None, help="The vertical size of the destiny image")(fn)
    fn = click.option("-r", "--rotation", type=click.FLOAT, nargs=3, multiple=True, default=[], help=ROTATION_HELP)(fn)
    return fn


@click.group()
def main():
    """photonbend commands on the MI355X remapper."""


@main.command("make-photo")
@click.argument("input_image", type=click.Path(exists=True, path_type=Path))
@click.option("--type", "otype", required=True, type=_type_choice, help="The type of the output image. " + TYPE_HELP)
@click.option("--lens", required=True, type=_lens_choice, help="The lens type to be used on the output photo.")
@click.option("--fov", required=True, type=click.FLOAT, help="The lens field of view of the output photo in degrees. " + DOUBLE_FOV_NOTE)
@_common
@click.argument("output_image", type=click.Path(exists=False, path_type=Path))
def make_photo(input_image, otype, lens, fov, output_image, rotation, size):
    """Make a photo out of a panorama.

    \b
    INPUT is the path to the source panorama.
    OUTPUT is the desired path of the destiny photo.
    """
    out = checked_output(output_image)
    pano = open_image(input_image)
    _, _, _ = pano.shape  # make_photo.py:112 unpacks three dimensions: grey inputs are a ValueError in the reference CLI
    shape = camera_shape(otype, pano, size)
    destiny = camera_object(otype, np.zeros(shape, np.uint8), radians_fov(fov, otype), lens, magnitude_for(otype, shape))
    run_chain(PanoramaImage(pano), destiny, rotation, out)


@main.command("alter-photo")
@click.argument("input_image", type=click.Path(exists=True, path_type=Path))
@click.option("--itype", required=True, type=_type_choice, help="The type of the input image. " + TYPE_HELP)
@click.option("--ilens", required=True, type=_lens_choice, help="The lens type that was used on the input photo.")
@click.option("--ifov", required=True, type=click.FLOAT, help="The lens field of view of the input photo in degrees. " + DOUBLE_FOV_NOTE)
@click.option("--otype", required=True, type=_type_choice, help="The type of the output image." + TYPE_HELP)
@click.option("--olens", required=True, type=_lens_choice, help="The lens type of the output photo. " + DOUBLE_FOV_NOTE)
@click.option("--ofov", required=True, type=click.FLOAT, help="The lens field of view of the output photo in degrees.")
@click.argument("output_image", type=click.Path(exists=False, path_type=Path))
@_common
def alter_photo(input_image, itype, ilens, ifov, otype, olens, ofov, output_image, rotation, size):
    """Change the the lens and FoV of a photo.

    \b
    INPUT is the path to the source photo.
    OUTPUT is the desired path of the destiny photo.
    """
    out = checked_output(output_image)
    photo = open_image(input_image)
    source = camera_object(itype, photo, radians_fov(ifov, itype), ilens, magnitude_for(itype, photo.shape))
    shape = camera_shape(otype, photo, size)
    # the destination magnitude comes from the SOURCE shape (alter_photo.py:142): only visible when --size differs
    destiny = camera_object(otype, np.zeros(shape, np.uint8), radians_fov(ofov, otype), olens, magnitude_for(otype, photo.shape))
    run_chain(source, destiny, rotation, out)


@main.command("make-pano")
@click.argument("input_image", type=click.Path(exists=True, path_type=Path))
@click.option("--type", "itype", required=True, type=_type_choice, help="The type of the input image. " + TYPE_HELP)
@click.option("--lens", required=True, type=_lens_choice, help="The lens type that was used on the input photo.")
@click.option("--fov", required=True, type=click.FLOAT, help="The lens field of view of the input photo in degrees. " + DOUBLE_FOV_NOTE)
@_common
@click.argument("output_image", type=click.Path(exists=False, path_type=Path))
def make_pano(input_image, itype, lens, fov, output_image, rotation, size):
    """Make a panorama out of a photo.

    \b
    INPUT is the path to the source photo.
    OUTPUT is the desired path of the destiny panorama.
    """
    out = checked_output(output_image)
    photo = open_image(input_image)
    source = camera_object(itype, photo, radians_fov(fov, itype), lens, magnitude_for(itype, photo.shape))
    h = photo.shape[0] if size is None else size
    destiny = PanoramaImage(np.zeros((h, int(h * 2), 3), np.uint8))  # make_pano.py:142-149
    run_chain(source, destiny, rotation, out)


if __name__ == "__main__":
    main()
