"""The parity case matrix shared by the golden generator (oracle/make_goldens.py),
the oracle-vs-golden tests and the HIP-vs-oracle tests.

A case = destination projection, optional rotations (degrees, as the CLI's
``-r pitch yaw roll``), source projection, and which synthetic frame feeds it.
Projections are written as (kind, height, width, lens, fov_degrees, magnitude).
``magnitude`` None means the class default (height / 2).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

P = Tuple[str, int, int, str, float, Optional[float]]


@dataclass
class Case:
    name: str
    dst: P
    src: P
    rotations: List[Tuple[float, float, float]] = field(default_factory=list)
    mask: int = 0  # synthetic-frame circle mask (0 none, 1 single, 2 double)
    keep_map: bool = False  # also pin the float64 coordinate map(s)


def cam(h, w, lens, fov, magnitude=None) -> P:
    return ("camera", h, w, lens, float(fov), magnitude)


def dbl(h, w, lens, fov) -> P:
    return ("double", h, w, lens, float(fov), None)


def pano(h, w) -> P:
    return ("pano", h, w, "equidistant", 0.0, None)


def inscribed(n):  # CLI rule commands/__init__.py:98-99
    return n / 2 - 0.5


def full_frame(h, w):  # commands/__init__.py:102-105
    return math.sqrt((w / 2.0 - 0.5) ** 2 + (h / 2.0 - 0.5) ** 2)


LENS_FOVS = {
    "equidistant": (140, 180, 195, 360),
    "equisolid": (140, 180, 360),
    "stereographic": (140, 180, 360),  # 360 degenerates (f_distance ~ 1e-15)
    "orthographic": (140, 180, 360),  # 360 degenerates (f_distance ~ 2.6e17)
    "rectilinear": (100, 140),  # > 178 raises at construction
    "thoby": (140, 180),
}


def small_cases() -> List[Case]:
    cs: List[Case] = []
    # A. pano source <- fisheye destination (make-photo), every lens x fov
    for lens, fovs in LENS_FOVS.items():
        for fov in fovs:
            cs.append(
                Case(f"A_photo_{lens}_{fov}", cam(48, 48, lens, fov, inscribed(48)), pano(64, 128), keep_map=(fov == fovs[0]))
            )
    # odd / non-square / default magnitude / full-frame magnitude
    cs.append(Case("A_photo_odd", cam(33, 35, "equidistant", 180), pano(50, 100), keep_map=True))
    cs.append(Case("A_photo_full", cam(32, 48, "rectilinear", 120, full_frame(32, 48)), pano(64, 128)))
    cs.append(Case("A_photo_pano1000", cam(40, 40, "equidistant", 360, inscribed(40)), pano(1000, 2000)))
    # B. fisheye source <- pano destination (make-pano)
    for lens, fovs in LENS_FOVS.items():
        for fov in fovs:
            cs.append(Case(f"B_pano_{lens}_{fov}", pano(40, 80), cam(48, 48, lens, fov, inscribed(48)), mask=1, keep_map=(lens == "equidistant" and fov == 140)))
    cs.append(Case("B_pano_odd", pano(31, 63), cam(33, 35, "equisolid", 190), keep_map=True))
    # C. fisheye <- fisheye (alter-photo), with and without rotation
    cs.append(Case("C_alter_eqd_eqs_rot", cam(48, 48, "equisolid", 360, inscribed(48)), cam(48, 48, "equidistant", 360, inscribed(48)), [(30, 45, 10)], keep_map=True))
    cs.append(Case("C_alter_ste_ort", cam(40, 40, "orthographic", 170, inscribed(40)), cam(44, 44, "stereographic", 200, inscribed(44)), [(-15, 100, 200)]))
    cs.append(Case("C_alter_rect_thoby", cam(36, 54, "thoby", 160), cam(40, 60, "rectilinear", 150, full_frame(40, 60)), [(5, -20, 33)]))
    cs.append(Case("C_alter_norot", cam(40, 40, "equisolid", 180, inscribed(40)), cam(56, 56, "equidistant", 200, inscribed(56))))
    # D. rotations on the pano paths, docs/scripts.md angle triples, a chain
    cs.append(Case("D_photo_rot", cam(48, 48, "equidistant", 360, inscribed(48)), pano(64, 128), [(30, 45, 10)], keep_map=True))
    cs.append(Case("D_photo_rot_m90", cam(48, 48, "equidistant", 180, inscribed(48)), pano(64, 128), [(-90, 0, 0)]))
    cs.append(Case("D_photo_rot_m90_195", cam(48, 48, "equisolid", 180, inscribed(48)), pano(64, 128), [(-90, 0, 195)]))
    cs.append(Case("D_pano_rot_m90_90", pano(40, 80), cam(48, 48, "equidistant", 360, inscribed(48)), [(-90, 0, 90)]))
    cs.append(Case("D_pano_chain", pano(40, 80), cam(48, 48, "equidistant", 360, inscribed(48)), [(10, 20, 30), (-40, 5, 77)], keep_map=True))
    cs.append(Case("D_pano_pano_rot", pano(32, 64), pano(48, 96), [(12, 34, 56)]))
    cs.append(Case("D_pano_pano_identity", pano(32, 64), pano(32, 64)))
    # E. double fisheye as source (Gear-360 stitch) and as destination
    for fov in (180, 195, 220):
        cs.append(Case(f"E_stitch_{fov}_masked", pano(32, 64), dbl(40, 80, "equidistant", fov), mask=2))
        cs.append(Case(f"E_stitch_{fov}_raw", pano(32, 64), dbl(40, 80, "equidistant", fov)))
    cs.append(Case("E_stitch_eqs_rot", pano(33, 66), dbl(40, 80, "equisolid", 200), [(3, 90, -7)], mask=2))
    cs.append(Case("E_double_dst", dbl(32, 64, "equidistant", 195), pano(48, 96), keep_map=True))
    cs.append(Case("E_double_dst_rot", dbl(32, 64, "equisolid", 190), cam(48, 48, "equidistant", 360, inscribed(48)), [(20, 30, 40)]))
    cs.append(Case("E_double_double", dbl(32, 64, "equidistant", 200), dbl(40, 80, "equidistant", 190), [(0, 15, 0)]))
    return cs


def full_cases() -> List[Case]:
    """BASELINE.json configs restated as API calls.  c4 = c2 x 512
    frames; c5 is additionally pinned at sensor fov 195 (docs/scripts.md:51)."""
    return [
        Case("c1", pano(2048, 4096), cam(3072, 3072, "equidistant", 360, inscribed(3072)), mask=1),
        Case("c2", cam(4096, 4096, "equidistant", 360, inscribed(4096)), pano(4096, 8192)),
        Case("c3", cam(4096, 4096, "equisolid", 360, inscribed(4096)), cam(4096, 4096, "equidistant", 360, inscribed(4096)), [(30, 45, 10)]),
        Case("c5_180", pano(4096, 8192), dbl(3888, 7776, "equidistant", 180), mask=2),
        Case("c5_195", pano(4096, 8192), dbl(3888, 7776, "equidistant", 195), mask=2),
    ]


def mid_cases() -> List[Case]:
    """1-2 k pixel reference pins (tests/golden/mid.json): the lenses BASELINE's configs do not touch, in both
    directions with a rotation - sizes at which the windowed hot kernel runs hundreds of LEAN / DIRECT tiles - and
    the degenerate identity / near-identity remaps whose pre-truncation coordinates sit on integers."""
    cs: List[Case] = []
    for lens, fov in (("stereographic", 200), ("orthographic", 170), ("thoby", 180)):
        cs.append(Case(f"M_photo_{lens}", cam(1280, 1280, lens, fov, inscribed(1280)), pano(1024, 2048), [(12, -30, 7)]))
        cs.append(Case(f"M_pano_{lens}", pano(768, 1536), cam(1280, 1280, lens, fov, inscribed(1280)), [(5, 60, -20)], mask=1))
    cs.append(Case("M_photo_rectilinear", cam(1024, 1536, "rectilinear", 120, full_frame(1024, 1536)), pano(1024, 2048), [(12, -30, 7)]))
    cs.append(Case("M_pano_rectilinear", pano(768, 1536), cam(1024, 1536, "rectilinear", 120, full_frame(1024, 1536)), [(5, 60, -20)]))
    cs.append(Case("M_alter_ste_thoby", cam(1152, 1152, "thoby", 190, inscribed(1152)), cam(1280, 1280, "stereographic", 220, inscribed(1280)), [(-8, 15, 100)], mask=1))
    # identity and near-identity: every pre-truncation coordinate is (near) an integer, the truncated index follows the
    # last bit of cos / sin / atan2 - the reference's own output is rounding noise there (SURVEY 7 hard part 3)
    cs.append(Case("M_ident_eqd", cam(768, 768, "equidistant", 180, inscribed(768)), cam(768, 768, "equidistant", 180, inscribed(768)), mask=1))
    cs.append(Case("M_ident_pano", pano(512, 1024), pano(512, 1024)))
    cs.append(Case("M_near_eqs", cam(640, 640, "equisolid", 180, inscribed(640)), cam(768, 768, "equisolid", 180, inscribed(768)), mask=1))
    cs.append(Case("M_ident_eqd_rot0", cam(768, 768, "equidistant", 180, inscribed(768)), cam(768, 768, "equidistant", 180, inscribed(768)), [(0, 0, 0)], mask=1))
    return cs


def case_by_name(name: str) -> Case:
    for c in small_cases() + full_cases() + mid_cases():
        if c.name == name:
            return c
    raise KeyError(name)


# ---- images and lenses beyond uint8 RGB + built-ins (tests/golden/generic.npz) ------------------------------
def custom_forward(theta):  # a user lens: NOT one of the six built-ins (thoby has 1.47 / 0.713)
    import numpy as np

    return 1.3 * np.sin(0.8 * theta)


def custom_reverse(r):
    import numpy as np

    return np.arcsin(r / 1.3) / 0.8


def thoby_like_forward(theta):  # the built-in thoby formulas as user callables: must reproduce the built-in's bytes
    import numpy as np

    return 1.47 * np.sin(0.713 * theta)


def thoby_like_reverse(r):
    import numpy as np

    return np.arcsin(r / 1.47) / 0.713


def generic_cases():
    """(name, Case, layout).  Lens name "custom" / "thobylike" = a Lens built from the callables above; layout = the
    source image's array layout (oracle.synth.synth_image)."""
    rot = [(10, 20, 30)]
    nine = [(5.0 * k, -7.0 * k, 3.0 * k + 1) for k in range(1, 10)]
    out = []
    for layout in ("RGBA", "L", "RGB16", "I;16"):
        out.append((f"G_pano_{layout}", Case("g", cam(40, 40, "equisolid", 190, inscribed(40)), pano(32, 64), rot), layout))
    out.append(("G_cam_RGBA", Case("g", pano(32, 64), cam(48, 48, "equidistant", 360, inscribed(48)), mask=1), "RGBA"))
    out.append(("G_cam_I;16", Case("g", pano(32, 64), cam(48, 48, "stereographic", 200, inscribed(48)), rot, mask=1), "I;16"))
    out.append(("G_double_RGBA", Case("g", pano(32, 64), dbl(40, 80, "equidistant", 195), mask=2), "RGBA"))
    out.append(("G_double_RGB16", Case("g", pano(32, 64), dbl(40, 80, "equidistant", 200), rot, mask=2), "RGB16"))
    out.append(("G_double_src_odd", Case("g", pano(32, 64), dbl(40, 81, "equidistant", 195), mask=2), "RGB"))
    out.append(("G_double_src_odd_rot", Case("g", cam(36, 36, "equidistant", 180, inscribed(36)), dbl(40, 79, "equisolid", 200), rot, mask=2), "RGB"))
    out.append(("G_double_dst_odd", Case("g", dbl(32, 65, "equidistant", 195), pano(48, 96)), "RGB"))
    out.append(("G_rot9", Case("g", cam(40, 40, "equidistant", 200, inscribed(40)), pano(32, 64), nine), "RGB"))
    out.append(("G_rot9_double_gray", Case("g", pano(24, 48), cam(40, 40, "equisolid", 190, inscribed(40)), nine, mask=1), "L"))
    out.append(("G_custom_dst", Case("g", cam(40, 40, "custom", 170, inscribed(40)), pano(32, 64), rot), "RGB"))
    out.append(("G_custom_src", Case("g", pano(32, 64), cam(48, 48, "custom", 170, inscribed(48)), rot, mask=1), "RGB"))
    out.append(("G_custom_src_norot_RGBA", Case("g", pano(32, 64), cam(48, 48, "custom", 175, inscribed(48)), mask=1), "RGBA"))
    out.append(("G_custom_double_src", Case("g", pano(32, 64), dbl(40, 80, "custom", 195), mask=2), "RGB"))
    out.append(("G_custom_double_dst", Case("g", dbl(32, 64, "custom", 190), pano(48, 96), rot), "RGB"))
    out.append(("G_custom_both", Case("g", cam(40, 40, "custom", 160, inscribed(40)), cam(48, 48, "custom", 170, inscribed(48)), rot, mask=1), "I;16"))
    out.append(("G_thobylike_dst", Case("g", cam(40, 40, "thobylike", 180, inscribed(40)), pano(32, 64), rot), "RGB"))
    out.append(("G_thobylike_src", Case("g", pano(32, 64), cam(48, 48, "thobylike", 180, inscribed(48)), mask=1), "RGB"))
    return out


def cli_cases():
    """(name, command, option list, (input height, input width, circle mask)) for the CLI facade (f-2).
    INPUT / OUTPUT paths are appended by the runner; outputs are PNG (lossless)."""
    return [
        ("photo_inscribed", "make-photo", ["--type", "inscribed", "--lens", "equidistant", "--fov", "360", "-s", "72"], (64, 128, 0)),
        ("photo_rot2", "make-photo", ["--type", "inscribed", "--lens", "equisolid", "--fov", "180", "-r", "-90", "0", "195", "-r", "10", "20", "30"], (48, 96, 0)),
        ("photo_double", "make-photo", ["--type", "double", "--lens", "equidistant", "--fov", "195", "-s", "40"], (64, 128, 0)),
        ("photo_full_rect", "make-photo", ["--type", "full", "--lens", "rectilinear", "--fov", "120", "-s", "50"], (64, 128, 0)),
        ("photo_cropped_stereo", "make-photo", ["--type", "cropped", "--lens", "stereographic", "--fov", "200"], (40, 80, 0)),
        ("pano_inscribed", "make-pano", ["--type", "inscribed", "--lens", "equidistant", "--fov", "360", "-s", "48"], (72, 72, 1)),
        ("pano_double_195", "make-pano", ["--type", "double", "--lens", "equidistant", "--fov", "195"], (40, 80, 2)),
        ("pano_ortho_rot", "make-pano", ["--type", "inscribed", "--lens", "orthographic", "--fov", "170", "-r", "15", "-40", "5"], (60, 60, 1)),
        ("alter_eqd_eqs_rot", "alter-photo", ["--itype", "inscribed", "--ilens", "equidistant", "--ifov", "360", "--otype", "inscribed", "--olens", "equisolid", "--ofov", "360", "-r", "30", "45", "10"], (64, 64, 1)),
        ("alter_size_quirk", "alter-photo", ["--itype", "inscribed", "--ilens", "equidistant", "--ifov", "180", "--otype", "full", "--olens", "rectilinear", "--ofov", "100", "-s", "40"], (64, 64, 1)),
        ("alter_double_in", "alter-photo", ["--itype", "double", "--ilens", "equidistant", "--ifov", "200", "--otype", "inscribed", "--olens", "equidistant", "--ofov", "180"], (48, 96, 2)),
        # inputs that are not RGB: the reference hands Pillow's array over unconverted (commands/__init__.py:135-143)
        ("photo_rgba", "make-photo", ["--type", "inscribed", "--lens", "equisolid", "--fov", "200", "-s", "60", "-r", "10", "20", "30"], (64, 128, 0, "RGBA")),
        ("pano_gray", "make-pano", ["--type", "inscribed", "--lens", "equidistant", "--fov", "360", "-s", "40"], (72, 72, 1, "L")),
        ("pano_gray16", "make-pano", ["--type", "inscribed", "--lens", "stereographic", "--fov", "200", "-r", "15", "-40", "5"], (60, 60, 1, "I;16")),
        ("alter_rgba_double_in", "alter-photo", ["--itype", "double", "--ilens", "equidistant", "--ifov", "200", "--otype", "inscribed", "--olens", "equidistant", "--ofov", "180"], (48, 96, 2, "RGBA")),
    ]
