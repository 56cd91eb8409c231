"""f-2: the CLI facade.  CPU: the host-side rules (magnitude per type, sizes, fov validation, suffix and
overwrite behaviour).  GPU: every command against the pixel arrays the REFERENCE's own CLI produced on
the same synthetic PNG inputs (tests/golden/cli.npz, written by oracle/make_goldens.py --cli)."""

import math
import os

import numpy as np
import pytest
from click.testing import CliRunner
from PIL import Image

from oracle.synth import synth_frame, synth_image
from photonbend_amd.scripts import cli
from tests import helpers as H
from tests.cases import cli_cases


def test_magnitude_size_and_fov_rules():
    # commands/__init__.py:91-109
    assert cli.magnitude_for("inscribed", (100, 200, 3)) == 99.5
    assert cli.magnitude_for("cropped", (100, 200, 3)) == 99.5
    assert cli.magnitude_for("double", (100, 200, 3)) == 49.5
    assert cli.magnitude_for("full", (60, 80, 3)) == math.sqrt(39.5**2 + 29.5**2)
    with pytest.raises(ValueError):
        cli.magnitude_for("full", (1, 2, 3, 4))
    # commands/__init__.py:180-191, make_pano.py:142-149
    src = np.zeros((50, 100, 3), np.uint8)
    assert cli.camera_shape("double", src, None) == (50, 100, 3)
    assert cli.camera_shape("inscribed", src, 33) == (33, 33, 3)
    # commands/__init__.py:171-177
    with pytest.raises(ValueError):
        cli.radians_fov(179.0, "double")
    with pytest.raises(ValueError):
        cli.radians_fov(361.0, "inscribed")
    assert cli.radians_fov(180.0, "double") == math.pi


def test_output_suffix_and_overwrite_prompt(tmp_path):
    inp = tmp_path / "in.png"
    Image.fromarray(synth_frame(8, 16)).save(inp)
    bad = CliRunner().invoke(cli.main, ["make-photo", str(inp), "--type", "inscribed", "--lens", "equidistant", "--fov", "180", str(tmp_path / "out.bmp")])
    assert bad.exit_code == 1 and "JPG or PNG" in bad.output
    existing = tmp_path / "there.png"
    existing.write_bytes(b"x")
    no = CliRunner().invoke(cli.main, ["make-photo", str(inp), "--type", "inscribed", "--lens", "equidistant", "--fov", "180", str(existing)], input="maybe\nn\n")
    assert no.exit_code == 0 and "Exiting!" in no.output and existing.read_bytes() == b"x"
    missing = CliRunner().invoke(cli.main, ["make-pano", str(tmp_path / "nope.png"), "--type", "inscribed", "--lens", "equidistant", "--fov", "180", str(tmp_path / "o.png")])
    assert missing.exit_code != 0
    wrong = CliRunner().invoke(cli.main, ["make-pano", str(inp), "--type", "inscribed", "--lens", "fisheye", "--fov", "180", str(tmp_path / "o.png")])
    assert wrong.exit_code == 2  # click rejects the lens choice


@pytest.mark.gpu
@pytest.mark.parametrize("case", cli_cases(), ids=lambda c: c[0])
def test_cli_matches_reference_cli(case, tmp_path):
    name, cmd, opts, spec = case
    h, w, mask, layout = (*spec, "RGB")[:4]
    gold = np.load(os.path.join(H.GOLD, "cli.npz"))[name]
    inp, outp = tmp_path / "in.png", tmp_path / "out.png"
    Image.fromarray(synth_image(h, w, layout, frame=5, circle_mask=mask)).save(inp)
    res = CliRunner().invoke(cli.main, [cmd, str(inp), *opts, str(outp)])
    if gold.dtype.kind == "U":  # the reference CLI rejects this input: so must ours, with the same exception type
        assert str(gold).startswith("raises:") and type(res.exception).__name__ == str(gold)[7:], (res.output, res.exception)
        return
    assert res.exit_code == 0, (res.output, res.exception)
    got = np.asarray(Image.open(outp))
    assert got.shape == gold.shape and got.dtype == gold.dtype  # RGBA stays RGBA: nothing is converted
    d = np.abs(got.astype(np.int16) - gold.astype(np.int16))
    d = np.minimum(d, 256 - d)
    # (round 4: the device chain runs the reference's own libm / NumPy kernels bit for bit, so the float64 blend of a double-fisheye
    # source lands on the reference's bytes too - no 1-LSB allowance any more)
    assert int((d > 0).sum()) == 0, f"{int((d > 0).any(axis=2).sum())} pixels differ from the reference CLI"
