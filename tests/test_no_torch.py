"""VERDICT r3 missing 4: the reference needs numpy / Pillow / click (pyproject.toml:9-14); the drop-in must not need PyTorch.
Both tests run a child interpreter in which ``import torch`` FAILS (sys.modules['torch'] = None): on CPU the package imports, builds
lazy recipes and refuses to remap without a GPU (PbError, no fallback); on the GPU box the NumPy workflow of core/__init__.py:66-92 -
ndarray in, fresh ndarray out - runs end to end on the library's own device memory, streams and page-locked host memory
(_device.py / _hostpipe.py) and matches the oracle bit for bit."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CPU = r"""
import sys
sys.modules['torch'] = None            # `import torch` raises ImportError from here on
import numpy as np
import photonbend_amd as pb
from photonbend_amd import _native as nat, batch, parallel, _hostpipe
assert nat.torch is None and 'torch' not in [m for m in sys.modules if sys.modules[m] is not None and m.split('.')[0] == 'torch']
assert nat.load().pb_abi_version() == nat.ABI_VERSION
dst = pb.CameraImage(np.zeros((48, 48, 3), np.uint8), pb.utils.to_radians(180), pb.equidistant())
cmap = pb.Rotation(0.1, 0.2, 0.3).rotate_coordinate_map(dst.get_coordinate_map())
assert cmap.shape == (48, 48, 3) and cmap.is_lazy and len(cmap.rotations) == 1
assert pb.utils.calculate_size_panorama_to_photo is not None
if nat.device_count() == 0:
    try:
        pb.PanoramaImage(np.zeros((32, 64, 3), np.uint8)).process_coordinate_map(cmap)
        print("NO-ERROR"); sys.exit(3)
    except nat.PbError as e:
        assert "no HIP device" in str(e)
print("OK")
"""

_GPU = r"""
import sys
sys.modules['torch'] = None
sys.path.insert(0, '.')
import numpy as np
import photonbend_amd as pb
from photonbend_amd import _native as nat, batch
from oracle import reference_path as orc      # the checker
from oracle.synth import synth_frame, synth_image
assert nat.torch is None
h = 192
rot = tuple(map(pb.utils.to_radians, (30, 45, 10)))
fov = pb.utils.to_radians(360)
frames = [synth_frame(h, 2 * h, frame=f) for f in range(5)]
dst = pb.CameraImage(np.zeros((h, h, 3), np.uint8), fov, pb.equidistant(), magnitude=h / 2 - 0.5)
od, os_ = orc.Proj("camera", h, h, "equidistant", fov, h / 2 - 0.5), orc.Proj("pano", h, 2 * h)
want = [orc.remap(od, os_, f, [rot]) for f in frames]
# 1. the facade: ndarray in, fresh ndarray out; the first use runs the deferred (float64) plan, the second prepares the tile path
for rep in range(3):
    cmap = pb.Rotation(*rot).rotate_coordinate_map(dst.get_coordinate_map())
    out = pb.PanoramaImage(frames[0]).process_coordinate_map(cmap)
    assert isinstance(out, np.ndarray) and out.dtype == np.uint8 and np.array_equal(out, want[0]), rep
# 2. a capture buffer that is refilled: seen twice -> page-locked in place -> one DMA out of the caller's memory; same bytes
buf = np.empty_like(frames[0])
for k in range(5):
    buf[...] = frames[k]
    cmap = pb.Rotation(*rot).rotate_coordinate_map(dst.get_coordinate_map())
    out = pb.PanoramaImage(buf).process_coordinate_map(cmap)
    assert np.array_equal(out, want[k]), k
    keep = out  # (results are recycled page-locked blocks: holding one must not let the next call overwrite it)
    out2 = pb.PanoramaImage(buf).process_coordinate_map(cmap)
    assert np.array_equal(keep, want[k]) and np.array_equal(out2, want[k]) and out2.ctypes.data != keep.ctypes.data
# 3. streaming: results in order, equal to the single calls
plan = batch.plan_for(dst, [pb.Rotation(*rot)], pb.PanoramaImage(frames[0]))
outs = list(batch.remap_frames(plan, (f for f in frames), depth=2))
assert len(outs) == 5 and all(np.array_equal(o, w) for o, w in zip(outs, want))
# 4. the paths beside the fused one: a materialised map, a grey 16-bit image, the bilinear mode
m = np.asarray(pb.Rotation(*rot).rotate_coordinate_map(dst.get_coordinate_map()))
assert m.shape == (h, h, 3) and m.dtype == np.float64
assert np.array_equal(pb.PanoramaImage(frames[1]).process_coordinate_map(m.copy()), want[1])
grey = synth_image(h, 2 * h, "I;16")
cmap = pb.Rotation(*rot).rotate_coordinate_map(dst.get_coordinate_map())
g = pb.PanoramaImage(grey).process_coordinate_map(cmap)
idx = orc.remap_index(od, os_, [rot])
assert g.dtype == grey.dtype and np.array_equal(g, np.where(idx >= 0, grey.reshape(-1)[np.maximum(idx, 0)], 0).astype(grey.dtype))
bl = pb.PanoramaImage(frames[2]).process_coordinate_map(pb.Rotation(*rot).rotate_coordinate_map(dst.get_coordinate_map()), interpolation="bilinear")
assert bl.shape == (h, h, 3) and not np.array_equal(bl, want[2])
print("OK")
"""


def _run(script, timeout=600):
    return subprocess.run([sys.executable, "-c", script], cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_package_imports_and_plans_without_torch():
    res = _run(_CPU)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr


@pytest.mark.gpu
def test_numpy_workflow_runs_without_torch_on_the_gpu():
    res = _run(_GPU)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr
