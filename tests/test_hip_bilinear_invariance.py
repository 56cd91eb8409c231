"""The SPEED-ONLY features of the opt-in bilinear mode must not move a bit (round 5): half windows (PB_TILE_HALVES), unguarded table
tiles (PB_TILE_TAB_PLAIN), the table tiles' walk order, half windows in the pair slots of a stitch and the small LDS pool decide HOW a
tile's taps reach the lanes, never which taps or weights.  The diagnostic build (-DPB_ABLATION, loaded through PB_LIB_PATH; the product
reads no environment) switches them off at plan creation with PB_BIL_OFF; two child processes - everything on, everything off - remap
the same noise frames at the benchmark geometries and at mid-size ones, and every output must have the same SHA-256.  The run with the
features on must actually use them (tile mix), the run with them off must not."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import hashlib, json, sys
import numpy as np
import torch
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import Case, pano, cam, dbl, full_cases, mid_cases
assert nat.LIB_PATH.endswith("libphotonbend_hip_diag.so"), nat.LIB_PATH
cases = list(full_cases()) + list(mid_cases()) + [
    Case("mid_rim", cam(1024, 1024, "equidistant", 360, 511.5), pano(1024, 2048)),
    Case("mid_rot", cam(1024, 1024, "equisolid", 360, 511.5), cam(1024, 1024, "equidistant", 360, 511.5), rotations=[(30.0, 45.0, 10.0)]),
    Case("mid_dbl", pano(1024, 2048), dbl(972, 1944, "equidistant", 190), mask=2),
    Case("mid_pole", pano(512, 1024), pano(1536, 3072), rotations=[(90.0, 0.0, 0.0)]),
]
out = {}
for case in cases:
    plan = H.pb_plan_private(case)
    frame = nat.synth_frame(case.src[1], case.src[2], frame=3, seed=11, circle_mask=case.mask)
    got = plan.remap(frame, interpolation="bilinear")
    out[case.name] = {"sha256": hashlib.sha256(got.cpu().numpy().tobytes()).hexdigest(), "mix": plan.bilinear_tile_mix(),
                      "float64_tiles": plan.info()["bilinear_float64_tiles"], "shape": plan.bilinear_launch_shape()}
json.dump(out, open(sys.argv[1], "w"))
"""


def _run(tmp_path, tag, off):
    from photonbend_amd.build import DIAG_LIB_PATH

    res_path = str(tmp_path / f"{tag}.json")
    env = dict(os.environ, PB_LIB_PATH=DIAG_LIB_PATH, PB_BIL_OFF=str(off))
    res = subprocess.run([sys.executable, "-c", _WORKER, res_path], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    return json.load(open(res_path))


@pytest.mark.gpu
def test_speed_only_features_do_not_move_a_bit(tmp_path):
    from photonbend_amd.build import DIAG_LIB_PATH

    if not os.path.exists(DIAG_LIB_PATH):
        pytest.skip("needs the diagnostic build (python -m photonbend_amd.build --diag)")
    on, off = _run(tmp_path, "on", 0), _run(tmp_path, "off", 1 | 2 | 4 | 8 | 16)
    assert set(on) == set(off) and len(on) >= 9
    bad = [name for name in on if on[name]["sha256"] != off[name]["sha256"]]
    assert not bad, f"outputs depend on speed-only features: {bad}"
    assert all(r["float64_tiles"] == 0 for r in on.values())
    # the features were really exercised in one run and really absent in the other
    assert sum(r["mix"]["half_windows"] for r in on.values()) > 1000 and sum(r["mix"]["table_plain"] for r in on.values()) > 1000
    assert sum(r["mix"]["half_windows"] for r in off.values()) == 0 and sum(r["mix"]["table_plain"] for r in off.values()) == 0
    assert any(r["shape"]["lds_bytes"] in (40448, 20224, 23392) for r in on.values()) and all(r["shape"]["lds_bytes"] not in (40448, 20224, 23392) for r in off.values())  # (the small LDS pool)
    for name in ("c2", "c3"):
        assert on[name]["mix"]["half_windows"] > 0 and on[name]["mix"]["window"] > off[name]["mix"]["window"], (name, on[name]["mix"], off[name]["mix"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c3", "mid_dbl", "mid_pole"])
def test_frames_lds_dma_cannot_address_give_the_same_pixels(name):
    """A source frame that is not 16-byte aligned cannot be staged by LDS-DMA: window and half-window tiles then take the direct-gather
    path (same taps, same weights).  Same bytes as from an aligned frame - single frames and a batch whose frame stride is odd - and as
    from the plan in MODE_FAST_DIRECT (no windows at all)."""
    import torch

    from photonbend_amd import _native as nat
    from tests import helpers as H
    from tests.cases import Case, dbl, full_cases, pano

    cases = {c.name: c for c in full_cases()}
    cases["mid_dbl"] = Case("mid_dbl", pano(1024, 2048), dbl(972, 1944, "equidistant", 190), mask=2)
    cases["mid_pole"] = Case("mid_pole", pano(512, 1024), pano(1536, 3072), rotations=[(90.0, 0.0, 0.0)])
    case = cases[name]
    plan = H.pb_plan_private(case)
    _, h, w, *_ = case.src
    n = h * w * 3
    frames = [nat.synth_frame(h, w, frame=f, seed=5, circle_mask=case.mask) for f in range(2)]
    want = [plan.remap(f, interpolation="bilinear") for f in frames]
    # one byte into a larger buffer: the pointer is odd
    buf = torch.empty(2 * n + 64, dtype=torch.uint8, device="cuda")
    for k, f in enumerate(frames):
        odd = buf[1:1 + n].view(h, w, 3)
        odd.copy_(f)
        assert odd.data_ptr() % 16 != 0
        got = plan.remap(odd, interpolation="bilinear")
        assert torch.equal(got, want[k]), f"frame {k}: an unaligned frame changes {int((got != want[k]).any(dim=2).sum())} pixels"
    # a batch of two frames at an odd stride (the second frame is unaligned): through the C ABI
    stride = n + 3
    both = torch.empty(2 * stride, dtype=torch.uint8, device="cuda")
    for k, f in enumerate(frames):
        both[k * stride:k * stride + n].copy_(f.reshape(-1))
    out = torch.empty((2, case.dst[1], case.dst[2], 3), dtype=torch.uint8, device="cuda")
    nat.check(nat.load().pb_remap_bilinear_u8(plan._h, both.data_ptr(), out.data_ptr(), 2, stride, 0, nat.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])
    plan.set_mode(nat.MODE_FAST_DIRECT)
    got = plan.remap(frames[0], interpolation="bilinear")
    assert torch.equal(got, want[0]), f"MODE_FAST_DIRECT changes {int((got != want[0]).any(dim=2).sum())} pixels"
