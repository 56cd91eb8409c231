"""One-off: the per-pixel DEFINITION kernel of the bilinear mode (a materialised map through the facade) against oracle.remap_bilinear on N random
geometries (tests/test_hip_random.random_case): equal bytes expected where the live NumPy is the goldens' NumPy, 1 LSB otherwise.
usage: fuzz_definition.py [N] [seed0]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import photonbend_amd as pb
from oracle import reference_path as orc
from oracle.synth import synth_frame
from tests import helpers as H
from tests.test_hip_random import random_case
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 880000
exact = H.live_numpy_is_the_goldens_numpy()
bad = diff_total = 0
t0 = time.time()
for k in range(N):
    case = random_case(np.random.default_rng(seed0 + k), k)
    try:
        frame = synth_frame(case.src[1], case.src[2], frame=k % 5, seed=1)
        cmap = H.pb_obj(case.dst).get_coordinate_map()
        for rot in case.rotations:
            cmap = pb.Rotation(*map(pb.utils.to_radians, rot)).rotate_coordinate_map(cmap)
        host_map = np.array(np.asarray(cmap))
        with np.errstate(all="ignore"):
            want = orc.remap_bilinear(H.orc_proj(case.dst), H.orc_proj(case.src), frame, H.orc_rots(case))
        got = H.pb_obj(case.src, frame).process_coordinate_map(host_map, interpolation="bilinear")
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        if case.src[0] == "double":
            d = np.minimum(d, 256 - d)
        nd = int((d != 0).sum()); diff_total += nd
        if int(d.max(initial=0)) > 1 or (exact and nd):
            bad += 1
            print(f"BAD {case.name} {case.dst} <- {case.src} rots {len(case.rotations)}: differing {nd}, max {int(d.max())}", flush=True)
    except Exception as ex:
        bad += 1
        print(f"EXC {case.name} {case.dst} <- {case.src}: {type(ex).__name__} {str(ex)[:160]}", flush=True)
    if k % 50 == 49: print(f"... {k + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("done", N, "cases,", bad, "bad; exact expected:", exact, "; samples differing in total:", diff_total)
