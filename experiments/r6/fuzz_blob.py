"""One-off: plan blobs on random geometries - serialize, deserialize, same bytes from both samplers, same tile mix.  usage: fuzz_blob.py [N] [seed0]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.test_hip_random import random_case
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 770000
bad = 0
for k in range(N):
    case = random_case(np.random.default_rng(seed0 + k), k)
    try:
        src, cmap = H.pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
        d, rots, s = cmap.dst_proj, cmap.rotations, src._proj()
        frame = nat.synth_frame(case.src[1], case.src[2], frame=k % 7)
        plan = nat.Plan(d, rots, s, bilinear=bool(k & 1))
        if not plan.info()["fast_path"]:
            continue
        a, b = plan.remap(frame), plan.remap(frame, interpolation="bilinear")
        twin = nat.Plan.deserialize(plan.serialize(), d, rots, s)
        ok = torch.equal(twin.remap(frame), a) and torch.equal(twin.remap(frame, interpolation="bilinear"), b) and twin.bilinear_tile_mix() == plan.bilinear_tile_mix() \
            and twin.info() == plan.info()
        if not ok:
            bad += 1
            print(f"BAD {case.name} {case.dst} <- {case.src} rots {len(case.rotations)}", flush=True)
        del plan, twin
    except Exception as ex:
        bad += 1
        print(f"EXC {case.name} {case.dst} <- {case.src}: {type(ex).__name__} {str(ex)[:200]}", flush=True)
print("done", N, "cases,", bad, "bad")
