"""One-off: 3-8 chained rotations on random geometries - the reference's sampler against the live oracle, the bilinear tiles against the definition
kernel.  usage: fuzz_rotations.py [N] [seed0]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import reference_path as orc
from oracle.synth import synth_frame
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import Case
from tests.test_hip_random import random_case
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 440000
bad = 0
for k in range(N):
    rng = np.random.default_rng(seed0 + k)
    c = random_case(rng, k)
    nrot = int(rng.integers(3, 9))
    rots = [tuple(float(v) for v in rng.uniform(-180, 180, 3)) for _ in range(nrot)]
    case = Case(f"fr{k}", c.dst, c.src, rots, c.mask)
    try:
        frame = synth_frame(case.src[1], case.src[2], frame=2)
        with np.errstate(all="ignore"):
            want = orc.remap(H.orc_proj(case.dst), H.orc_proj(case.src), frame, H.orc_rots(case))
        plan = H.pb_plan_private(case)
        dev = torch.from_numpy(frame).cuda()
        got = plan.remap(dev).cpu().numpy()
        src, cmap = H.pb_chain(case, dev)
        wb = nat.sample_map_bilinear(src._proj("src"), cmap.device_tensor(), dev, 3, np.uint8).reshape(case.dst[1], case.dst[2], 3).to(torch.int16)
        gb = plan.remap(dev, interpolation="bilinear").to(torch.int16)
        d = (gb - wb).abs(); d = torch.minimum(d, 256 - d).amax(dim=2)
        lim = 2 if case.src[0] == "double" else 1
        if not np.array_equal(got, want) or int((d > lim).sum()):
            bad += 1
            print(f"BAD {case.name} {case.dst} <- {case.src} rots {nrot}: nearest differing bytes {int((got != want).sum())}, bilinear beyond {lim}: {int((d > lim).sum())} max {int(d.max())}", flush=True)
    except Exception as ex:
        bad += 1
        print(f"EXC {case.name} {case.dst} <- {case.src}: {type(ex).__name__} {str(ex)[:200]}", flush=True)
print("done", N, "cases,", bad, "bad")
