"""One-off: batches on random geometries - N frames at a padded byte stride in ONE launch (both samplers) and a ring of separately allocated frames
(pb_remap_u8v) against one launch per frame.  usage: fuzz_batches.py [N] [seed0]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.test_hip_random import random_case
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 660000
bad = 0
for k in range(N):
    rng = np.random.default_rng(seed0 + k)
    c = random_case(rng, k)
    f = int(rng.integers(1, 6))
    up = lambda p: (p[0], p[1] * f, p[2] * f, p[3], p[4], None if p[5] is None else p[5] * f)
    case = type(c)(f"fb{k}", up(c.dst), up(c.src), c.rotations, c.mask)
    try:
        plan = H.pb_plan_private(case)
        nf = int(rng.integers(2, 6))
        sb, db = 3 * case.src[1] * case.src[2], 3 * case.dst[1] * case.dst[2]
        g = 1 if os.environ.get("ODD") == "1" else 16  # (ODD=1: any byte stride - frames the windows' 16-byte LDS-DMA cannot address take the direct-gather kernels)
        ss, ds = sb + g * int(rng.integers(0, 9)), db + g * int(rng.integers(0, 9))
        src = torch.zeros(nf * ss, dtype=torch.uint8, device="cuda")
        frames = []
        for i in range(nf):
            fr = nat.synth_frame(case.src[1], case.src[2], frame=i + k)
            src[i * ss : i * ss + sb] = fr.reshape(-1)
            frames.append(fr)
        for mode in ("nearest", "bilinear"):
            singles = [plan.remap(fr, interpolation=mode) for fr in frames]
            dst = torch.full((nf * ds,), 7, dtype=torch.uint8, device="cuda")
            plan.launch(src.data_ptr(), dst.data_ptr(), nf, None, mode, ss, ds)
            torch.cuda.synchronize()
            ok = all(torch.equal(dst[i * ds : i * ds + db].reshape(singles[i].shape), singles[i]) for i in range(nf))
            ok = ok and all(bool((dst[i * ds + db : (i + 1) * ds] == 7).all()) for i in range(nf))  # the padding is untouched
            if not ok:
                bad += 1
                print(f"BAD {mode} batch {case.name} {case.dst} <- {case.src} frames {nf} strides +{ss - sb} +{ds - db}", flush=True)
        ring = plan.remap_each(frames)
        if not all(torch.equal(a, b) for a, b in zip(ring, [plan.remap(fr) for fr in frames])):
            bad += 1
            print(f"BAD ring {case.name}", flush=True)
    except Exception as ex:
        bad += 1
        print(f"EXC {case.name} {case.dst} <- {case.src}: {type(ex).__name__} {str(ex)[:200]}", flush=True)
print("done", N, "cases,", bad, "bad")
