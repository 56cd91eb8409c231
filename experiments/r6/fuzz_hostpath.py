"""One-off: the NumPy-in / NumPy-out path on random geometries - batch.remap_frames over a list and over an iterator (fresh arrays, a refilled
buffer, views), depth 2-4, both samplers, and _hostpipe.remap_ndarray - against the device results.  usage: fuzz_hostpath.py [N] [seed0]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from photonbend_amd import _native as nat, _hostpipe, batch
from tests import helpers as H
from tests.test_hip_random import random_case
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 550000
bad = 0
for k in range(N):
    rng = np.random.default_rng(seed0 + k)
    c = random_case(rng, k)
    f = int(rng.integers(1, 9))
    up = lambda p: (p[0], p[1] * f, p[2] * f, p[3], p[4], None if p[5] is None else p[5] * f)
    case = type(c)(f"fh{k}", up(c.dst), up(c.src), c.rotations, c.mask)
    try:
        plan = H.pb_plan_private(case)
        nf = int(rng.integers(1, 7))
        dev = [nat.synth_frame(case.src[1], case.src[2], frame=i + 3 * k) for i in range(nf)]
        host = [d.cpu().numpy() for d in dev]
        mode = "bilinear" if k % 3 == 0 else "nearest"
        want = [plan.remap(d, interpolation=mode).cpu().numpy() for d in dev]
        depth = int(rng.integers(2, 5))
        kind = k % 4
        if kind == 0:
            got = list(batch.remap_frames(plan, host, depth=depth, interpolation=mode))
        elif kind == 1:
            got = list(batch.remap_frames(plan, iter(host), depth=depth, interpolation=mode))
        elif kind == 2:  # one buffer the producer refills
            buf = np.empty_like(host[0])
            def gen():
                for h in host:
                    buf[...] = h
                    yield buf
            got = [g.copy() for g in batch.remap_frames(plan, gen(), depth=depth, interpolation=mode)]
        else:  # views of a bigger array (never page-locked in place: the staged copy)
            big = np.zeros((nf, case.src[1] + 2, case.src[2], 3), np.uint8)
            for i, h in enumerate(host): big[i, 1:-1] = h
            got = list(batch.remap_frames(plan, [big[i, 1:-1] for i in range(nf)], depth=depth, interpolation=mode))
        ok = len(got) == nf and all(np.array_equal(a, b) for a, b in zip(got, want))
        ok = ok and np.array_equal(_hostpipe.remap_ndarray(plan, host[0], interpolation=mode), want[0])
        if not ok:
            bad += 1
            print(f"BAD {case.name} kind {kind} mode {mode} depth {depth} frames {nf} {case.dst} <- {case.src}", flush=True)
    except Exception as ex:
        bad += 1
        print(f"EXC {case.name} {case.dst} <- {case.src}: {type(ex).__name__} {str(ex)[:200]}", flush=True)
print("done", N, "cases,", bad, "bad")
