"""Where a double-fisheye source's tile kernels land 2 LSB from the definition: for one fuzz case (seed, index as fuzz_bilinear.py) prints the
pixels beyond 1 LSB with their source coordinates, merge factors and both outputs.   python experiments/r6/two_lsb_probe.py SEED0 K [scale]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import Case
from tests.test_hip_random import random_case
seed0, k = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0 + k)
c = random_case(rng, k)
f = int(rng.integers(int(os.environ.get("FZ_LO", "1")), int(os.environ.get("FZ_HI", "7"))))
up = lambda p: (p[0], p[1] * f, p[2] * f, p[3], p[4], None if p[5] is None else p[5] * f)
case = Case(f"fz{k}", up(c.dst), up(c.src), c.rotations, c.mask)
print(case)
frame = nat.synth_frame(case.src[1], case.src[2], frame=k % 7)
plan = H.pb_plan_private(case)
got = plan.remap(frame, interpolation="bilinear").to(torch.int16)
src, cmap = H.pb_chain(case, frame)
dm = cmap.device_tensor()
want = nat.sample_map_bilinear(src._proj("src"), dm, frame, 3, np.uint8).reshape(case.dst[1], case.dst[2], 3).to(torch.int16)
plan.set_mode(nat.MODE_FAITHFUL)
f64 = plan.remap(frame, interpolation="bilinear").to(torch.int16)
d = (got - want).abs(); d = torch.minimum(d, 256 - d)
d64 = (f64 - want).abs(); d64 = torch.minimum(d64, 256 - d64)
print("tile kernels: beyond 1:", int((d.amax(dim=2) > 1).sum()), " float64-mode kernel: beyond 1:", int((d64.amax(dim=2) > 1).sum()), "beyond 0:", int((d64.amax(dim=2) > 0).sum()), "of", d.shape[0] * d.shape[1])
ys, xs = torch.nonzero(d.amax(dim=2) > 1, as_tuple=True)
m = dm.reshape(case.dst[1], case.dst[2], 3).cpu().numpy()
mix = plan.bilinear_tile_mix(); print(mix)
P = plan.params() if hasattr(plan, "params") else None
for y, x in list(zip(ys.tolist(), xs.tolist()))[:40]:
    print(f"  px ({y},{x}) tile ({y // 32},{x // 32}) lat {m[y, x, 0]:.6f} lon {m[y, x, 1]:.6f}  got {got[y, x].tolist()} want {want[y, x].tolist()} f64 {f64[y, x].tolist()}")

# ---- the tile models at the bad pixels: full model against its degree <= 3 part (the plan's own tables, from the serialised blob) ----
import struct
blob = plan.serialize()
magic, version, params_size, entry_size = struct.unpack_from("<4I", blob, 0)
sec = struct.unpack_from("<13Q", blob, 16 + 16 + 32 + 8)
hdr = 16 + 16 + 32 + 8 + 104 + 8
nt = struct.unpack_from("<I", blob, 32)[0]
base = hdr + params_size
tabs = [np.frombuffer(blob, dtype=np.uint8, count=sec[0], offset=base).reshape(nt, 256), np.frombuffer(blob, dtype=np.uint8, count=sec[1], offset=base + sec[0]).reshape(nt, 256)]
tiles_x = (case.dst[2] + 31) // 32

def entry(tab, t):
    raw = tab[t].tobytes()
    ar, ac, fl, wr = struct.unpack_from("<4i", raw, 0)
    c = np.frombuffer(raw, dtype=np.float32, count=50, offset=16).reshape(25, 2).astype(np.float64)
    tail = struct.unpack_from("<10i", raw, 216)
    return ar, ac, fl, c, tail

def model(c, u, v, td3):
    out = np.zeros(2)
    for mm in range(5):
        for nn in range(5):
            if td3 and mm + nn > 3:
                continue
            out += c[mm * 5 + nn] * (v ** mm) * (u ** nn)
    return out

for y, x in list(zip(ys.tolist(), xs.tolist()))[:40]:
    t = (y // 32) * tiles_x + x // 32
    u, v = ((x % 32) - 15.5) / 15.5, ((y % 32) - 15.5) / 15.5
    for eye, tab in enumerate(tabs):
        ar, ac, fl, c, tail = entry(tab, t)
        if fl & 8:
            continue
        full, part = model(c, u, v, False), model(c, u, v, True)
        print(f"  px ({y},{x}) eye {eye} flags {fl:#x} td3 {bool(fl & 8192)} coarse {bool(fl & 4096)} lean {bool(fl & 4)} direct {bool(fl & 16)} bil_off {tail[-1]}  full ({ar + full[0]:.5f},{ac + full[1]:.5f})  td3-full ({(part - full)[0] * 1024:.3f},{(part - full)[1] * 1024:.3f}) /1024 px")
