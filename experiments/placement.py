"""Does the time of a config depend on WHERE its buffers lie?  One process, one plan; the frame pools are re-allocated at shifted
addresses (a junk allocation of varying size in front), then the plan is re-created with the pools kept.
    python experiments/placement.py <lib.so | -> <config> [batch]"""
import os, sys
lib = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if lib != '-':
    os.environ['PB_LIB_PATH'] = os.path.abspath(lib)
import numpy as np, torch
import bench
from photonbend_amd import _native as nat
L = nat.load()
name = sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = bench.CONFIGS[name]
d, rots, s = bench.build_projs(cfg)
sb, db = 3 * s.height * s.width, 3 * d.height * d.width
pool = max(2 * batch, int((1280 << 20) // (sb + db)) + 1)
pool = (pool + batch - 1) // batch * batch
st = nat.current_stream()

def timeit(plan, srcs, dsts, ss, ds):
    def step(k):
        i = (k % (pool // batch)) * batch
        nat.check(L.pb_remap_u8(plan.handle, srcs + i * ss, dsts + i * ds, batch, ss, ds, st))
    for k in range(20): step(k)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(40): step(k + rep)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 40 / batch)
    return float(np.median(ts))

plan = nat.Plan(d, rots, s)
frame = nat.synth_frame(s.height, s.width, frame=0, seed=0, circle_mask=cfg['mask']).reshape(-1)
big = torch.empty(pool * (sb + db) + (64 << 20), dtype=torch.uint8, device='cuda')
base = big.data_ptr()
print('big buffer at %#x' % base)
# (a) pools inside one big buffer at shifted offsets, tightly packed frames
for shift in (0, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, 17 << 20):
    so = shift; do = shift + pool * sb
    do = (do + 15) & ~15
    for f in range(pool): big[so + f * sb: so + (f + 1) * sb] = frame
    t = timeit(plan, base + so, base + do, sb, db)
    print('shift %9d  src %#x dst %#x : %6.2f us/frame' % (shift, base + so, base + do, t), flush=True)
# (b) padded strides
for pad in (0, 256, 4096, 4096 + 256, 65536, 65536 + 4096):
    ss, ds = sb + pad, db + pad
    if pool * (ss + ds) > big.numel(): break
    so, do = 0, pool * ss
    for f in range(pool): big[so + f * ss: so + f * ss + sb] = frame
    t = timeit(plan, base + so, base + do, ss, ds)
    print('pad   %9d  : %6.2f us/frame' % (pad, t), flush=True)
# (c) fresh plans, same buffers
for k in range(4):
    p2 = nat.Plan(d, rots, s)
    for f in range(pool): big[f * sb: (f + 1) * sb] = frame
    t = timeit(p2, base, base + pool * sb, sb, db)
    print('fresh plan %d : %6.2f us/frame' % (k, t), flush=True)
    junk = torch.empty((k + 1) * 1234567, dtype=torch.uint8, device='cuda')
