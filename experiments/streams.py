"""Does the time of a config depend on the STREAM (hardware queue) of one process?  python experiments/streams.py <config>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from photonbend_amd import _native as nat
L = nat.load()
cfg = bench.CONFIGS[sys.argv[1]]
d, rots, s = bench.build_projs(cfg)
sb, db = 3 * s.height * s.width, 3 * d.height * d.width
pool = int((1280 << 20) // (sb + db)) + 1
plan = nat.Plan(d, rots, s)
srcs = torch.empty((pool, s.height, s.width, 3), dtype=torch.uint8, device='cuda')
for f in range(pool): nat.synth_frame(s.height, s.width, frame=f, seed=0, circle_mask=cfg['mask'], out=srcs[f])
dsts = torch.empty((pool, d.height, d.width, 3), dtype=torch.uint8, device='cuda')
torch.cuda.synchronize()
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(5)]
for rnd in range(2):
    for si, stream in enumerate(streams):
        st = int(stream.cuda_stream)
        def step(k):
            i = k % pool
            nat.check(L.pb_remap_u8(plan.handle, srcs.data_ptr() + i * sb, dsts.data_ptr() + i * db, 1, sb, db, st))
        for k in range(20): step(k)
        stream.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for k in range(40): step(k + rep)
            e1.record(stream); stream.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 40)
        print('round %d stream %d (%#x): %6.2f us/frame' % (rnd, si, st, float(np.median(ts))), flush=True)
