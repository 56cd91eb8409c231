"""us/frame of BASELINE configs with one build of the library, frames rotating through a pool larger than the Infinity Cache:
    python experiments/ab_case.py <lib.so | -> <config>[:<batch>][@<budget>] ...
'-' = the in-tree product library.  One line per (config, batch, budget): median / mean of 5 x 40 launches, tile classes."""
import os, sys, json
lib = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if lib != '-':
    os.environ['PB_LIB_PATH'] = os.path.abspath(lib)
import numpy as np, torch
import bench
from photonbend_amd import _native as nat
L = nat.load()
REMAP = L.pb_remap_bilinear_u8 if os.environ.get('PB_AB_BILINEAR') == '1' else L.pb_remap_u8  # PB_AB_BILINEAR=1: the opt-in bilinear mode
dev = torch.device('cuda', 0)
for spec in sys.argv[2:]:
    budget = 0
    if '@' in spec: spec, b = spec.split('@'); budget = int(b)
    batch = 1
    if ':' in spec: spec, b = spec.split(':'); batch = int(b)
    cfg = bench.CONFIGS[spec]
    d, rots, s = bench.build_projs(cfg)
    plan = nat.Plan(d, rots, s, budget=budget)
    sb, db = 3 * s.height * s.width, 3 * d.height * d.width
    pool = max(2 * batch, int((int(os.environ.get('PB_POOL_MB', '1280')) << 20) // (sb + db)) + 1)  # PB_POOL_MB: bytes of frames rotated (default 1280 MB = 5 x the 256 MB Infinity Cache; 320 MB left c1 / c3 partly cached)
    pool = (pool + batch - 1) // batch * batch
    srcs = torch.empty((pool, s.height, s.width, 3), dtype=torch.uint8, device=dev)
    for f in range(pool): nat.synth_frame(s.height, s.width, frame=f, seed=0, circle_mask=cfg['mask'], out=srcs[f])
    dsts = torch.empty((pool, d.height, d.width, 3), dtype=torch.uint8, device=dev)
    st = nat.current_stream()
    def step(k):
        i = (k % (pool // batch)) * batch
        rc = REMAP(plan.handle, srcs.data_ptr() + i * sb, dsts.data_ptr() + i * db, batch, sb, db, st)
        if rc: nat.check(rc)
    for k in range(20): step(k)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(40): step(k + rep)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 40 / batch)
    i = plan.info()
    print('%-22s %-8s pool %3d batch %2d budget %5d : median %7.2f us/frame  (min %7.2f max %7.2f)  lean %5d direct %5d packed %5d black %5d fail %4d' % (
        os.path.basename(lib), spec, pool, batch, i['window_budget'], float(np.median(ts)), min(ts), max(ts), i['lean_tiles'], i['direct_tiles'], i.get('packed_tiles', 0), i['black_tiles'], i['fix_tiles']), flush=True)
    del srcs, dsts, plan
    torch.cuda.empty_cache()
