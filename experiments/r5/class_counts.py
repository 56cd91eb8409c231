"""tile classes of the BASELINE geometries under the bilinear mode's window budget (12 KiB) and the nearest mode's (7 KiB)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from photonbend_amd import _native as nat
for name in (sys.argv[1:] or ['c1', 'c2', 'c3', 'c5']):
    d, rots, s = bench.build_projs(bench.CONFIGS[name])
    plan = nat.Plan(d, rots, s)
    for b in (7168, 12288):
        plan.set_window_budget(b)
        i = plan.info()
        print(name, b, {k: i[k] for k in ('tiles', 'lean_tiles', 'direct_tiles', 'black_tiles', 'fix_tiles', 'fix_pixels')}, flush=True)
    print(name, 'bilinear mix', plan.bilinear_tile_mix(), plan.bilinear_launch_shape(), flush=True)
