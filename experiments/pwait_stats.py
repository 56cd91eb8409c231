"""How much slack do partial waits have?  LEAN tiles of a config: loads still allowed in flight at each band (of T load instructions)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ['PB_LIB_PATH'] = os.path.abspath('experiments/libpb_pwait_trace.so')
import bench
from photonbend_amd import _native as nat
lib = nat.load()
for name in sys.argv[1:]:
    cfg = bench.CONFIGS[name]; d, rots, s = bench.build_projs(cfg)
    plan = nat.Plan(d, rots, s); nt = plan.info()['tiles']
    tab = (ctypes.c_int32 * (nt * 64))()
    assert lib.pb_debug_copy_table(plan.handle, tab, nt * 256) == 0
    E = np.frombuffer(tab, dtype=np.int32).reshape(nt, 64)
    flags, rows, n16, wb = E[:, 2], E[:, 3], E[:, 57], E[:, 63]
    lean = (flags & 4) != 0
    rpp = 64 // np.maximum(n16, 1); T = (rows + rpp - 1) // rpp
    N = np.stack([(wb >> (6 * i)) & 63 for i in range(4)], axis=1)
    print(name, 'LEAN tiles', int(lean.sum()), 'load instructions per window: mean %.1f' % T[lean].mean(), 'bottom-up %.0f %%' % (100 * ((wb[lean] >> 24) & 1).mean()),
          'in flight at band 0..3: mean', np.round(N[lean].mean(axis=0), 2), 'as a fraction of T', np.round((N[lean] / T[lean, None]).mean(axis=0), 2))
