import sys, os, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
plan = H.pb_plan(case)
plan.set_mode(nat.MODE_FAST)
for dbg in (0, 7, 15, 14):
    os.environ['PB_DEBUG'] = str(dbg)
    for i in range(2): plan.index_map()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10): idx = plan.index_map()
    e1.record(); torch.cuda.synchronize()
    print('debug mask %d: index-map %.1f us' % (dbg, e0.elapsed_time(e1) * 1e3 / 10))
