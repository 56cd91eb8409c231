"""Create / destroy plans in a loop and watch free device memory (plans own tables, tuning allocates scratch)."""
import sys, gc, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from photonbend_amd.core.projection import _PLAN_CACHE
from tests import helpers as H
from tests.cases import Case, cam, dbl, pano, inscribed
cases = [
    Case("a", cam(2048, 2048, "equidistant", 360, inscribed(2048)), pano(2048, 4096), [(10, 20, 30)]),
    Case("b", pano(2048, 4096), dbl(1920, 3840, "equidistant", 195), [(3, 90, -7)]),
    Case("c", cam(2048, 2048, "equisolid", 360, inscribed(2048)), cam(2048, 2048, "equidistant", 360, inscribed(2048)), [(30, 45, 10)]),
]
torch.cuda.init()
free0 = None
for it in range(40):
    for c in cases:
        _PLAN_CACHE.clear()
        plan = H.pb_plan(c)
        del plan
    _PLAN_CACHE.clear(); gc.collect(); torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 2: free0 = free
    if it % 10 == 9: print('iteration', it + 1, 'free MiB', free >> 20)
print('drift since iteration 3: %d KiB' % ((free0 - free) >> 10))
