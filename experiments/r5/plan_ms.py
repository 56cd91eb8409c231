import sys, time
sys.path.insert(0,'/root/repo')
import bench, math
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
for case in full_cases():
    ts=[]
    for k in range(5):
        t0=time.perf_counter()
        plan=H.pb_plan_private(case)
        ts.append((time.perf_counter()-t0)*1e3)
        tm=plan.timing()
        del plan
    print(case.name, 'wall ms', [round(t,2) for t in ts], 'prepare_ms', round(tm['prepare_ms'],3), flush=True)
