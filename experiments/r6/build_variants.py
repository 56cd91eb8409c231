import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sys
from concurrent.futures import ThreadPoolExecutor
from photonbend_amd.build import build_diagnostic, build_library, build_libm_flavour
vs = [int(a) for a in sys.argv[1:]]
with ThreadPoolExecutor(7) as p:
    jobs = [p.submit(build_library, force=True), p.submit(build_diagnostic, force=True)]
    jobs += [p.submit(build_library, force=True, out=f'build/libpb_r6_skip{v}.so', defines=(f'PB_R6_SKIP={v}',)) for v in vs]
    [j.result() for j in jobs]
print('built')
