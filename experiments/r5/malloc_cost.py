"""What does hipMalloc / hipFree cost on this box?  (a plan makes 12-18 allocations)"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photonbend_amd import _native as nat
lib = nat.load()
nat.require_gpu()
for size in (4096, 65536, 1 << 20, 4 << 20, 16 << 20):
    ps = []
    p = C.c_void_p()
    for _ in range(3):
        lib.pb_malloc(C.byref(p), C.c_size_t(size)); lib.pb_free(p)
    t0 = time.perf_counter()
    for _ in range(20):
        q = C.c_void_p(); lib.pb_malloc(C.byref(q), C.c_size_t(size)); ps.append(q)
    t1 = time.perf_counter()
    for q in ps: lib.pb_free(q)
    t2 = time.perf_counter()
    print(f"{size:>9} B: malloc {(t1 - t0) / 20 * 1e6:6.1f} us   free {(t2 - t1) / 20 * 1e6:6.1f} us")
