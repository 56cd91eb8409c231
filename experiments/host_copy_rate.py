"""Host memcpy rates on the GPU box: NumPy single-threaded vs par_copy (threads) into pageable / pinned memory."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from photonbend_amd.utils.hostcopy import par_copy
a = np.random.randint(0, 255, (4096, 8192, 3), dtype=np.uint8)
b = np.empty_like(a)
pin = torch.empty(a.shape, dtype=torch.uint8).pin_memory().numpy()
def rate(fn, n=5):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return a.nbytes * n / (time.perf_counter() - t0) / 1e9
print('numpy copy pageable->pageable %.1f GB/s' % rate(lambda: np.copyto(b, a)))
print('numpy copy pageable->pinned   %.1f GB/s' % rate(lambda: np.copyto(pin, a)))
for parts in (2, 4, 8, 16):
    print('par_copy %2d parts -> pinned    %.1f GB/s' % (parts, rate(lambda: par_copy(pin, a, parts))))
    print('par_copy %2d parts pinned -> pageable %.1f GB/s' % (parts, rate(lambda: par_copy(b, pin, parts))))
t = torch.from_numpy(a); tp = torch.from_numpy(pin)
print('torch copy_ pageable->pinned  %.1f GB/s' % rate(lambda: tp.copy_(t)))
import os; print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads())
