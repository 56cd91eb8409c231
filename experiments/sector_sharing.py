"""How many 128-byte source lines (or 64-byte sectors: second argument) do the hot kernel's work units share?  (CPU, oracle index map.)
    python experiments/sector_sharing.py c2
must-move = sectors sampled at least once; per-tile / per-workgroup / per-super-tile / per-XCD sums say what HBM would
see if NOTHING were shared beyond that unit (the measured fetch lies between the per-XCD sum and the per-tile sum)."""
import sys, numpy as np
G = int(sys.argv[2]) if len(sys.argv) > 2 else 128  # bytes per unit of fetch: 128 (a line: what an L2 miss moves) or 64 (a sector)
sys.path.insert(0, '.')
from oracle import reference_path as orc
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
idx = orc.remap_index(H.orc_proj(case.dst), H.orc_proj(case.src), H.orc_rots(case))
idx = np.asarray(idx).reshape(case.dst[1], case.dst[2])
Hh, W = idx.shape
ok = idx >= 0
b0 = np.where(ok, idx.astype(np.int64) * 3, -1)
def count(unit_id, n_units):
    # unique (unit, sector) pairs over first and last byte of every sample
    tot = 0
    for off in (0, 2):
        pass
    s0 = np.where(ok, b0 // G, -1); s1 = np.where(ok, (b0 + 2) // G, -1)
    key = np.concatenate([(unit_id.astype(np.int64) << 32 | s0)[ok], (unit_id.astype(np.int64) << 32 | s1)[ok]])
    return np.unique(key).size
yy, xx = np.mgrid[0:Hh, 0:W]
one = np.zeros_like(yy)
must = count(one, 1)
print('%s: must-move fetch %.1f MB' % (case.name, must * G / 1e6))
tile = (yy // 32) * ((W + 31) // 32) + xx // 32
wg = (yy // 64) * ((W + 63) // 64) + xx // 64
st = (yy // 256) * ((W + 255) // 256) + xx // 256
xcd = (xx // 256) & 7
for name, u in (('tile 32x32', tile), ('workgroup 64x64', wg), ('super-tile 256x256', st), ('XCD (column x, x+8)', xcd), ('XCD x row of super-tiles', (yy // 256) * 8 + xcd)):
    n = count(u, 0)
    print('  %-28s %.1f MB (%.2fx)' % (name, n * G / 1e6, n / must))
