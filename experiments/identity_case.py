"""The degenerate 'identity' remap (same lens, fov, size, no rotation): every pre-truncation coordinate is an
integer +- 1e-13, so the truncated index follows the last bit of cos/sin/atan2 - the reference's own output
is decided by libm rounding noise.  How far apart are the device and NumPy there?"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import reference_path as orc
from oracle.synth import synth_frame
from tests import helpers as H
from tests.cases import Case, cam, inscribed
for lens in ('equidistant', 'equisolid', 'stereographic'):
    case = Case('id', cam(200, 200, lens, 180, inscribed(200)), cam(200, 200, lens, 180, inscribed(200)))
    od, os_ = H.orc_proj(case.dst), H.orc_proj(case.src)
    with np.errstate(all='ignore'):
        want = orc.remap_index(od, os_)
        fragile = orc.fragile_mask(orc.pretrunc(od, os_))
    got = H.pb_plan(case).index_map().cpu().numpy()
    bad = got != want
    ident = np.arange(200 * 200).reshape(200, 200)
    print('%-14s fragile pixels %5d of 40000; device != NumPy at %5d (outside fragile set: %d); NumPy == identity at %5d, device == identity at %5d'
          % (lens, fragile.sum(), bad.sum(), (bad & ~fragile).sum(), (want == ident).sum(), (got == ident).sum()))
