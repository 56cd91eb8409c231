"""How much of a tile model's 5 x 5 coefficient table is needed?  For the BASELINE geometries (plain float64 formulas of the path, no
quirks: statistics only) the per-tile monomial coefficients from 25 Chebyshev-Lobatto nodes, and the fraction of tiles whose terms beyond
total degree 2 / 3 (and beyond tensor degree 2 / 3) sum to less than 2^-12 px."""
import numpy as np
A = np.array([[0,0,1,0,0],[0.5,-2**.5,0,2**.5,-0.5],[-0.5,2,-3,2,-0.5],[-1,2**.5,0,-2**.5,1],[1,-2,2,-2,1]], float)
NODE = np.array([-1,-2**-.5,0,2**-.5,1.0])
def rotm(p, y, r):
    p, y, r = -p, -y, -r
    P = np.array([[1,0,0],[0,np.cos(p),np.sin(p)],[0,-np.sin(p),np.cos(p)]])
    Y = np.array([[np.cos(y),0,-np.sin(y)],[0,1,0],[np.sin(y),0,np.cos(y)]])
    R = np.array([[np.cos(r),np.sin(r),0],[-np.sin(r),np.cos(r),0],[0,0,1]])
    return P @ Y @ R
def chain(name, fi, fj):
    if name == 'c2':
        x = fj - 2047.5; y = 2047.5 - fi
        lat = np.hypot(x, y) / (2047.5 / np.pi); lon = np.arctan2(y, x)
        return lat / (np.pi / 4096), lon / (np.pi / 4096) + 4096
    if name == 'c3':
        x = fj - 2047.5; y = 2047.5 - fi
        fd = 2047.5 / (2 * np.sin(np.pi / 2))
        d = np.hypot(x, y) / fd
        lat = 2 * np.arcsin(np.clip(d / 2, -1, 1)); lon = np.arctan2(y, x)
        v = np.stack([np.cos(lon) * np.sin(lat), np.cos(lat), np.sin(lon) * np.sin(lat)])
        R = rotm(*np.radians([30, 45, 10]))
        w = np.tensordot(R, v, 1)
        lat2 = np.arccos(np.clip(w[1], -1, 1)); lon2 = np.arctan2(w[2], w[0])
        dist = lat2 * (2047.5 / np.pi)
        return -np.sin(lon2) * dist + 2047.5, np.cos(lon2) * dist + 2047.5
    if name in ('c1', 'c5'):
        H, W = (2048, 4096) if name == 'c1' else (4096, 8192)
        q = np.pi / W / 2
        lon = -np.pi + q + fj * ((2 * np.pi - 2 * q) / (W - 1)); lat = fi * (np.pi / (H - 1))
        fd = 1535.5 / np.pi if name == 'c1' else 1944.0 / (np.pi / 2)
        c = 1535.5 if name == 'c1' else 1943.5
        dist = lat * fd
        return -np.sin(lon) * dist + c, np.cos(lon) * dist + c
def study(name, H, W):
    ty, tx = np.mgrid[0:H // 32, 0:W // 32]
    fi = (ty[..., None, None] * 32 + 15.5 + 15.5 * NODE[None, None, :, None]) + 0 * NODE[None, None, None, :]
    fj = (tx[..., None, None] * 32 + 15.5 + 15.5 * NODE[None, None, None, :]) + 0 * NODE[None, None, :, None]
    out = {}
    f0, f1 = chain(name, fi, fj)
    for lab, F in (('row', f0), ('col', f1)):
        if name == 'c2' and lab == 'col':
            F = np.unwrap(np.unwrap(F, axis=-1, period=8192), axis=-2, period=8192)
        C = np.einsum('mi,nj,...ij->...mn', A, A, F)
        absC = np.abs(C)
        m, n = np.mgrid[0:5, 0:5]
        for key, mask in (('total>2', m + n > 2), ('total>3', m + n > 3), ('tensor>2', (m > 2) | (n > 2)), ('tensor>3', (m > 3) | (n > 3)), ('total>4', m + n > 4)):
            out.setdefault(key, []).append((absC * mask).sum((-1, -2)))
    n_tiles = ty.size
    print(name, 'tiles', n_tiles, {k: round(float((np.maximum(*v) < 2.0 ** -12).mean()), 3) for k, v in out.items()},
          ' (bound 2^-13:', {k: round(float((np.maximum(*v) < 2.0 ** -13).mean()), 3) for k, v in out.items()}, ')')
for name, H, W in (('c1', 2048, 4096), ('c2', 4096, 4096), ('c3', 4096, 4096), ('c5', 4096, 8192)):
    study(name, H, W)
