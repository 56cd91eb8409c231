"""How well does a per-tile tensor-product polynomial (nodes on a (D+1)x(D+1) Chebyshev-Lobatto
grid inside the tile) reproduce the exact float64 pre-truncation source coordinates?"""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import reference_path as orc
from tests.cases import full_cases
from tests import helpers as H

def study(case, T, D, sub=None):
    od, os_ = H.orc_proj(case.dst), H.orc_proj(case.src)
    co = orc.pretrunc(od, os_, H.orc_rots(case))
    Hh, Ww = co[0].shape
    # nodes: Chebyshev-Lobatto points mapped to pixel positions [0, T-1] (not integer!) -> we need the function
    # at non-integer pixel positions; approximate study: use equispaced integer nodes instead (0, (T-1)/D ...)
    nodes = np.round(np.linspace(0, T - 1, D + 1)).astype(int)
    u = np.arange(T)
    # Lagrange basis on nodes evaluated at all u
    L = np.ones((D + 1, T))
    for i in range(D + 1):
        for j in range(D + 1):
            if i != j:
                L[i] *= (u - nodes[j]) / (nodes[i] - nodes[j])
    res = []
    for a in co[:2]:
        a = a[: Hh // T * T, : Ww // T * T]
        t = a.reshape(Hh // T, T, Ww // T, T).transpose(0, 2, 1, 3)  # tiles
        F = t[:, :, nodes][:, :, :, nodes]                           # (ty,tx,D+1,D+1)
        fit = np.einsum('abij,iu,jv->abuv', F, L, L)
        err = np.abs(fit - t).max(axis=(2, 3))
        res.append(err)
    e = np.maximum(res[0], res[1])
    fin = np.isfinite(e)
    print(f"{case.name} T={T} D={D}: tiles {e.size}, finite {fin.sum()}, err<1e-6 {np.mean(e[fin]<1e-6):.3f}, <1e-5 {np.mean(e[fin]<1e-5):.3f}, <1e-4 {np.mean(e[fin]<1e-4):.3f}, <1e-3 {np.mean(e[fin]<1e-3):.3f}, <1e-2 {np.mean(e[fin]<1e-2):.3f}; median {np.median(e[fin]):.2e}")
    return e

cases = {c.name: c for c in full_cases()}
for name in sys.argv[1].split(','):
    for T, D in [(32, 3), (32, 4), (16, 3), (32, 5)]:
        study(cases[name], T, D)
