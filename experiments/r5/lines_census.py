import sys, numpy as np, math, warnings
warnings.simplefilter('ignore')
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import reference_path as rp
import bench
cfg=bench.CONFIGS['c3']
def proj(t):
    if t[0]=='pano': return rp.Proj('pano', t[1], t[2])
    return rp.Proj(t[0], t[1], t[2], t[3], math.radians(t[4]), t[5])
dst,src=proj(cfg['dst']),proj(cfg['src'])
a,b=rp.pretrunc(dst,src,cfg['rot'])
H,W=a.shape; T=32
yy,xx=np.mgrid[0:H,0:W]
r=np.hypot(yy-2047.5,xx-2047.5)
def tiles(x): return x.reshape(H//T,T,W//T,T).transpose(0,2,1,3).reshape(-1,T,T)
ta,tb,tr=tiles(a-0.5),tiles(b-0.5),tiles(r)
rmin=tr.reshape(-1,T*T).min(1); rmax=tr.reshape(-1,T*T).max(1)
sel=np.where((rmax>2047.5-96)&(rmin<2047.5))[0]   # proxy for c3's table tiles: the outer rings that reach into the circle
print('proxy table tiles', len(sel))
rowbytes=3*src.width
def lines_of(addr):  # distinct 128-B lines of an instruction's addresses (8-byte loads may straddle: count both)
    return len(set((addr//128).tolist())|set(((addr+7)//128).tolist()))
tot={'natural4x4':0,'walk':0,'sorted':0,'sorted_hw':0}; ninstr=0
for t in sel[:400]:
    sy=np.floor(ta[t]).astype(np.int64); sx=np.floor(tb[t]).astype(np.int64)
    addr=sy*rowbytes+3*sx   # tap row r0 (row r0+1: same pattern)
    # (a) natural: lane (xg,yb) handles px (4xg+k, yb+8jr): one instruction = fixed (jr,k), 64 lanes
    for jr in range(4):
        for k in range(4):
            ys=(np.arange(64)//8)+8*jr; xs=4*(np.arange(64)%8)+k
            tot['natural4x4']+=lines_of(addr[ys,xs])
    # (b) the walk: gradients -> by_rows, shear; instruction n: half-wave hh: pixel (p, (2n+hh+shift(p))&31) or transposed
    d_x=np.diff(ta[t],axis=1); d_y=np.diff(ta[t],axis=0)
    gx=d_x[np.abs(d_x)<64].mean() if (np.abs(d_x)<64).any() else 0.0
    gy=d_y[np.abs(d_y)<64].mean() if (np.abs(d_y)<64).any() else 0.0
    by_rows=abs(gy)<abs(gx)
    along,across=(gy,gx) if by_rows else (gx,gy)
    slope=-along/across if across!=0 else 0.0
    slope=min(max(slope,-1.9),1.9); q=int(round(slope*64))
    p=np.arange(32); shift=np.rint(q/64.0*(p-15.5)).astype(int)
    for n in range(16):
        ad=[]
        for hh in range(2):
            bcoord=(2*n+hh+shift)&31
            ad.append(addr[p,bcoord] if by_rows else addr[bcoord,p])
        tot['walk']+=lines_of(np.concatenate(ad))
    # (c) sorted by address: instruction n takes sorted[64n:64n+64]
    s=np.sort(addr.reshape(-1))
    for n in range(16):
        tot['sorted']+=lines_of(s[64*n:64*n+64])
    ninstr+=16
for k,v in tot.items(): print(k, 'lines per load instruction: %.1f'%(v/ninstr))
