"""Per-phase timeline of the hot kernel's waves from the -DPB_TRACE build (experiments/libpb_trace.so):
    python experiments/diag_trace.py <case> [budget]
Phases per tile class (microseconds, mean / p50 / p90), wave concurrency per CU, launch ramp."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
os.environ["PB_LIB_PATH"] = os.path.abspath('experiments/libpb_trace.so')
import photonbend_amd._native as nat
from tests import helpers as H
from tests.cases import full_cases

case = [c for c in full_cases() if c.name == sys.argv[1]][0]
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 12288
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # > 1: one launch of `batch` frames, the middle frame's waves are recorded
src, cmap = H.pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
plan = nat.Plan(cmap.dst_proj, cmap.rotations, src._proj(), budget=budget)
lib = nat.load()
_, h, w, *_ = case.src
frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(6)]
outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(6)]
for i in range(6): plan.remap(frames[i], outs[i])
torch.cuda.synchronize()
info = plan.info()
nt = info['tiles']
buf = (ctypes.c_ulonglong * (nt * 16))()
lib.pb_debug_trace(None, 0, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if batch > 1:
    fb = torch.stack([frames[i % 6] for i in range(batch)]); ob = torch.empty((batch,) + tuple(outs[0].shape), dtype=torch.uint8, device='cuda')
    plan.remap(fb, ob); torch.cuda.synchronize()
    assert lib.pb_debug_trace_frame(plan.handle, batch // 2) == 0
    lib.pb_debug_trace(None, 0, 1)
    e0.record(); plan.remap(fb, ob); e1.record(); torch.cuda.synchronize()
    print('batch of %d frames: %.1f us per frame' % (batch, e0.elapsed_time(e1) * 1e3 / batch))
else:
    e0.record(); plan.remap(frames[0], outs[0]); e1.record(); torch.cuda.synchronize()
assert lib.pb_debug_trace(buf, nt * 16, 0) == 0
T = np.frombuffer(buf, dtype=np.uint64).reshape(nt, 16).astype(np.int64)
tab = (ctypes.c_int32 * (nt * 64))()
assert lib.pb_debug_copy_table(plan.handle, tab, nt * 256) == 0
flags = np.frombuffer(tab, dtype=np.int32).reshape(nt, 64)[:, 2]
if case.src[0] == 'double':
    names = ['start->descs', 'descs->issued', 'issued->addr (math)', 'addr->landed', 'landed->stores issued', 'stores issued->done', 'done->wave end']
t0 = T[:, 0][T[:, 0] > 0].min()
us = lambda x: x * 0.01  # 100 MHz
print('event time %.1f us; trace span %.1f us (first wave start -> last wave done); budget %d' % (e0.elapsed_time(e1) * 1e3, us(T[:, 7].max() - t0), info['window_budget']), info)
names0 = ['start->entry', 'entry->addr', 'addr->issued', 'issued->landed', 'landed->stores issued', 'stores issued->done', 'done->wave end']
names = names if case.src[0] == 'double' else names0
classes = {'LEAN': (flags & 4) != 0, 'DIRECT': (flags & 16) != 0, 'BLACK': (flags & 8) != 0, 'GENERIC': (flags & (4 | 16 | 8 | 2)) == 0, 'FAILED': (flags & 2) != 0}
for cname, m in classes.items():
    n = int(m.sum())
    if not n: continue
    Tm = T[m]
    life = us(Tm[:, 7] - Tm[:, 0])
    print('%-8s %6d tiles  wave life mean %.2f us p50 %.2f p90 %.2f' % (cname, n, life.mean(), np.percentile(life, 50), np.percentile(life, 90)))
    if cname in ('LEAN', 'DIRECT') or case.src[0] == 'double':
        for i, nm in enumerate(names):
            a, b = (i, i + 1)
            d = us(Tm[:, b] - Tm[:, a])
            print('     %-24s mean %.2f  p50 %.2f  p90 %.2f' % (nm, d.mean(), np.percentile(d, 50), np.percentile(d, 90)))
    else:
        d = us(Tm[:, 1] - Tm[:, 0]); print('     %-24s mean %.2f' % ('start->entry', d.mean()))
        d = us(Tm[:, 6] - Tm[:, 1]); print('     %-24s mean %.2f' % ('entry->tile done', d.mean()))
# concurrency: waves alive over time, per CU (HW_ID: bits 8-11 CU id, 13-15 SE id, ... ; use the whole id minus wave/simd bits)
hw = T[:, 15]
cu_key = (hw >> 8) & 0xFFFFF  # cu, sh, se, ... (coarse: everything above the wave/simd fields)
starts, ends = T[:, 0], T[:, 7]
grid = np.arange(t0, ends.max(), 50)  # every 0.5 us
alive = [(int(((starts <= g) & (ends > g)).sum())) for g in grid]
print('waves alive chip-wide every 0.5 us:', alive)
for cname, m in classes.items():
    if cname == 'FAILED' or not m.any(): continue
    print('  alive %-8s every 2 us:' % cname, [int(((starts[m] <= g) & (ends[m] > g)).sum()) for g in grid[::4]])
print('distinct CU keys:', len(np.unique(cu_key)))
order = np.argsort(starts)
print('wave starts (us after first) deciles:', [round(us(np.percentile(starts - t0, q)), 1) for q in range(0, 101, 10)])
# per-XCD view (HW_REG_XCC_ID of the wave): does one XCD finish late?
if True:
    xcd = T[:, 14] & 15; wgid = T[:, 14] >> 8
    print('XCC_ID == workgroup & 7 for %.1f %% of waves' % (100.0 * ((wgid & 7) == xcd)[starts > 0].mean()))
    for x in range(8):
        m = (xcd == x) & (starts > 0) & (ends > 0)
        if not m.any(): continue
        print('XCD %d: %5d tiles (%5d black) first start %.1f last start %.1f last end %.1f us; busy wave-us %.0f' % (
            x, int(m.sum()), int((m & classes['BLACK']).sum()), us(starts[m].min() - t0), us(starts[m].max() - t0), us(ends[m].max() - t0), us((ends[m] - starts[m]).sum())))
if os.environ.get('PB_TRACE_SAVE'):
    extra = {}
    if case.src[0] == 'double':
        tab_r = (ctypes.c_int32 * (nt * 64))()
        if lib.pb_debug_copy_table_r(plan.handle, tab_r, nt * 256) == 0:
            extra['table_r'] = np.frombuffer(tab_r, dtype=np.int32).reshape(nt, 64)
    np.savez_compressed(os.environ['PB_TRACE_SAVE'], T=T, table=np.frombuffer(tab, dtype=np.int32).reshape(nt, 64), **extra)
