#!/bin/bash
# wave timelines of c5 (the two-eye kernel): -DPB_TRACE build, single launch and the middle frame of a batch of 8
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ab; mkdir -p $O
PB_TRACE_SAVE=$O/c5_single.npz timeout -k 10 300 python experiments/diag_trace.py c5_180 7168 > $O/c5_single.log 2>&1
PB_TRACE_SAVE=$O/c5_batch.npz timeout -k 10 300 python experiments/diag_trace.py c5_180 7168 8 > $O/c5_batch.log 2>&1
python - <<'PY'
import numpy as np
for tag in ('single', 'batch'):
    z = np.load(f'gpurun_out/r3ab/c5_{tag}.npz'); T = z['T']
    us = lambda x: x * 0.01
    ok = T[:, 0] > 0
    fl, fr = z['table'][:, 2], z['table_r'][:, 2]
    two = ok & ((fl & 8) == 0) & ((fr & 8) == 0)   # neither eye's tile is BLACK
    solo = ok & ~two
    t0 = T[ok, 0].min()
    print(tag, 'tiles', int(ok.sum()), 'two-eye', int(two.sum()), 'solo', int(solo.sum()), 'span %.1f us' % us(T[ok, 7].max() - t0))
    for nm, m in (('two-eye', two), ('solo', solo)):
        life = us(T[m, 7] - T[m, 0])
        print('  %-8s wave life mean %.2f p50 %.2f p90 %.2f max %.2f; starts p10 %.1f p50 %.1f p90 %.1f; ends max %.1f' % (nm, life.mean(), np.percentile(life, 50), np.percentile(life, 90), life.max(),
              *[us(np.percentile(T[m, 0] - t0, q)) for q in (10, 50, 90)], us(T[m, 7].max() - t0)))
    names = ['start->entry', 'entry->descs', 'descs->issued', 'issued->math done', 'math->landed', 'landed->stores issued', 'stores->done']
    for i, nm in enumerate(names):
        d = us(T[two, i + 1] - T[two, i])
        print('     two-eye %-24s mean %.2f p50 %.2f p90 %.2f' % (nm, d.mean(), np.percentile(d, 50), np.percentile(d, 90)))
    d = us(T[solo, 1] - T[solo, 0]); print('     solo start->entry mean %.2f; entry->done mean %.2f' % (d.mean(), us(T[solo, 6] - T[solo, 1]).mean()))
    grid = np.arange(t0, T[ok, 7].max(), 200)
    print('  alive two-eye every 2 us:', [int(((T[two, 0] <= g) & (T[two, 7] > g)).sum()) for g in grid])
    print('  alive solo    every 2 us:', [int(((T[solo, 0] <= g) & (T[solo, 7] > g)).sum()) for g in grid])
PY
head -3 $O/c5_single.log; head -3 $O/c5_batch.log
