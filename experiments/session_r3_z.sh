#!/bin/bash
# PB_TILE_COARSE: fuzz the bilinear tile path on noise frames, count float64 tiles of the BASELINE plans, time bilinear + preparation
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3z; mkdir -p $O
timeout -k 10 500 python experiments/fast_vs_faithful_sweep.py 400 20000 1 > $O/s1.log 2>&1; tail -2 $O/s1.log | cut -c1-200
timeout -k 10 500 python experiments/fast_vs_faithful_sweep.py 150 30000 5 > $O/s5.log 2>&1; tail -2 $O/s5.log | cut -c1-200
timeout -k 10 300 python - > $O/counts.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import bench, torch, time
from photonbend_amd import _native as nat
for name in ('c1', 'c2', 'c3', 'c5'):
    cfg = bench.CONFIGS[name]; d, rots, s = bench.build_projs(cfg)
    plan = nat.Plan(d, rots, s); i = plan.info()
    src = nat.synth_frame(s.height, s.width, frame=1, seed=0, circle_mask=cfg['mask'])
    for _ in range(3): out = plan.remap(src, interpolation='bilinear')
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = plan.remap(src, out, interpolation='bilinear')
    e1.record(); torch.cuda.synchronize()
    print(name, 'tiles', i['tiles'], 'failed', i['fix_tiles'], 'bilinear float64 tiles', i['bilinear_float64_tiles'], 'bilinear us/frame %.1f' % (e0.elapsed_time(e1) * 100))
PY
cat $O/counts.log
timeout -k 10 300 python experiments/faithful_time.py - >> $O/counts.log 2>&1; tail -4 $O/counts.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "bilinear or plan_api or plan" > $O/tests.log 2>&1; tail -3 $O/tests.log
