#!/bin/bash
# bilinear mode, single sources: sheared direct gathers (product) - tests, then us per frame for c1 c2 c3 (previous build: libpb_prev.so)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3as; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_bilinear.py tests/test_cli.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
timeout -k 10 400 python - > $O/time.log 2>&1 <<'PY'
import os, sys, subprocess
code = r'''
import os, sys; sys.path.insert(0, '.')
import bench, torch
from photonbend_amd import _native as nat
for name in ('c1', 'c2', 'c3'):
    for b in (1, 8):
        cfg = bench.CONFIGS[name]; d, rots, s = bench.build_projs(cfg)
        plan = nat.Plan(d, rots, s)
        n = 12 if b == 1 else 16
        srcs = torch.stack([nat.synth_frame(s.height, s.width, frame=f, seed=0, circle_mask=cfg['mask']) for f in range(n)])
        outs = torch.empty((n, d.height, d.width, 3), dtype=torch.uint8, device='cuda')
        def go(k):
            i = (k * b) % n
            plan.remap(srcs[i:i + b] if b > 1 else srcs[i], outs[i:i + b] if b > 1 else outs[i], interpolation='bilinear')
        for k in range(6): go(k)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(24): go(k)
        e1.record(); torch.cuda.synchronize()
        print(os.environ.get('PB_LIB_PATH', 'product')[-16:], name, 'frames per launch', b, 'bilinear us/frame %.1f' % (e0.elapsed_time(e1) * 1e3 / 24 / b), flush=True)
        del srcs, outs, plan
'''
for lib in ('experiments/libpb_prev.so', '', 'experiments/libpb_prev.so', ''):
    env = dict(os.environ)
    if lib: env['PB_LIB_PATH'] = os.path.abspath(lib)
    print(subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True).stdout, flush=True)
PY
cat $O/time.log
