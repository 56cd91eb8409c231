#!/bin/bash
# round 3, session j: bilinear for double-fisheye sources through the per-eye tile models
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_bilinear.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -15 $O/tests.log | cut -c1-220
for cfg in c5 c2 c3 c1; do
  timeout -k 10 300 python bench.py --config $cfg --sampling bilinear --no-cpu-baseline --steps 40 --warmup 5 > $O/bil_$cfg.json 2> $O/bil_$cfg.err
  python -c "
import json,sys
j=json.loads(open('$O/bil_$cfg.json').read().strip().splitlines()[-1]); print('$cfg bilinear', j['ms_per_step'], 'ms/step')"
done
timeout -k 10 300 python bench.py --config c5shard --sampling bilinear --no-cpu-baseline --steps 10 --warmup 2 > $O/bil_c5shard.json 2> $O/bil_c5shard.err
python -c "
import json
j=json.loads(open('$O/bil_c5shard.json').read().strip().splitlines()[-1]); print('c5shard bilinear', j['ms_per_step']/j['config']['frames_per_launch'], 'ms/frame')"
