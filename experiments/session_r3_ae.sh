#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ae; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_plan.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r3ae/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_mean'], j['roofline']['frac'], j.get('wall_ms_per_frame_by_streams'), j.get('graph_replay_ms_per_frame'), j.get('single_image_ms'), j.get('faithful_kernel_ms'))
for k, v in j['configs'].items(): print(' ', k, v['kernel_ms_per_frame'], v.get('frac'), v.get('wall_ms_per_frame_by_streams'), v.get('plan_create_warm_ms'))
PY
