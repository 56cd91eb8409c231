#!/bin/bash
# round 3, session i: correctly rounded sin / cos / atan2 in the faithful chain - the whole GPU suite (every reference pin must
# hold), identity-remap differences and map ulps as printed by the tests, plan creation and faithful-kernel times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3i; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
grep -h "identity\|differ\|\[maps D_pano_chain\|\[maps C_alter\|\[maps D_photo_rot" $O/tests.log | cut -c1-230 | head -40
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 50 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3i/bench.json').read().strip().splitlines()[-1])
print('c2', j['ms_per_step'], 'plan cold/warm', j['plan_create_ms'], j['plan_create_warm_ms'], 'single_image', j.get('single_image_ms'), 'faithful', j.get('faithful_kernel_ms'))
for k,v in j.get('configs',{}).items(): print('  ',k, v['kernel_ms_per_frame'], 'plan warm', v['plan_create_warm_ms'])
PY
