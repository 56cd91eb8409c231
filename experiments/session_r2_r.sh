#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 1 0; do PB_SOLO=$v python -m pytest tests/test_hip_double.py tests/test_hip_full.py tests/test_hip_plan.py tests/test_plan_api.py tests/test_hip_parity.py tests/test_generic.py tests/test_hip_random.py -m gpu -x -q 2>&1 | tail -2; done
for rep in 1 2; do
bash experiments/variants4.sh r2r "PB_SOLO=0 7168 c5" "PB_SOLO=1 7168 c5" "PB_SOLO=0 7168 c5shard" "PB_SOLO=1 7168 c5shard" "PB_SOLO=1 8176 c5" "PB_SOLO=1 12288 c5"
done
python3 -c "
import json
d=json.loads(open('gpurun_out/r2r/c5_PB_SOLO_1_b7168.json').read().strip().splitlines()[-1]); print(d['roofline']['plan'])"
