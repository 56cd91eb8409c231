#!/bin/bash
# round 3, session h: the whole GPU suite (new: two-eye taps + raw pins at full size, 8x random geometries, RCCL on one GPU, the
# launch-table error path), then the default bench line with its `configs` block, then bench under PB_FORCE_DIST=1 (RCCL, world 1)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3h; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
PB_FORCE_DIST=1 timeout -k 10 600 python bench.py --no-cpu-baseline --no-configs --steps 50 > $O/bench_rccl.json 2> $O/bench_rccl.err; echo "bench rccl rc $?"
python - <<'PY'
import json
for f in ('bench_default','bench_rccl'):
    try:
        j=json.loads(open(f'gpurun_out/r3h/{f}.json').read().strip().splitlines()[-1])
        print(f, j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('ranks_seen'), j.get('collective_backend'), j.get('single_image_ms'), j.get('faithful_kernel_ms'))
        for k,v in j.get("configs",{}).items(): print("  ",k, v["kernel_ms_per_frame"], v.get("frac"), v.get("plan_create_warm_ms"))
        if 'cpu_baseline' in j: print('  cpu', j['cpu_baseline']['value'])
    except Exception as e: print(f, 'ERR', e)
PY
