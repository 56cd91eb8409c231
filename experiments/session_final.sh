#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
PB_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 40 --warmup 5 > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "2rank rc=$?"
python bench.py --gpus 3 > /dev/null 2>&1; echo "gpus 3 on a 1-GPU box rc=$? (expect 2)"
python3 - <<PY
import json
for f in ('bench_default','bench_2rank'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f, 'n_gpus',d['n_gpus'],'value',d['value'],'ms/step',d['ms_per_step'],'kernel',r['kernel_ms_mean'],'frac',r['frac'],'traffic',r['traffic'],'copy',r['copy_ceiling_gbs'],'create',d['plan_create_ms'],d['plan_create_warm_ms'],'cpu',d.get('cpu_baseline',{}).get('value'))
PY
