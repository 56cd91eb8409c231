#!/bin/bash
# round-2 GPU session A: full GPU test suite, one bench line per config, 2-rank rehearsal, kernel stats of c2
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
for cfg in c2 c1 c3 c5 c4shard c5shard; do
  timeout -k 10 300 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc=$?" | tee -a $O/summary.txt
done
for cfg in c4shard c5shard; do
  timeout -k 10 300 python bench.py --config $cfg --batch 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_${cfg}_b1.json 2> $O/bench_${cfg}_b1.err; echo "bench $cfg b1 rc=$?" | tee -a $O/summary.txt
done
PB_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 40 --warmup 5 > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "bench 2rank rc=$?" | tee -a $O/summary.txt
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_c2.log 2>&1; echo "prof rc=$?" | tee -a $GRAFT_REPO_ROOT/$O/summary.txt
cd $GRAFT_REPO_ROOT
cat $O/summary.txt; tail -3 $O/pytest_gpu.log; for f in $O/bench_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print(d['config']['name'], 'B',d['config']['frames_per_launch'],'n_gpus',d['n_gpus'],'value',d['value'],'ms/frame',r['kernel_ms_per_frame'],'frac',r['frac'],'budget',r['window_budget'],'create',d['plan_create_ms'],'first',d['first_frame_ms'],'copy',r['copy_ceiling_gbs'],'att',r['attainable_frac'], 'p10/p90', r['kernel_ms_p10'], r['kernel_ms_p90'])
except Exception as e: print('ERR', e)
"; done
