#!/bin/bash
# do BATCH launches gain from several streams too?  (wall ms per frame = ms_per_step / frames per launch)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3af; mkdir -p $O
for cfg in c4shard c5shard "c3 --batch 8" "c1 --batch 8"; do
for n in 1 2 3 1 2 3; do
  timeout -k 10 300 python bench.py --config $cfg --streams $n --no-configs --no-cpu-baseline --steps 60 2>> $O/err.log | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg streams $n: value %.0f Mpx/s  wall ms per frame %.5f' % (j['value'], j['ms_per_step'] / j['config']['frames_per_launch']))" >> $O/streams.log
done
done
cat $O/streams.log
