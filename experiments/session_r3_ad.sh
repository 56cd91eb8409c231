#!/bin/bash
# independent single-frame launches dealt to 1 / 2 / 3 / 4 streams: does the next launch's ramp hide in the previous one's drain?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ad; mkdir -p $O
for cfg in c2 c5 c3 c1; do
for n in 1 2 3 4 1 2; do
  timeout -k 10 300 python bench.py --config $cfg --streams $n --no-configs --no-cpu-baseline --steps 200 2>> $O/err.log | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg streams $n: value %.0f Mpx/s  ms_per_step %.5f  kernel_ms_mean %.5f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms_mean']))" >> $O/streams.log
done
done
cat $O/streams.log
