#!/bin/bash
# round 4 baseline: where does the bilinear mode's time go (tile kernel vs float64 pass), per config
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in c1 c2 c3 c5; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$c -- python3 $R/bench.py --config $c --sampling bilinear --steps 40 --warmup 5 --no-cpu-baseline --no-configs > $O/${c}_bench.json 2> $O/${c}.err || exit 1
  cp $(ls $O/raw_$c/*/*kernel_stats.csv | head -1) $O/${c}_kernel_stats.csv
  echo "== $c"; head -5 $O/${c}_kernel_stats.csv | cut -c1-200
done
rm -rf $O/raw_*
