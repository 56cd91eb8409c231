#!/bin/bash
# round-6 profiles: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only) per config at the budget bench.py
# pins for it - the nearest (reference) sampler AND the opt-in bilinear mode.  Summaries land in gpurun_out/prof_r6/;
# experiments/make_traffic_r6.py turns them into profiles/r06_* and profiles/traffic_<config>_<budget>.json.
# usage: profile_r6.sh [nearest|bilinear|both] [config ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # tag, bench args...
  tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_${tag}_stats -- python3 $R/bench.py "$@" --steps 60 --warmup 10 --no-cpu-baseline --no-configs --no-live-traffic --detail $O/${tag}_bench.json > $O/${tag}_line.json 2> $O/${tag}_stats.err || return 1
  cp $(ls $O/raw_${tag}_stats/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/raw_${tag}_$c -- python3 $R/bench.py "$@" --steps 12 --warmup 2 --no-cpu-baseline --no-configs --no-events --no-live-traffic > /dev/null 2> $O/${tag}_$c.err || return 1
    python3 $R/experiments/pmc_summary.py $O/raw_${tag}_$c > $O/${tag}_pmc_$c.txt
  done
  echo "== $tag"; head -3 $O/${tag}_kernel_stats.csv | cut -c1-150
  rm -rf $O/raw_${tag}_*
}
what=${1:-both}; shift
near=${@:-c2 c1 c3 c5 c4shard c5shard}; bil=${@:-c2 c1 c3 c5}   # optional: the configs to run
if [ $what != bilinear ]; then
  for c in $near; do run $c --config $c || exit 1; done
fi
if [ $what != nearest ]; then
  for c in $bil; do run ${c}_bilinear --config $c --sampling bilinear || exit 1; done
fi
