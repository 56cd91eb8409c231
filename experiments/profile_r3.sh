#!/bin/bash
# round-3 profiles: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only) per config at the budget
# bench.py pins for it.  Summaries land in gpurun_out/prof_r3/; experiments/make_traffic_r3.py turns them into profiles/r03_*.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # tag, bench args...
  tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_${tag}_stats -- python3 $R/bench.py "$@" --steps 60 --warmup 10 --no-cpu-baseline --no-configs > $O/${tag}_bench.json 2> $O/${tag}_stats.err
  cp $(ls $O/raw_${tag}_stats/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/raw_${tag}_$c -- python3 $R/bench.py "$@" --steps 12 --warmup 2 --no-cpu-baseline --no-configs --no-events > /dev/null 2> $O/${tag}_$c.err
    python3 $R/experiments/pmc_summary.py $O/raw_${tag}_$c > $O/${tag}_pmc_$c.txt
  done
  echo "== $tag"; head -4 $O/${tag}_kernel_stats.csv | cut -c1-160; grep -A2 "pb_hot" $O/${tag}_pmc_FETCH_SIZE.txt | head -3; grep -A2 "pb_hot" $O/${tag}_pmc_WRITE_SIZE.txt | head -3
}
run c2 --config c2
run c1 --config c1
run c3 --config c3
run c5 --config c5
run c4shard --config c4shard
run c5shard --config c5shard
rm -rf $O/raw_*
