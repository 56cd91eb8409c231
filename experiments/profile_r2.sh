#!/bin/bash
# round-2 profiles: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only) per config at the
# budget bench.py pins for it, plus the counter calibration kernels.  Summaries land in gpurun_out/prof_r2/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # tag, bench args...
  tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_${tag}_stats -- python3 $R/bench.py "$@" --steps 60 --warmup 10 --no-cpu-baseline > $O/${tag}_bench.json 2> $O/${tag}_stats.err
  cp $(ls $O/raw_${tag}_stats/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/raw_${tag}_$c -- python3 $R/bench.py "$@" --steps 12 --warmup 2 --no-cpu-baseline --no-events > /dev/null 2> $O/${tag}_$c.err
    python3 $R/experiments/pmc_summary.py $O/raw_${tag}_$c > $O/${tag}_pmc_$c.txt
  done
  echo "== $tag"; head -4 $O/${tag}_kernel_stats.csv; grep -A2 "pb_hot" $O/${tag}_pmc_FETCH_SIZE.txt | head -3; grep -A2 "pb_hot" $O/${tag}_pmc_WRITE_SIZE.txt | head -3
}
run c2 --config c2
run c1 --config c1
run c3 --config c3
run c5 --config c5
run c4shard --config c4shard
run c5shard --config c5shard
run c2_alldirect --config c2 --budget 4224
run c2_b12288 --config c2 --budget 12288
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/raw_calib_$c -- $R/experiments/exp_calib > $O/calib_$c.log 2>&1
  python3 $R/experiments/pmc_summary.py $O/raw_calib_$c > $O/calib_pmc_$c.txt; cat $O/calib_pmc_$c.txt
done
rm -rf $O/raw_*
