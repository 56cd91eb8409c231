#!/bin/bash
# round 4: bilinear tests, then per-config kernel times of the bilinear mode (rocprofv3 kernel stats); $1 = output tag
R=$GRAFT_REPO_ROOT; T=${1:-r4b}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R && timeout -k 10 900 python -m pytest tests/test_hip_bilinear.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for c in c1 c2 c3 c5; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$c -- python3 $R/bench.py --config $c --sampling bilinear --steps 40 --warmup 5 --no-cpu-baseline --no-configs > $O/${c}_bench.json 2> $O/${c}.err || { echo "bench $c failed"; tail -5 $O/${c}.err; exit 1; }
  cp $(ls $O/raw_$c/*/*kernel_stats.csv | head -1) $O/${c}_kernel_stats.csv
  echo "== $c"; grep -i "bilinear" $O/${c}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-
done
rm -rf $O/raw_*
