#!/bin/bash
# round 3, session r: XCD = angular sector of the output (PB_ORDER=4) against the shipped launch order, c2: time and FETCH_SIZE
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3r; mkdir -p $O
for ord in 0 4 0 4; do
  PB_ORDER=$ord timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c2 c2:8 2>> $O/ab.err | sed "s/^/ORDER=$ord /" >> $O/ab.log
done
cut -c1-120 $O/ab.log
cd /tmp && export TMPDIR=/tmp
for ord in 0 4; do
  PB_ORDER=$ord PB_LIB_PATH=$GRAFT_REPO_ROOT/build/libphotonbend_hip_diag.so timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/raw_$ord -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 12 --warmup 2 --no-cpu-baseline --no-configs --no-events > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_$ord.err
  echo "ORDER=$ord"; python3 $GRAFT_REPO_ROOT/experiments/pmc_summary.py $GRAFT_REPO_ROOT/$O/raw_$ord | grep -A1 "pb_hot_win"
done
rm -rf $GRAFT_REPO_ROOT/$O/raw_*
