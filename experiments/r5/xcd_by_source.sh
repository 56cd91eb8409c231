#!/bin/bash
# VERDICT r4 item 8 - ONE bounded experiment: camera -> camera / camera -> pano plans with their super-tiles dealt to XCDs by the eighth of the
# SOURCE rows they sample (PB_XCD_BY_SRC=1, diagnostic build) against today's rule; kernel time (bench.py, HIP events) and FETCH_SIZE / WRITE_SIZE.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_xcd; mkdir -p $O; cd $R
export PB_LIB_PATH=$R/build/libphotonbend_hip_diag.so
for rep in 1 2 3; do
  for c in c3 c1; do
    for k in 0 1; do
      l=$(PB_XCD_BY_SRC=$k timeout -k 10 120 python3 bench.py --config $c --steps 80 --warmup 10 --no-cpu-baseline --no-configs 2>>$O/err.log | tail -1)
      echo "time $c by_src=$k $(echo "$l" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms_per_frame"])')" | tee -a $O/xcd.log
      l=$(PB_XCD_BY_SRC=$k timeout -k 10 120 python3 bench.py --config $c --batch 8 --steps 40 --warmup 5 --no-cpu-baseline --no-configs 2>>$O/err.log | tail -1)
      echo "time8 $c by_src=$k $(echo "$l" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms_per_frame"])')" | tee -a $O/xcd.log
    done
  done
done
cd /tmp && export TMPDIR=/tmp
for c in c3 c1; do
  for k in 0 1; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      PB_XCD_BY_SRC=$k timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/raw_${c}_${k}_$ctr -- python3 $R/bench.py --config $c --steps 12 --warmup 2 --no-cpu-baseline --no-configs --no-events > /dev/null 2>> $O/err.log
      echo "pmc $c by_src=$k $(python3 $R/experiments/pmc_summary.py $O/raw_${c}_${k}_$ctr | grep -A3 pb_hot_win | grep $ctr)" | tee -a $O/xcd.log
      rm -rf $O/raw_${c}_${k}_$ctr
    done
  done
done
