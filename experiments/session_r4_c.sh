#!/bin/bash
# VERDICT r3 item 4, the closing measurement: the store shape of a 64 x 16 strip (PB_EXP=1024 in the diagnostic build: whole 192-byte row
# pieces per store instruction, loads unchanged, pixels in the wrong places) = the upper bound of ANY two-sub-tile strip variant, against this
# round's kernels, cold frame pools: single launches and 8 frames per launch, and WRITE_SIZE of both shapes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c_strip; mkdir -p $O; cd $R
for exp in 0 1024 0 1024 0 1024; do
  PB_EXP=$exp timeout -k 10 300 python experiments/ab_case.py build/libphotonbend_hip_diag.so c3 c1 c3:8 c1:8 2>> $O/ab.err | sed "s/^/EXP=$exp /" >> $O/ab.log
done
cut -c1-125 $O/ab.log
cd /tmp && export TMPDIR=/tmp
for exp in 0 1024; do for c in c3 c1; do
  PB_EXP=$exp PB_LIB_PATH=$R/build/libphotonbend_hip_diag.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw -- python3 $R/bench.py --config $c --steps 12 --warmup 2 --no-cpu-baseline --no-configs --no-events > /dev/null 2>> $O/pmc.err
  echo "EXP=$exp $c WRITE_SIZE (KB): $(python3 $R/experiments/pmc_summary.py $O/raw | grep -A1 'pb_hot_win' | tail -1)" | tee -a $O/write.log
  rm -rf $O/raw
done; done
