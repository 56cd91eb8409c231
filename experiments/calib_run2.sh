#!/bin/bash
# calibration incl. the sparse kernels: FETCH_SIZE + kernel durations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r2; mkdir -p $O
timeout -k 5 60 $R/experiments/exp_calib || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_calib_stats -- $R/experiments/exp_calib > $O/calib_stats.log 2>&1
cat $O/raw_calib_stats/*/*kernel_stats.csv | cut -d, -f1-4
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw_calib_F -- $R/experiments/exp_calib > $O/calib_F.log 2>&1
python3 $R/experiments/pmc_summary.py $O/raw_calib_F
timeout -k 10 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/raw_calib_R -- $R/experiments/exp_calib > $O/calib_R.log 2>&1
python3 $R/experiments/pmc_summary.py $O/raw_calib_R
rm -rf $O/raw_*
