#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O
timeout -k 5 60 $R/experiments/exp_wc || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_stats -- $R/experiments/exp_wc > $O/stats.log 2>&1
cut -d, -f1-4 $(ls $O/raw_stats/*/*kernel_stats.csv | head -1)
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw_w -- $R/experiments/exp_wc > $O/w.log 2>&1
python3 $R/experiments/pmc_summary.py $O/raw_w
rm -rf $O/raw_*
