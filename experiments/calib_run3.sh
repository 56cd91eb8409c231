#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_calib_stats -- $R/experiments/exp_calib > $O/calib_stats.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob('$O/raw_calib_stats/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)): print('%-90s calls %s avg %.2f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf $O/raw_*
