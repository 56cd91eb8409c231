#!/bin/bash
# every kernel of one config's plan preparation (6 plans per run: 1 + 5 warm), average us per call: plan_breakdown.sh <config>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $R/experiments/faithful_time.py - $1 > /tmp/pk.log 2>&1
tail -1 /tmp/pk.log
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pk/*/*kernel_stats.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:16]: print('   ', r["Name"][:60].ljust(60), r["Calls"].rjust(5), str(round(float(r["AverageNs"])/1e3,1)).rjust(8), 'us avg', str(round(float(r["TotalDurationNs"])/1e3/6,1)).rjust(8), 'us per plan')
PY
