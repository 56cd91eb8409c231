#!/bin/bash
# HIP API time inside warm plan preparation: plan_api.sh <config>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa; timeout -k 10 200 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/pa -- python3 $R/experiments/r4/plan_only.py $1 > /tmp/pa.log 2>&1
tail -1 /tmp/pa.log
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pa/*/*hip_api_stats.csv")
if not f: print("no hip_api_stats", glob.glob("/tmp/pa/*/*")); raise SystemExit
rows=sorted(csv.DictReader(open(f[0])), key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:14]: print('   ', r["Name"][:40].ljust(40), r["Calls"].rjust(6), str(round(float(r["AverageNs"])/1e3,1)).rjust(8), 'us avg', str(round(float(r["TotalDurationNs"])/1e3/21,1)).rjust(8), 'us per plan')
PY
