#!/bin/bash
# every kernel of a config's plan preparation (rocprofv3 kernel stats) with the given library: plan_profile.sh <lib|-> <config...>
R=$GRAFT_REPO_ROOT; lib=$1; shift; cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/pk; [ "$lib" != "-" ] && export PB_LIB_PATH=$R/$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $R/experiments/faithful_time.py - $c > /tmp/pk.log 2>&1
  echo "== $c $(grep 'faithful kernel' /tmp/pk.log | cut -c30-)"
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pk/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    if r["Name"].startswith(("void pb_","pb_")): print('   %-62s %3s x %8.1f us' % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
