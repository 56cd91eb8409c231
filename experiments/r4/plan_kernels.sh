#!/bin/bash
# per-kernel time of one config's plan preparation with each experiments/r4/libpb_f_*.so: plan_kernels.sh <config>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for lib in $R/photonbend_amd/libphotonbend_hip.so $(ls $R/experiments/r4/libpb_f_*.so 2>/dev/null); do
  rm -rf /tmp/pk; PB_LIB_PATH=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $R/experiments/faithful_time.py - $1 > /tmp/pk.log 2>&1
  echo "== $(basename $lib) $(tail -1 /tmp/pk.log | cut -c30-)"
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pk/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f))):
    if 'certify' in r["Name"] or 'sep_check' in r["Name"]: print('   ', r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
