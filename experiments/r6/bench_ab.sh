#!/bin/bash
# same-box: bench.py's own figure for a config under two library builds, alternating.  bench_ab.sh "<bench args>" label=lib.so ...   (empty lib = the product)
R=$GRAFT_REPO_ROOT; cd $R; ARGS=$1; shift
for rep in 1 2; do for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ -n "$lib" ]; then export PB_LIB_PATH=$R/$lib; else unset PB_LIB_PATH; fi
  l=$(timeout -k 10 200 python3 bench.py $ARGS --no-cpu-baseline --no-configs --no-live-traffic 2>/dev/null | tail -1)
  echo "$label $(echo "$l" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], r["kernel_ms_mean"], r["kernel_ms_p10"], r["kernel_ms_p90"])')"
done; done
