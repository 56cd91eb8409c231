#!/bin/bash
# usage: variants4.sh <outdir> "<envs;comma-separated K=V> <budget> <config> [extra bench args]" ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
for spec in "$@"; do
  set -- $spec; envs=$1; bud=$2; cfg=$3; shift 3
  e2=${envs//\/root\/repo\/experiments\//}; tag=${cfg}_${e2//[,=.]/_}_b${bud}$(echo "$*" | tr -d ' -')
  env $(echo $envs | tr ',' ' ') timeout -k 10 200 python bench.py --config $cfg --budget $bud --steps 120 --warmup 10 --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err || echo "FAILED $tag"
  python3 - <<PY
import json
try:
    d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; p=r['plan']
    print('%-40s us/frame %7.2f  p10 %7.2f p90 %7.2f  frac %.4f  lean %5d direct %5d' % ('$tag', r['kernel_ms_per_frame']*1e3, r['kernel_ms_p10']*1e3/d['config']['frames_per_launch'], r['kernel_ms_p90']*1e3/d['config']['frames_per_launch'], r['frac'], p['lean_tiles'], p['direct_tiles']))
except Exception as e: print('$tag ERR', e)
PY
done | tee -a $O/summary.txt
