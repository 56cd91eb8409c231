#!/bin/bash
# occ.sh label=lib ...: SQ counters of the c5 bilinear kernel under builds on THIS box (one rocprofv3 --pmc pass each, kernel-trace only): waves, VALU per
# wave, resident waves, VALU busy, and the kernel's duration in the same pass (-> effective shader clock)
R=$GRAFT_REPO_ROOT; cd $R
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
B="--config ${CFG:-c5} --sampling bilinear --steps 12 --warmup 2 --no-cpu-baseline --no-configs --no-events --no-live-traffic"
O=$R/gpurun_out/r6_occ; mkdir -p $O
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ -n "$lib" ]; then export PB_LIB_PATH=$R/$lib; else unset PB_LIB_PATH; fi
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $O/raw_$label -- python3 $R/bench.py $B > $O/$label.log 2>&1 ) || { echo "FAILED $label"; tail -5 $O/$label.log; exit 1; }
  python3 - $O/raw_$label $label <<'PY'
import csv, glob, sys, collections
d, label = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bilinear' in r['Kernel_Name'] and 'hot_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
dur = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bilinear' in r['Kernel_Name'] and 'hot_kernel' in r['Kernel_Name']:
            dur.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
v = {k: sum(x) / len(x) for k, x in acc.items()}
w = v['SQ_WAVES']; cyc = v['SQ_BUSY_CYCLES'] / 32.0; ns = sum(dur) / len(dur)
print(f"{label}: waves {w:.0f}  VALU/wave {v['SQ_INSTS_VALU'] / w:.0f}  kernel {ns / 1e3:.1f} us under the counters = {cyc:.0f} cycles (SQ_BUSY_CYCLES / 32) -> {cyc / ns * 1e3:.0f} MHz  "
      f"wave life {4 * v['SQ_WAVE_CYCLES'] / w:.0f} cycles  resident waves / SIMD {4 * v['SQ_WAVE_CYCLES'] / cyc / 1024:.2f}  VALU busy {v['SQ_ACTIVE_INST_VALU'] / (8 * v['SQ_BUSY_CYCLES']):.3f}  "
      f"waiting {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}  issue-stalled {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}", flush=True)
PY
  rm -rf $O/raw_$label
done
