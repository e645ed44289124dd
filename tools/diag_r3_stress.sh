#!/bin/bash
# Fresh-process stress of the drop-in loop (graphs + streams against eager), with crash tracers
# (tools/stress_first_run.py prints the native and the Python stack of a crash).  Stops at the first crash.
#   tools/diag_r3_stress.sh OUTDIR N [ENV=VALUE ...]
out=${1:-gpurun_out/d1}; n=${2:-8}; shift 2
mkdir -p "$out"
export STRESS_EPOCHS=${STRESS_EPOCHS:-5}
for kv in "$@"; do export "$kv"; done
bad=0
for i in $(seq 1 "$n"); do
  timeout -k 10 200 python -u tools/stress_first_run.py > "$out/run_$i.out" 2> "$out/run_$i.err"
  rc=$?
  d=$(grep -c DIFFERS "$out/run_$i.out")
  if [ $rc -ne 0 ] || [ "$d" != "0" ]; then
    bad=$((bad+1)); echo "run $i rc=$rc differs=$d"; grep DIFFERS "$out/run_$i.out" | cut -c1-160
  else
    rm -f "$out/run_$i.out" "$out/run_$i.err"
  fi
  if [ $rc -ne 0 ]; then tail -n 40 "$out/run_$i.err"; echo "stopped after a crash: $bad bad of $i"; exit 1; fi
done
echo "finished: $bad bad of $n"
