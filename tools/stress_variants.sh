#!/bin/bash
# usage: tools/stress_variants.sh <processes per variant>; one line per variant: runs that differ / runs
n=${1:-30}
for v in "{}" "dict(lookahead=False)" "dict(hip_graphs=False)" "dict(hip_graphs=False,lookahead=False)" "dict(concurrent_streams=False)" "NOREUSE"; do
  bad=0; tot=0
  for i in $(seq 1 $n); do
    if [ "$v" = "NOREUSE" ]; then
      out=$(CGRNA_NO_ENCODER_REUSE=1 timeout -k 10 120 python -u tools/stress_first_run.py 2>&1 | grep "^identity True") || { echo "run failed"; exit 1; }
    else
      out=$(STRESS_OVER="$v" timeout -k 10 120 python -u tools/stress_first_run.py 2>&1 | grep "^identity True") || { echo "run failed"; exit 1; }
    fi
    tot=$((tot+1)); case "$out" in *DIFFERS*) bad=$((bad+1)); echo "  $v: ${out:0:200}";; esac
  done
  echo "variant $v: $bad of $tot differ"
done
