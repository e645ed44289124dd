#!/bin/bash
# tools/diag_hw_queues.sh TAG PATTERN [env assignments...]: one traced run of tools/diag_hw_queues.py
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; pat=$2; shift 2
out=$R/gpurun_out/hwq/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 "$R/tools/diag_hw_queues.py" "$pat" > "$out/stdout.txt" 2> "$out/stderr.txt"
rc=$?
python3 "$R/tools/diag_hw_queues_report.py" "$out/trace" > "$out/report.txt" 2>&1
echo "$tag rc=$rc"; tail -n 3 "$out/stdout.txt"
exit $rc
