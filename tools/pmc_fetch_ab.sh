#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the GEMM kernels for a given build of the library: tools/pmc_fetch_ab.sh TAG [LIB]
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; lib=$2
O=$R/gpurun_out/pmc_ab/$tag; mkdir -p $O
[ -n "$lib" ] && export CGRNA_LIB=$lib
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace -d $O/$c -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-bf16-line --no-api-loop --no-big-shard --eager --sequential > $O/$c.log 2>&1 || exit 1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{o}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_f32_dma" in r["Kernel_Name"] and r["Counter_Name"] == c:
                t = tot[c]; t[0] += float(r["Counter_Value"]); t[1] += 1
for c, (v, n) in tot.items():
    print(c, "per launch: %.1f KB raw over %d launches" % (v / n, n))
f, w = tot["FETCH_SIZE"], tot["WRITE_SIZE"]
print("fabric bytes per launch (FETCH x 2 + WRITE): %.1f MB" % ((2 * f[0] / f[1] + w[0] / w[1]) * 1024 / 1e6))
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
