"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one pass each, as the TCC block cannot hold
both) of `bench.py --eager --no-roofline --no-cpu-baseline` into profiles/<tag>_pmc.json.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_f32.json

Per kernel family: launches, mean FETCH_SIZE / WRITE_SIZE (KB) and the corrected HBM bytes per launch.  On gfx950
FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, i.e. half the bytes: it is doubled; WRITE_SIZE is
exact for 16-B-per-lane stores and taken as is.
"""
import collections
import csv
import glob
import json
import sys


def family(name):
    for key in ("gemm_f32", "gemm_bf16", "splitk_reduce", "bn_", "attn_", "adam_"):
        if key in name:
            return key.rstrip("_")
    return "other"


def mean_per_family(d):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        a = agg[family(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in agg.items()}


def main(fetch_dir, write_dir, out):
    fe, wr = mean_per_family(fetch_dir), mean_per_family(write_dir)
    res = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; FETCH_SIZE x2 (gfx950 "
                     "counts half the bytes of wide coalesced reads); values are KB per dispatch",
           "command": "bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --eager", "kernels": {}}
    for k in sorted(set(fe) | set(wr)):
        n, f = fe.get(k, (0, 0.0))
        _, w = wr.get(k, (0, 0.0))
        res["kernels"][k] = dict(launches=n, fetch_kb_raw=f, write_kb=w, hbm_bytes_per_launch=(2 * f + w) * 1024)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
