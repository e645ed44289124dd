"""Summarise the rocprofv3 passes of tools/profile_r03.sh for one mode (f32 | bf16) into one JSON:

  kernels      per kernel family: launches in the counter run, mean FETCH_SIZE / WRITE_SIZE (KB), corrected HBM bytes
               per launch (FETCH_SIZE x 2: gfx950 counts 64 B per 128-B request of wide coalesced reads; WRITE_SIZE is
               exact for 16-byte-per-lane stores), mean duration in the un-instrumented kernel-trace run, and the
               achieved HBM rate = bytes / duration
  hbm_kernels  the HBM-bound families (Adam, BatchNorm, token, attention, losses / elementwise) with that rate as a
               fraction of the 8 TB/s peak
  mfma         for the MFMA kernels: matrix-core busy share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles),
               kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs), summed over all launches of the run;
               MFMA operations issued; share of wave time parked in s_waitcnt / barriers

    python tools/profile_summary.py gpurun_out/r02 f32 profiles/r02_pmc_f32.json
"""
import collections
import csv
import glob
import json
import sys

HBM_PEAK = 8.0e12


def family(name):
    table = (("gemm_f32_dma", "gemm_f32_dma"), ("gemm_f32", "gemm_f32"), ("gemm_bf16", "gemm_bf16"),
             ("gemm_small", "gemm_small"), ("splitk_reduce", "splitk_reduce"), ("adam_", "adam"), ("bn_", "batchnorm"),
             ("tok_", "token"), ("attn_", "attention"))
    for key, fam in table:
        if key in name:
            return fam
    if "(anonymous namespace)::" in name and "at::native" not in name:
        return "elementwise_and_losses"
    return "torch_rng_copy_fill"


def counters(d):
    """family -> counter -> [sum over dispatches, dispatches]"""
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c = out[family(r["Kernel_Name"])][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    return out


def durations(d):
    """family -> [total ns, calls] from a --kernel-trace --stats run"""
    out = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c = out[family(r["Name"])]
            c[0] += float(r["TotalDurationNs"])
            c[1] += int(r["Calls"])
    return out


def main(root, mode, out_path):
    fe, wr, pm = counters(f"{root}/pf_{mode}"), counters(f"{root}/pw_{mode}"), counters(f"{root}/pm_{mode}")
    du = durations(f"{root}/ks_{mode}")
    res = {"method": "rocprofv3, separate runs: --kernel-trace --stats (durations, 24 iterations) and --pmc FETCH_SIZE / "
                     "--pmc WRITE_SIZE / --pmc <SQ counters> GRBM_GUI_ACTIVE (6 iterations each), all of "
                     "`bench.py --eager --sequential --no-roofline --no-cpu-baseline --no-bf16-line` (one stream, every "
                     "kernel alone on the chip).  FETCH_SIZE x 2 (gfx950 counts half the bytes of wide coalesced "
                     "reads); KB per dispatch.",
           "mode": mode, "kernels": {}, "hbm_kernels": {}, "mfma": {}}
    for fam in sorted(set(fe) | set(wr) | set(du)):
        f, nf = fe.get(fam, {}).get("FETCH_SIZE", [0.0, 0])
        w, nw = wr.get(fam, {}).get("WRITE_SIZE", [0.0, 0])
        ns, calls = du.get(fam, [0.0, 0])
        fkb = f / nf if nf else 0.0
        wkb = w / nw if nw else 0.0
        hbm = (2 * fkb + wkb) * 1024
        avg_us = ns / calls / 1e3 if calls else None
        rate = hbm / (avg_us * 1e-6) if avg_us else None
        res["kernels"][fam] = dict(launches=max(nf, nw), fetch_kb_raw=fkb, write_kb=wkb, hbm_bytes_per_launch=hbm,
                                   avg_us=avg_us, launches_per_iteration=calls / 24.0 if calls else None,
                                   hbm_gb_per_s=rate / 1e9 if rate else None)
        if fam in ("adam", "batchnorm", "token", "attention", "elementwise_and_losses", "splitk_reduce") and rate:
            res["hbm_kernels"][fam] = dict(gb_per_s=rate / 1e9, frac_of_8_tb_per_s=rate / HBM_PEAK, avg_us=avg_us,
                                           mb_per_launch=hbm / 1e6)
    for fam, c in pm.items():
        if not fam.startswith("gemm_f32") and fam != "gemm_bf16":
            continue
        busy, cyc = c["SQ_VALU_MFMA_BUSY_CYCLES"][0], c["GRBM_GUI_ACTIVE"][0] / 8.0
        waves, parked = c["SQ_WAVE_CYCLES"][0], c["SQ_WAIT_ANY"][0]
        res["mfma"][fam] = dict(launches=c["GRBM_GUI_ACTIVE"][1],
                                mfma_busy_share=busy / (1024.0 * cyc) if cyc else None,
                                mfma_mops_f32=c["SQ_INSTS_VALU_MFMA_MOPS_F32"][0],
                                mfma_mops_bf16=c["SQ_INSTS_VALU_MFMA_MOPS_BF16"][0],
                                wave_time_parked_share=parked / waves if waves else None,
                                mean_kernel_cycles=cyc / c["GRBM_GUI_ACTIVE"][1] if c["GRBM_GUI_ACTIVE"][1] else None)
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
