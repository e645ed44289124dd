# Measurements of the current build, on the GPU box (run from the repository root): rocprofv3
# kernel statistics of the step on one stream (float32 and bf16 mode), and three counter passes per mode (HBM read
# bytes, HBM written bytes, matrix-core counters; separate runs, counters only with --kernel-trace).  Writes under
# gpurun_out/r03/; tools/profile_summary.py turns the passes into profiles/r03_pmc_<mode>.json.
set -e -o pipefail
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in f32 bf16; do
  rm -rf $O/ks_$mode $O/pf_$mode $O/pw_$mode $O/pm_$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ks_$mode -o ks --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-bf16-line --no-api-loop --no-big-shard --steps 20 --warmup 4 --eager --sequential --dtype $mode > $O/ks_$mode.log 2>&1
  echo "stats $mode done"
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf_$mode -o pf --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-bf16-line --no-api-loop --no-big-shard --eager --sequential --dtype $mode > $O/pf_$mode.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw_$mode -o pw --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-bf16-line --no-api-loop --no-big-shard --eager --sequential --dtype $mode > $O/pw_$mode.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pm_$mode -o pm --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-bf16-line --no-api-loop --no-big-shard --eager --sequential --dtype $mode > $O/pm_$mode.log 2>&1
  echo "pmc $mode done"
  python3 $R/tools/kernel_stats_summary.py $O/ks_$mode 24 $O/${mode}_sequential_kernel_stats.csv > /dev/null
  python3 $R/tools/profile_summary.py $O $mode $O/pmc_$mode.json > /dev/null
  # the raw traces are large; the summaries are what travels back
  rm -rf $O/ks_$mode $O/pf_$mode $O/pw_$mode $O/pm_$mode
done
# the bench line last, so that its counter figures (roofline.traffic, mfma_counters) are those of this build: bench.py
# reads them from profiles/r03_pmc_<mode>.json
for mode in f32 bf16; do
  cp $O/pmc_$mode.json $R/profiles/r03_pmc_$mode.json
  cp $O/${mode}_sequential_kernel_stats.csv $R/profiles/r03_${mode}_sequential_kernel_stats.csv
done
cd $R
timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
cut -c1-300 $O/bench.json
