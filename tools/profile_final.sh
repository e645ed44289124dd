# Measurements of the current build, on the GPU box: default bench line, rocprofv3 kernel statistics of the step on
# one stream (float32 and bf16 mode), and the two PMC passes for HBM traffic.  Writes under gpurun_out/.
set -e -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 500 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cd /tmp && export TMPDIR=/tmp
for mode in f32 bf16; do
  rm -rf /tmp/ks_$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$mode -o ks --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-bf16-line --steps 20 --warmup 4 --eager --sequential --dtype $mode > /tmp/ks_$mode.log 2>&1
  python3 $R/tools/kernel_stats_summary.py /tmp/ks_$mode 24 $R/gpurun_out/final_${mode}_sequential_kernel_stats.csv > /dev/null
done
rm -rf /tmp/pf /tmp/pw
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o pf --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-bf16-line --eager > /tmp/pf.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o pw --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-bf16-line --eager > /tmp/pw.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/pf /tmp/pw $R/gpurun_out/final_pmc_f32.json > /dev/null
cat $R/gpurun_out/bench_final.json | cut -c1-400
