# A/B of two trees on one GPU box in alternating runs: this tree against a copy of another commit's tree built under
# _old_tree/ (git worktree add + build.py; not committed).  Prints ms per iteration (float32, bf16 mode) per run.
set -e
mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for t in new old; do
    if [ $t = new ]; then B=bench.py; else B=_old_tree/bench.py; fi
    timeout -k 10 200 python $B --no-cpu-baseline --no-roofline --steps 60 --warmup 8 > gpurun_out/ab/${t}_$r.json 2> gpurun_out/ab/${t}_$r.err
    python3 -c "
import json;d=json.load(open('gpurun_out/ab/${t}_$r.json'));print('$t', $r, round(d['ms_per_step'],4), round(d['bf16_mfma']['ms_per_step'],4))"
  done
done
