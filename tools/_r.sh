set -e -o pipefail
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], r["achieved"], r["frac"])'
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --genes 20000 --no-cpu-baseline --no-bf16-line --steps 20 --warmup 4 2>/dev/null | python3 -c "$P"
  CGRNA_GEMM_NO_TABLE=1 timeout -k 10 300 python3 bench.py --genes 20000 --no-cpu-baseline --no-bf16-line --steps 20 --warmup 4 2>/dev/null | python3 -c "$P"
done
