set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-bf16-line --gemm-shapes --steps 20 --warmup 4 > gpurun_out/shapes.json 2> gpurun_out/shapes_f32.txt
tail -60 gpurun_out/shapes_f32.txt
