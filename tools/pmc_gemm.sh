# One rocprofv3 counter pass over a single cg_gemm shape.  usage: pmc_gemm.sh TAG "M N K tA tB acc tile splits"
# (run on the GPU box from the repository root; results under gpurun_out/pmc_TAG/)
set -e
TAG=$1; ARGS=$2
OUT=$PWD/gpurun_out/pmc_$TAG
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/gemm_one.py $ARGS 10 > $OUT/run.log 2>&1
