#!/bin/bash
# Round-6 counter passes (each counter set in its own rocprofv3 run, the program directly behind `--`), GPU box:
#   bash tools/pmc_r6.sh srgan   -> the SRGAN step's own launches (tools/srgan_step.py): MFMA-busy, LDS, HBM bytes per kernel and grid
#   bash tools/pmc_r6.sh esrgan  -> the ESRGAN step's (tools/esrgan_step.py)
#   bash tools/pmc_r6.sh infer   -> the shipped bf16-native inference kernels (tools/bench_c64.py trunk / tools/bench_infer.py)
# Output: gpurun_out/pmc6_<what>.txt (copy into profiles/)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$1
cd /tmp && export TMPDIR=/tmp
case $W in
  srgan) PROG="$R/tools/srgan_step.py 2"; KS="" ;;
  esrgan) PROG="$R/tools/esrgan_step.py 2"; KS="" ;;
  infer) PROG="$R/tools/bench_infer.py bf16"; KS="--kernels c64_bf16_kernel,t9_bf16_kernel" ;;
  *) echo "usage: pmc_r6.sh srgan|esrgan|infer"; exit 2 ;;
esac
i=0
DIRS=""
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  D=$R/gpurun_out/pmc6_${W}_$i
  rm -rf $D
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -- python3 $PROG > $D.log 2>&1
  tail -1 $D.log
  DIRS="$DIRS $D"
done
python3 $R/tools/pmc_table.py $DIRS $KS --by-grid --top 24 > $R/gpurun_out/pmc6_$W.txt
head -60 $R/gpurun_out/pmc6_$W.txt
rm -rf $DIRS
