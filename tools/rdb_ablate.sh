#!/bin/bash
# What bounds rdb_kernel (round-4 review, item 3): the fused dense block in isolation (tools/bench_rdb.py: 69 dependent launches in a
# hipGraph, batch 16 x 32x32) as shipped and with the developer ablations of rdb.hip (SRX_RDB_ABLATE, results garbage):
#   1 = MFMAs on constant registers, no LDS fragment reads;  2 = every fragment read, no MFMA;  3 = no stage epilogues;
# then PMC passes of the SHIPPED kernel (LDS bank conflicts / LDS-active cycles / MFMA-busy), each in its own rocprofv3 run.
# usage (GPU box): bash tools/rdb_ablate.sh > gpurun_out/rdb_ablate.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for a in 0 1 2 3; do
  echo "== SRX_RDB_ABLATE=$a"
  SRX_ALLOW_GARBAGE_RESULTS=1 SRX_RDB_ABLATE=$a python3 $R/tools/bench_rdb.py 2>&1 | grep -v amdgpu.ids
done
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rdb5_$i -- python3 $R/tools/bench_rdb.py > $R/gpurun_out/pmc_rdb5_$i.log 2>&1
  for k in "rdb_kernel<false" "rdb_kernel<true"; do echo "== pass $i: $k"; python3 $R/tools/pmc_avg.py $R/gpurun_out/pmc_rdb5_$i "$k" || tail -3 $R/gpurun_out/pmc_rdb5_$i.log; done
done
