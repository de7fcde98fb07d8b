#!/bin/bash
# Counter passes of the bf16-native conv on the 1080p trunk shape.  Usage (on the GPU box): tools/pmc_c64.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a -- python3 tools/bench_c64.py trunk > $OUT.a.log 2>&1
python3 tools/pmc_avg.py $OUT/a c64_bf16
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -- python3 tools/bench_c64.py trunk > $OUT.b.log 2>&1
python3 tools/pmc_avg.py $OUT/b c64_bf16
