#!/bin/bash
# Three rocprofv3 passes (SQ counters, FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC
# slots") of one tools/pmc_workloads.py workload.  Usage: tools/pmc_run.sh <workload> <outdir>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=$1; OUT=$2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- python3 tools/pmc_workloads.py $W > $OUT.sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 tools/pmc_workloads.py $W > $OUT.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 tools/pmc_workloads.py $W > $OUT.write.log 2>&1
