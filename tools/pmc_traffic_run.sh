#!/bin/bash
# FETCH_SIZE and WRITE_SIZE passes (separate: they do not fit one pass) of several tools/pmc_workloads.py workloads.
# Usage: tools/pmc_traffic_run.sh <outdir> <workload>...   (run on the GPU box, from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; shift
for W in "$@"; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$W.fetch -- python3 tools/pmc_workloads.py $W > $OUT.$W.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$W.write -- python3 tools/pmc_workloads.py $W > $OUT.$W.write.log 2>&1
  echo "== $W"
  python tools/pmc_traffic.py $OUT/$W.fetch $OUT/$W.write "gconv_kernel"
done
