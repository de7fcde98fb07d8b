#!/bin/bash
# Round 4: HBM traffic of every GEMM shape of the headline step's dominant kernel (two --pmc passes per shape).
# Usage (GPU box, repo root): tools/pmc_traffic_all.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; mkdir -p $OUT
for W in s18432x256x1152 s73728x128x1152 s36864x256x576 s9216x128x2304 s36864x128x1152 s4608x256x4608 s36864x128x576 s4608x256x2304 s2304x512x2304; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$W.fetch -- python3 tools/pmc_workloads.py $W 10 > $OUT/$W.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$W.write -- python3 tools/pmc_workloads.py $W 10 > $OUT/$W.write.log 2>&1
  echo "== $W"
  python3 tools/pmc_traffic.py $OUT/$W.fetch $OUT/$W.write "gconv_kernel" | tr '\n' ' '
  grep -h "gconv_kernel" $OUT/$W.fetch/*/*kernel_trace.csv | head -1 | cut -c1-200
  echo
done
