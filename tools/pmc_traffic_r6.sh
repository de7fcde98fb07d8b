#!/bin/bash
# HBM traffic of the Winograd kernel on the headline step's layer shapes: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes.
# GPU box, repo root:  bash tools/pmc_traffic_r6.sh   -> gpurun_out/r06_traffic.json (copy into profiles/)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/traffic6
WL="vgg512 vgg512h vgg256 vgg256h s18432x256x1152 vgg128 s73728x128x1152 s4608x256x4608 s2304x512x2304"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
for W in $WL; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$W.fetch -- python3 $R/tools/pmc_workloads.py $W > $OUT/$W.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$W.write -- python3 $R/tools/pmc_workloads.py $W > $OUT/$W.write.log 2>&1
  echo "done $W"
done
python3 $R/tools/pmc_traffic_r6.py $OUT $WL > $R/gpurun_out/r06_traffic.json
cat $R/gpurun_out/r06_traffic.json | head -40
rm -rf $OUT
