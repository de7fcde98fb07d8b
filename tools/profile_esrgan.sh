#!/bin/bash
# Kernel table of the ESRGAN GAN step (BASELINE configs[3]) per hipGraph replay (tools/prof_replay.py).  GPU box:
#   bash tools/profile_esrgan.sh [out-name]   -> gpurun_out/<out-name> (default r6_esrgan_replay.txt; copy into profiles/)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r6_esrgan_replay.txt}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_esrgan6
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_esrgan6 -- python3 $R/tools/esrgan_step.py 20 > $R/gpurun_out/prof_esrgan6.log 2>&1
python3 $R/tools/prof_replay.py $R/gpurun_out/prof_esrgan6 --top 200 > $O
grep "ms/step" $R/gpurun_out/prof_esrgan6.log >> $O
tail -2 $O
rm -rf $R/gpurun_out/prof_esrgan6
