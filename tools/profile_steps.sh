#!/bin/bash
# Kernel tables of the two train steps, reduced PER hipGraph REPLAY (tools/prof_replay.py: the trace is cut at the Adam ticks and only
# the replays of the timed step are averaged -- set-up, eager warm-up and any other workload drop out).  GPU box:
#   bash tools/profile_steps.sh      -> gpurun_out/r5_srgan_replay.txt, gpurun_out/r5_esrgan_replay.txt (copy into profiles/)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_srgan5 $R/gpurun_out/prof_esrgan5
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_srgan5 -- python3 $R/tools/srgan_step.py 30 > $R/gpurun_out/prof_srgan5.log 2>&1
python3 $R/tools/prof_replay.py $R/gpurun_out/prof_srgan5 --by-grid --top 200 > $R/gpurun_out/r5_srgan_replay.txt
tail -3 $R/gpurun_out/r5_srgan_replay.txt; grep "ms/step" $R/gpurun_out/prof_srgan5.log
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_esrgan5 -- python3 $R/tools/esrgan_step.py 20 > $R/gpurun_out/prof_esrgan5.log 2>&1
python3 $R/tools/prof_replay.py $R/gpurun_out/prof_esrgan5 --top 200 > $R/gpurun_out/r5_esrgan_replay.txt
tail -3 $R/gpurun_out/r5_esrgan_replay.txt; grep "ms/step" $R/gpurun_out/prof_esrgan5.log
# the raw traces are large: keep the reduced tables only
rm -rf $R/gpurun_out/prof_srgan5 $R/gpurun_out/prof_esrgan5
