#!/bin/bash
# The bench command itself under rocprofv3 (program directly behind `--`): the --stats summary of the whole process and, from the same
# trace, the table of the TIMED region only -- tools/prof_replay.py cuts the trace at the Adam ticks and averages the hipGraph replays
# of the timed step (the roofline leg's eager launches, set-up and warm-up drop out).  GPU box:
#   bash tools/profile_bench.sh   -> gpurun_out/r6_bench_stats.txt (copy into profiles/r05_bench_kernel_summary.txt)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/gpurun_out/prof_bench6
cd /tmp && export TMPDIR=/tmp
rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-dp-rehearsal > $R/gpurun_out/r6_bench_prof_line.json 2> $D.err
O=$R/gpurun_out/r6_bench_stats.txt
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-dp-rehearsal   (round 6)"
  echo "# (1) the TIMED region: per hipGraph replay (tools/prof_replay.py --by-grid); durations under the profiler run a few per cent long"
  python3 $R/tools/prof_replay.py $D --by-grid --top 200
  echo
  echo "# (2) rocprofv3's own --stats over the WHOLE process (set-up, eager warm-up, 25 replays, parity step, the roofline leg's eager pass): top 30"
  F=$(ls $D/*/*kernel_stats.csv | head -1)
  head -31 $F | cut -c1-220
  echo
  echo "# bench line of this (profiled) run:"
  tail -1 $R/gpurun_out/r6_bench_prof_line.json | cut -c1-700
} > $O
tail -5 $O | cut -c1-300
rm -rf $D
