#!/bin/bash
# gpurun, retried ONLY while the pool answers "no slot free" (exit code 3: nothing ran, nothing was charged).  Any other exit code
# -- success, a failing command, a refusal -- is final.  usage: tools/gpurun_wait.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 12); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 100
done
exit 3
