#!/bin/bash
# HIP runtime environment knobs vs the step time (same box): kernel arguments in device memory, graph packet capture.
B="python bench.py --no-cpu-baseline --no-parity --no-roofline"
ms() { python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  echo default $($B | ms)
  echo HIP_FORCE_DEV_KERNARG=1 $(HIP_FORCE_DEV_KERNARG=1 $B | ms)
  echo HIP_FORCE_DEV_KERNARG=0 $(HIP_FORCE_DEV_KERNARG=0 $B | ms)
  echo DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 $(DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 $B | ms)
  echo DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $(DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $B | ms)
done 2>&1 | grep -v amdgpu.ids
