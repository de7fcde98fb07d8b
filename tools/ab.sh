#!/bin/bash
# Same-box A/B of two builds of libsrx_hip.so (devices differ by a few per cent: never compare runs from two gpurun calls).
# usage: tools/ab.sh <libA> <libB> [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    SRX_LIB=$L python bench.py --no-cpu-baseline --no-parity --no-roofline --no-other-configs --no-dp-rehearsal | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], d['ms_per_step'])"
  done
done
