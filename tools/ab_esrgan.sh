#!/bin/bash
# Same-box A/B of two builds of libsrx_hip.so on the ESRGAN GAN step (config 4): usage tools/ab_esrgan.sh <libA> <libB> [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in $A $B; do
    echo -n "$(basename $L) "; SRX_LIB=$L python tools/esrgan_step.py 20 2>/dev/null | tail -1
  done
done
