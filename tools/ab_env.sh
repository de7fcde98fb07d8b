#!/bin/bash
# Same-box A/B of one developer switch: usage tools/ab_env.sh SRX_NO_BN_TAIL [rounds]
V=$1; R=${2:-3}
for i in $(seq $R); do
  for v in 1 0; do
    env $V=$v python bench.py --no-cpu-baseline --no-parity --no-roofline --no-other-configs --no-dp-rehearsal | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=$v', d['ms_per_step'])"
  done
done
