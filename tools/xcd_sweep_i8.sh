#!/bin/bash
# experiment: Gram launch order super-block size vs the int8 Gram kernel (operand-delivery bound)
for b in 0 12 36 72 144 288 576; do
  echo -n "GAUSS_XCD_BLOCK=$b  "
  GAUSS_XCD_BLOCK=$b python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --gram-dtype i8 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('step', round(d['ms_per_step'],3), 'gram', round(d['stage_ms_per_step']['gram'],3))"
done
