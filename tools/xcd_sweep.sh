#!/bin/bash
# experiment: Gram launch order super-block size (GAUSS_XCD_BLOCK) vs step time
set -e
for b in 0 6 36 72 144 288; do
  echo "== GAUSS_XCD_BLOCK=$b"
  GAUSS_XCD_BLOCK=$b python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-i8-variant | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['stage_ms_per_step']['gram'], d['roofline']['achieved'])"
done
