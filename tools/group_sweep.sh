#!/bin/bash
# experiment: K length of a Gram work item (GAUSS_GROUP_TARGET, GAUSS_SEG_MAX) x launch-order block vs time, both dtypes
for dt in f32 i8; do
for gt in 4096 2048 1024; do
for sm in 8192 2048; do
for xb in 36 72; do
  echo -n "dtype=$dt group=$gt segmax=$sm xcd=$xb  "
  GAUSS_GROUP_TARGET=$gt GAUSS_SEG_MAX=$sm GAUSS_XCD_BLOCK=$xb python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-i8-variant --gram-dtype $dt 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('step', round(d['ms_per_step'],3), 'gram', round(d['stage_ms_per_step']['gram'],3), 'epi', round(d['stage_ms_per_step']['ld_epilogue'],3), 'items', d['roofline']['work_items'])"
done; done; done; done
