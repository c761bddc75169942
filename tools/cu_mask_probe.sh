#!/bin/bash
# What would a CU-masked tail stream cost and buy?    Every launch of the context is confined to the
# first N CUs (GAUSS_CU_MASK_MAIN=N): (a) the 36-window bench job at 256 / 248 / 240 / 224 CUs -- what the Gram kernel and the
# other stages lose when CUs are set aside; (b) one 8-rank share (the windows of rank 0 of --emulate-world 8) at 256 / 32 / 16 / 8
# CUs -- what the factorisation chain takes on a handful of CUs.
for n in 0 248 240 224; do
  GAUSS_CU_MASK_MAIN=$n python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-i8-variant --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('36 windows, CUs $n (0 = all, two streams):', 'step %.2f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})"
done
for n in 0 32 16 8; do
  GAUSS_CU_MASK_MAIN=$n python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-i8-variant --no-e2e --windows 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('5 windows, CUs $n:', 'step %.2f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})"
done
