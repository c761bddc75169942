#!/bin/bash
# A/B of the experimental 128 x 64 wave-tile Gram kernel (GAUSS_GRAM_W128=1, k_gram_w128.hip) against the shipped one, interleaved
# on one box; the 16-column edge routine is not in the experiment, so GAUSS_GRAM_EDGE16=0 is the like-for-like baseline
for i in 1 2; do
  for cfg in "X=0" "GAUSS_GRAM_EDGE16=0" "GAUSS_GRAM_W128=1"; do
    env $cfg python3 bench.py --no-cpu-baseline --no-i8-variant --no-from-text --emulate-world 0 --no-parity-spot --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$cfg'.ljust(24), 'step %.3f gram %.3f frac %.4f alone %.3f frac_alone %.4f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('alone_launch_ms',0), r.get('frac_alone',0)))"
  done
done
