#!/bin/bash
# Work-item granularity of a SMALL job (the 4-5 windows an 8-rank strong-scaling share holds): segment cap, run length per item,
# fine pairs, XCD block.  Prints step / stage times, item count and the Gram kernel's fraction of peak per setting.
for cfg in "X=1" "GAUSS_SEG_MAX=4096" "GAUSS_SEG_MAX=2048" "GAUSS_GROUP_TARGET=1024" "GAUSS_GROUP_TARGET=0" "GAUSS_FINE_PAIRS=4" "GAUSS_FINE_PAIRS=1" \
           "GAUSS_SEG_MAX=4096 GAUSS_FINE_PAIRS=4" "GAUSS_XCD_BLOCK=4" "GAUSS_XCD_BLOCK=16" "GAUSS_GROUP_TARGET=1024 GAUSS_FINE_PAIRS=4"; do
  env $cfg python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-i8-variant --no-e2e --windows ${WINDOWS:-5} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$cfg'.ljust(46), 'step %.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()}, 'items', d['roofline']['work_items'], 'frac %.3f' % d['roofline']['frac'])"
done
