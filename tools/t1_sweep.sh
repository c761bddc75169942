#!/bin/bash
# the 36-window headline job under a list of environment settings (through gpurun): step and Gram time per setting
for cfg in "$@"; do
  env $cfg python3 bench.py --steps 20 --warmup 4 --emulate-world 0 --no-cpu-baseline --no-i8-variant --no-e2e --no-parity-spot --no-tails-alone 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); s=d['stage_ms_per_step']
print('$cfg'.ljust(50), 'step %.3f gram %.3f pack %.3f epi %.3f solve %.3f frac %.4f items %d' % (d['ms_per_step'], s['gram'], s['pack_stats'], s['ld_epilogue'], s['solve'], d['roofline']['frac'], d['roofline']['work_items']))"
done
