#!/bin/bash
# 8-rank emulation under a list of environment settings: one line per setting with the one-GPU step, the slowest share and the
# shares' Gram stage times (tools/emu_sweep.sh "A=1" "B=2 C=3" ...; through gpurun)
for cfg in "$@"; do
  env $cfg python3 bench.py --steps 10 --warmup 3 --emulate-world 8 --no-cpu-baseline --no-i8-variant --no-e2e --no-parity-spot 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d['emulated_strong_scaling']
print('$cfg'.ljust(44), 'T1 %.3f gram %.3f | slowest %.3f eff %.3f | per rank' % (d['ms_per_step'], d['stage_ms_per_step']['gram'], e['slowest_rank_ms'], e['predicted_efficiency']),
      ' '.join('%.2f/%.2f' % (r['ms_per_step'], r['stage_ms']['gram']) for r in e['per_rank']), 'same_bits', e['pieces_bit_identical_to_one_job'])"
done
