#!/bin/bash
# A/B of the headline step on ONE box: the tree in _old/ (an earlier commit, built in place) against this tree, interleaved
for i in 1 2; do
  for d in _old .; do
    (cd $d && python3 bench.py --no-cpu-baseline --no-i8-variant --no-from-text --emulate-world 0 --no-parity-spot --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$d', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['stage_ms_per_step'])")
  done
done
