#!/bin/bash
# the end_to_end block's cold / warm figures as the DEFAULT bench measures them (context and heap already warm from the headline run)
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-i8-variant --emulate-world 0 --no-parity-spot --no-from-text --no-tails-alone 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d['end_to_end']; c=e['stats_cold_run']
print('$cfg'.ljust(60), 'cold %.1f ms warm %.1f (%.2f x) | plan %.1f upload %.1f feeder-wait %.1f job-create %.1f gpu-wait %.1f tables %.1f span %.1f' % (e['cold_s']*1e3, e['warm_s_median']*1e3, e['cold_s']/e['warm_s_median'], c['t_plan']*1e3, c['t_panel_upload']*1e3, c['t_feeder_wait']*1e3, c['t_job_create']*1e3, c['t_gpu_wait']*1e3, c['t_tables']*1e3, c['gpu_span_ms']))"
  done
done
