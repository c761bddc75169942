#!/bin/bash
# cold (first call of a fresh process) end-to-end time of the chromosome driver under a list of settings
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python3 bench.py --mode e2e --steps 3 --no-from-text 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d.get('end_to_end') or d; c=e['stats_cold_run']
print('$cfg'.ljust(70), 'cold %.1f ms warm %.1f | plan %.1f upload %.1f feeder-wait %.1f job-create %.1f gpu-wait %.1f tables %.1f span %.1f batches %d' % (e['cold_s']*1e3, e['warm_s_median']*1e3, c['t_plan']*1e3, c['t_panel_upload']*1e3, c['t_feeder_wait']*1e3, c['t_job_create']*1e3, c['t_gpu_wait']*1e3, c['t_tables']*1e3, c['gpu_span_ms'], c['n_batches']))"
  done
done
