#!/bin/bash
# the cold call's timeline (GAUSS_TRACE=chrom,upload lines of the first chromosome call) under a list of settings,
# measured as the default bench measures it (end_to_end block after the headline run)
for cfg in "$@"; do
  echo "=== $cfg"
  env $cfg GAUSS_TRACE=chrom,upload python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-i8-variant --emulate-world 0 --no-parity-spot --no-from-text --no-tails-alone 2> gpurun_out/_trace.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d['end_to_end']; c=e['stats_cold_run']
print('cold %.1f ms warm %.1f (%.2f x) | plan %.1f upload %.1f feeder-wait %.1f job-create %.1f gpu-wait %.1f tables %.1f span %.1f' % (e['cold_s']*1e3, e['warm_s_median']*1e3, e['cold_s']/e['warm_s_median'], c['t_plan']*1e3, c['t_panel_upload']*1e3, c['t_feeder_wait']*1e3, c['t_job_create']*1e3, c['t_gpu_wait']*1e3, c['t_tables']*1e3, c['gpu_span_ms']))"
  grep -E "^\[(chrom|upload|job)\]" gpurun_out/_trace.err | awk '/data layer of batch 0/ {n++} n<2' 
done
