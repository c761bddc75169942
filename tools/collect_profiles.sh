#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r01
# 1. --kernel-trace --stats of the default bench.py run  -> gpurun_out/prof_<tag>/stats
# 2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (never combined with trace domains)
# Then tools/summarize_profiles.py copies the summaries into profiles/.
set -e -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 10 --warmup 3 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
echo "stats done" && tail -1 "$OUT/bench_under_rocprof.json" | head -c 600 && echo
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-i8-variant > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.log"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-i8-variant > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.log"
echo "write done"
python3 tools/summarize_profiles.py "$TAG" "$OUT"
