#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r01
# 1. --kernel-trace --stats of the default bench.py run (less the end_to_end block)  -> gpurun_out/prof_<tag>/stats
#    (the default run also times the eight emulated shares, the one-stream pass and the int8 variant: <tag>_kernel_stats_by_grid.csv
#    splits every kernel's launches by grid size, so that the headline job's launches stand on a line of their own)
# 2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (never combined with trace domains)
# Then tools/summarize_profiles.py copies the summaries into profiles/.
set -e -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
# every collection gets a directory of its own and none is ever deleted: the one trace that shows an outlier (profiles/README.md:
# a 57 ms Gram launch among 37, rounds 3 and 4) must survive the next collection
OUT=$ROOT/gpurun_out/prof_${TAG}_$(date -u +%Y%m%dT%H%M%SZ)
mkdir -p "$OUT"
(rocm-smi --showclocks --showperflevel --showtemp 2>&1 || true) > "$OUT/smi_before.txt"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 20 --warmup 5 --no-e2e > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
(rocm-smi --showclocks --showperflevel --showtemp 2>&1 || true) > "$OUT/smi_after_trace.txt"
echo "stats done" && tail -1 "$OUT/bench_under_rocprof.json" | head -c 600 && echo
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-i8-variant --no-e2e --emulate-world 0 --no-parity-spot > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.log"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-i8-variant --no-e2e --emulate-world 0 --no-parity-spot > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.log"
echo "write done"
# 3. SQ wave-state + MFMA-utilisation counters (8 SQ slots per pass, own pass, no trace domains)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-i8-variant --no-e2e --emulate-world 0 --no-parity-spot > "$OUT/pmc_sq.json" 2> "$OUT/pmc_sq.log"
echo "sq done"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d "$OUT/pmc_sq2" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-i8-variant --no-e2e --emulate-world 0 --no-parity-spot > "$OUT/pmc_sq2.json" 2> "$OUT/pmc_sq2.log" || echo "sq2 pass failed (optional counters)"
python3 tools/summarize_profiles.py "$TAG" "$OUT"
