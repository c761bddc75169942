#!/bin/bash
# experiment harness (GPU box): rebuild k_gram_fp6.hip with extra -D flags, relink, run a short bench, print gram ms
# usage: bash tools/gram6_variants.sh "<flags 1>" "<flags 2>" ...   (GRAM_BENCH_FLAGS / GRAM_ENV for the bench)
set -e
cd "$(dirname "$0")/.."
OBJ=gauss_amd/lib/obj
for flags in "" "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c gauss_amd/csrc/k_gram_fp6.hip -o $OBJ/k_gram_fp6.o -Wno-unused-result -Wno-unused-value $flags
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o gauss_amd/lib/libgauss_hip.so $OBJ/gauss_hip.o $OBJ/k_gram.o $OBJ/k_gram_fp6.o $OBJ/k_pack_epilogue.o $OBJ/k_solve.o $OBJ/k_misc.o
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-i8-variant --no-e2e $GRAM_BENCH_FLAGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('flags [$flags] gram_ms', round(d['stage_ms_per_step']['gram'],3), 'pack', round(d['stage_ms_per_step']['pack_stats'],3), 'TF', round(d['roofline']['achieved'],2), 'step', round(d['ms_per_step'],3))"
done
