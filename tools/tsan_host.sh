#!/bin/bash
# CPU-only ThreadSanitizer pass over the multi-threaded host data layer: preload_lines' worker threads, the
# process-wide index / GWAS / packed-panel caches, and gauss_host_prepare called from several threads at once (what
# the farm and gauss_host_impute_chromosome's feeder thread do).  Builds libgauss_host.so with -fsanitize=thread into
# gauss_amd/lib/tsan/ and runs the CPU feeder tests plus a concurrent-prepare stress on it with 8 threads per call.
# (The GPU pipeline itself cannot run here; its host-side ordering is covered by the gpu tests.)
set -e
cd "$(dirname "$0")/.."
mkdir -p gauss_amd/lib/tsan gpurun_out
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=thread -fno-omit-frame-pointer \
    -o gauss_amd/lib/tsan/libgauss_host.so gauss_amd/csrc/host/host_*.cpp gauss_amd/csrc/host/bgzf_io.cpp \
    gauss_amd/csrc/host/packed_panel.cpp -Lgauss_amd/lib -lgauss_hip -Wl,-rpath,"$(pwd)/gauss_amd/lib" -lz -lpthread -ldl
cat > gpurun_out/tsan_run.py <<PY
import os, sys, tempfile
sys.path.insert(0, "$(pwd)"); sys.path.insert(0, "$(pwd)/tests")
import gauss_amd.api as api
api.HOST_LIB_PATH = "$(pwd)/gauss_amd/lib/tsan/libgauss_host.so"
api.set_host_threads(8)
import pytest
rc = pytest.main(["-x", "-q", "-m", "not gpu", "tests/test_feeder.py", "-p", "no:cacheprovider",
                 "-k", "not concurrent_callers and not pooled_blocks and not reader_fallbacks"])      # those tests start child processes: fork from a multi-threaded process hangs under TSAN (the stress below opens windows from 8 threads)
# concurrent prepares on one text panel and one packed panel (shared caches, 8 caller threads x 8 line threads)
from concurrent.futures import ThreadPoolExecutor
from gauss_amd import panel
d = tempfile.mkdtemp(prefix="gauss_tsan_")
pops = [("AAA", 120, "EUR"), ("BBB", 95, "EUR"), ("CCC", 110, "ASN"), ("DDD", 83, "AFR")]
st = panel.make_synthetic_study(d, pops, n_snp=900, bp_lo=1_000_000, bp_hi=4_000_000, frac_measured=0.3, seed=23)
p = st["paths"]
gpk = os.path.join(d, "panel.gpk")
api.pack_panel(p["index.gz"], p["data.gz"], p["desc.txt"], gpk)
wgt = (["AAA", "CCC", "DDD"], [0.5, 0.3, 0.261])
def prep(k):
    s = 1_000_001 + 250_000 * (k % 12)
    data = gpk if k % 2 else p["data.gz"]
    q = api.Prepared(api.KIND_DISTMIX, chr=22, start_bp=s, end_bp=s + 249_999, wing_size=200_000, pop_wgt_df=wgt,
                     input_file=p["gwas.txt"], reference_index_file=p["index.gz"], reference_data_file=data,
                     reference_pop_desc_file=p["desc.txt"])
    m = (q.M, q.U)
    q.close()
    return m
with ThreadPoolExecutor(max_workers=8) as pool:
    a = list(pool.map(prep, range(48)))
b = [prep(k) for k in range(48)]
assert a == b, "concurrent prepares disagree with serial ones"
print("tsan stress ok:", len(a), "prepares")
sys.exit(rc)
PY
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66" LD_PRELOAD=$(gcc -print-file-name=libtsan.so) python gpurun_out/tsan_run.py
