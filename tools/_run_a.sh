mkdir -p gpurun_out/r3g && cd /root/repo &&
for w in 8 4 2; do timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-e2e --no-i8-variant --no-cpu-baseline --emulate-world $w > gpurun_out/r3g/emu$w.json 2> gpurun_out/r3g/emu$w.err || exit 1; done
echo done
