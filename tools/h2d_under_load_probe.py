"""How fast do pinned host -> device copies run while the Gram kernel holds the chip?  (round 3, cold path.)
A 256 MB pinned buffer is copied to the device on a stream of its own, alone and while the 36-window bench job runs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    args = bench.parse_args(["--no-cpu-baseline", "--no-i8-variant", "--no-e2e"])
    from gauss_amd import workload
    rig = bench.Rig(args)
    torch = rig.torch
    ch = workload.make_chromosome(args.snps, args.mode, seed=20260216, sample_scale=1.0)
    wins = workload.windows_of(ch, args.wing, args.windows)
    panel, ld = bench.synth_panel(rig, ch, 20260216)
    store, ld2 = bench.pack_store(rig, ch, panel, ld)
    del panel
    torch.cuda.synchronize()
    runner = bench.Runner(rig, bench.window_descs(ch, wins, store, ld2, args.mode), 1)
    host = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
    dev = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    prio = int(os.environ.get("PROBE_STREAM_PRIORITY", "0"))
    extra = [torch.cuda.Stream() for _ in range(int(os.environ.get("PROBE_EXTRA_STREAMS", "0")))]   # shift the stream -> hardware queue mapping
    s2 = torch.cuda.Stream(priority=prio)
    print("copy stream priority", prio, "after", len(extra), "other streams", flush=True)

    def copy_ms(n=3):
        out = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s2):
                a.record(s2)
                dev.copy_(host, non_blocking=True)
                b.record(s2)
            b.synchronize()
            out.append(a.elapsed_time(b))
        return out

    klib = None
    if os.path.exists("/tmp/libh2dk.so"):
        import ctypes as C
        klib = C.CDLL("/tmp/libh2dk.so")
        klib.h2dk_copy.restype = C.c_float
        klib.h2dk_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    shapes = [(256, 4), (512, 4), (1024, 4), (512, 8), (2048, 1)]

    def kcopy(n_wg, unroll):
        return klib.h2dk_copy(host.data_ptr(), dev.data_ptr(), host.numel(), n_wg, unroll)

    print("alone: 256 MB in", ["%.2f ms (%.1f GB/s)" % (t, 0.268 / t * 1e3) for t in copy_ms()], flush=True)
    if klib:
        for n_wg, un in shapes:
            t = min(kcopy(n_wg, un) for _ in range(3))
            print("copy kernel alone, %d workgroups x unroll %d: %.2f ms (%.1f GB/s)" % (n_wg, un, t, 0.268 / t * 1e3), flush=True)
        assert bool((dev == host.cuda()).all())
    for _ in range(3):
        runner.step()
    res = []
    for _ in range(4):
        runner.step()                      # queues a step (41 ms of GPU work) and returns
        time.sleep(0.004)
        res += copy_ms(1)
    runner.drain()
    print("while the job runs:", ["%.2f ms (%.1f GB/s)" % (t, 0.268 / t * 1e3) for t in res], flush=True)
    if klib:
        for n_wg, un in shapes:
            res = []
            t0 = time.perf_counter()
            for _ in range(4):
                runner.step()
                time.sleep(0.004)
                res.append(kcopy(n_wg, un))
            runner.drain()
            dt = (time.perf_counter() - t0) / 4 * 1e3
            print("copy kernel while the job runs, %d workgroups x unroll %d:" % (n_wg, un), ["%.2f ms (%.1f GB/s)" % (t, 0.268 / t * 1e3) for t in res],
                  "step %.2f ms" % dt, flush=True)
    runner.close()
    rig.close()


if __name__ == "__main__":
    main()
