"""What the chip sustains on write-heavy streams (the pack kernel writes 3.3 GB of operand rows and reads 0.9 GB per step):
torch fill_ (pure write), copy_ (1 read : 1 write) and an expand-style 1 : 4 stream (read n bytes, write 4 n), each 3.3 GB written.
    python tools/hbm_write_probe.py > gpurun_out/hbm_write_probe.txt"""
import time
import torch


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = 3_300_000_000 // 16 * 16
    dst = torch.empty(n, dtype=torch.uint8, device="cuda")
    src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 3)
    ms = timed(lambda: dst.fill_(0))
    print("fill_      : %.3f ms  %.2f TB/s written" % (ms, n / ms / 1e9))
    d32, s32 = dst.view(torch.int32), src.view(torch.int32)
    ms = timed(lambda: d32.copy_(s32))
    print("copy_      : %.3f ms  %.2f TB/s moved (read + write)" % (ms, 2 * n / ms / 1e9))
    q = src[: n // 4].view(torch.int32)
    out = dst.view(torch.int32).view(4, -1)
    ms = timed(lambda: out.copy_(q.unsqueeze(0).expand(4, -1)))
    print("1:4 expand : %.3f ms  %.2f TB/s moved (0.25 n read + n written)" % (ms, 1.25 * n / ms / 1e9))


if __name__ == "__main__":
    main()
