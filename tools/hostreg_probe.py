#!/usr/bin/env python3
"""How fast can a file mapping be pinned (hipHostRegister) and copied from, against staging it through pinned buffers?
(cold-path question: the chromosome driver's first call stages 846 MB of packed rows at ~35-40 GB/s with eight threads)"""
import ctypes as C, mmap, os, tempfile, time
import numpy as np
import torch
hip = C.CDLL("libamdhip64.so")
n = 846_400_000
d = tempfile.mkdtemp()
path = os.path.join(d, "rows.bin")
np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8).tofile(path)
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for flags, name in ((0, "default"), (0x40, "readonly?")):
    fd = os.open(path, os.O_RDONLY)
    mm = mmap.mmap(fd, n, prot=mmap.PROT_READ)
    addr = C.addressof(C.c_char.from_buffer_copy(b"x"))  # dummy
    buf = np.frombuffer(mm, dtype=np.uint8)
    ptr = buf.ctypes.data
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(C.c_void_p(ptr), C.c_size_t(n), C.c_uint(flags))
    t1 = time.perf_counter()
    print(f"hipHostRegister({name}) rc={rc}: {(t1 - t0) * 1e3:.1f} ms = {n / (t1 - t0) / 1e9:.1f} GB/s")
    if rc == 0:
        t0 = time.perf_counter()
        rc2 = hip.hipMemcpy(C.c_void_p(dev.data_ptr()), C.c_void_p(ptr), C.c_size_t(n), C.c_int(1))
        t1 = time.perf_counter()
        print(f"  hipMemcpy from the registered mapping rc={rc2}: {(t1 - t0) * 1e3:.1f} ms = {n / (t1 - t0) / 1e9:.1f} GB/s")
        ok = bool((dev[:1000].cpu().numpy() == buf[:1000]).all())
        t0 = time.perf_counter(); hip.hipHostUnregister(C.c_void_p(ptr)); print(f"  unregister {(time.perf_counter() - t0) * 1e3:.1f} ms, data ok {ok}")
    del buf
    mm.close(); os.close(fd)
os.remove(path)
