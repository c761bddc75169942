import sys, time, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gauss_amd import hotpath, _lib
t0=time.perf_counter(); ctx = hotpath.Context(0); print('ctx %.1f ms' % ((time.perf_counter()-t0)*1e3))
lib = ctx.lib
lib.gauss_pinned_alloc.argtypes=[C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
for sz in (32<<20, 32<<20, 8<<20, 128<<20):
    p=C.c_void_p(); t0=time.perf_counter(); rc=lib.gauss_pinned_alloc(ctx.handle, sz, C.byref(p)); print('pinned %d MB: %.2f ms rc %d' % (sz>>20, (time.perf_counter()-t0)*1e3, rc))
for sz in (900<<20, 900<<20):
    p=C.c_void_p(); t0=time.perf_counter(); rc=lib.gauss_store_alloc(ctx.handle, sz, C.byref(p)); print('store_alloc %d MB: %.2f ms rc %d' % (sz>>20, (time.perf_counter()-t0)*1e3, rc))
