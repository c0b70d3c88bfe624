"""Host -> device copy of one chunk, per strategy and size (median of 30, microseconds):
   sync      hipMemcpy from the caller's pageable buffer
   async     hipMemcpyAsync from it + hipStreamSynchronize
   staged    memcpy into a pinned non-coherent buffer + hipMemcpyAsync + synchronize
   pinned    hipMemcpyAsync from the pinned buffer alone (the DMA floor)"""
import ctypes, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sregex_amd as S
S.load_library()
hip = ctypes.CDLL('libamdhip64.so')
vp, sz = ctypes.c_void_p, ctypes.c_size_t
hip.hipMemcpy.argtypes = [vp, vp, sz, ctypes.c_int]
hip.hipMemcpyAsync.argtypes = [vp, vp, sz, ctypes.c_int, vp]
hip.hipStreamSynchronize.argtypes = [vp]
hip.hipHostMalloc.argtypes = [ctypes.POINTER(vp), sz, ctypes.c_uint]
stream = vp()
assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
big = 64 << 20
src = ctypes.create_string_buffer(big)
ctypes.memset(src, 1, big)
d = S.DeviceBuffer(big)
pin = vp()
assert hip.hipHostMalloc(ctypes.byref(pin), big, 0x80000000) == 0
def med(f, n, reps=30):
    ts = []
    for i in range(reps):
        off = (i * n) % (big - n + 1) if n < big else 0        # a different part of the buffer each time
        t0 = time.perf_counter(); f(off, n); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6
base = ctypes.addressof(src)
def sync(off, n): hip.hipMemcpy(d.ptr, base + off, n, 1)
def asyn(off, n): hip.hipMemcpyAsync(d.ptr, base + off, n, 1, stream); hip.hipStreamSynchronize(stream)
def staged(off, n): ctypes.memmove(pin, base + off, n); hip.hipMemcpyAsync(d.ptr, pin, n, 1, stream); hip.hipStreamSynchronize(stream)
def pinned(off, n): hip.hipMemcpyAsync(d.ptr, pin, n, 1, stream); hip.hipStreamSynchronize(stream)
print("%10s %10s %10s %10s %10s   (us; GB/s of the best)" % ("bytes", "sync", "async", "staged", "pinned"))
for n in (4096, 65536, 262144, 1 << 20, 4 << 20, 16 << 20, 64 << 20):
    r = [med(f, n) for f in (sync, asyn, staged, pinned)]
    print("%10d %10.1f %10.1f %10.1f %10.1f   %.1f" % (n, r[0], r[1], r[2], r[3], n / min(r[:3]) / 1e3), flush=True)
