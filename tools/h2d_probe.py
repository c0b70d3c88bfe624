import ctypes, time, sys
sys.path.insert(0,'/root/repo')
import sregex_amd as S
lib=S.load_library()
hip=ctypes.CDLL('libamdhip64.so')
n=256<<20
src=ctypes.create_string_buffer(n)
ctypes.memset(src,1,n)
d=S.DeviceBuffer(n)
for flags,name in ((0,'default'),(0x80000000,'noncoherent'),(0x4,'writecombined')):
    p=ctypes.c_void_p()
    assert hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(n), ctypes.c_uint(flags))==0
    t0=time.perf_counter(); ctypes.memmove(p, src, n); t1=time.perf_counter()
    print(name,'cpu memcpy into pinned: %.2f GB/s'%(n/(t1-t0)/1e9))
    t0=time.perf_counter(); ctypes.memmove(p, src, n); t1=time.perf_counter()
    print(name,'cpu memcpy into pinned (2nd): %.2f GB/s'%(n/(t1-t0)/1e9))
    hip.hipMemcpy.argtypes=[ctypes.c_void_p,ctypes.c_void_p,ctypes.c_size_t,ctypes.c_int]
    t0=time.perf_counter(); hip.hipMemcpy(d.ptr,p,n,1); t1=time.perf_counter()
    print(name,'H2D from pinned: %.2f GB/s'%(n/(t1-t0)/1e9))
    hip.hipHostFree(p)
t0=time.perf_counter(); hip.hipMemcpy(d.ptr,src,n,1); t1=time.perf_counter()
print('H2D from pageable: %.2f GB/s'%(n/(t1-t0)/1e9))
t0=time.perf_counter(); hip.hipMemcpy(d.ptr,src,n,1); t1=time.perf_counter()
print('H2D from pageable (2nd): %.2f GB/s'%(n/(t1-t0)/1e9))
dst=ctypes.create_string_buffer(n)
t0=time.perf_counter(); ctypes.memmove(dst, src, n); t1=time.perf_counter()
print('plain memcpy: %.2f GB/s'%(n/(t1-t0)/1e9))
t0=time.perf_counter(); ctypes.memmove(dst, src, n); t1=time.perf_counter()
print('plain memcpy (2nd): %.2f GB/s'%(n/(t1-t0)/1e9))
r=hip.hipHostRegister(src, ctypes.c_size_t(n), ctypes.c_uint(0)); t2=time.perf_counter()
t0=time.perf_counter(); r=hip.hipHostRegister(dst, ctypes.c_size_t(n), ctypes.c_uint(0)); t1=time.perf_counter()
print('hipHostRegister 256 MiB: rc',r,'%.2f ms'%((t1-t0)*1e3))
t0=time.perf_counter(); hip.hipMemcpy(d.ptr,dst,n,1); t1=time.perf_counter()
print('H2D from registered: %.2f GB/s'%(n/(t1-t0)/1e9))
