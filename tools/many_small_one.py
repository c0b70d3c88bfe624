"""one many-streams case for rocprofv3: MODE N [match]"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sregex_amd as S
mode, n = int(sys.argv[1]), int(sys.argv[2])
L = 96
line = ((b"GET /index.html user a@abc.cc " if len(sys.argv) > 3 else b"GET /index.html user nobody ") + b"x" * 96)[:L - 1] + b"\n"
with S.Pool() as pool:
    prog = S.compile(pool, S.parse(pool, [rb"([a-z]+)@([a-z]+)\.[a-z]+"]))
    sc = S.Scanner(pool, prog, mode, S.ENGINE_AUTO)
    buf = S.DeviceBuffer.from_bytes(line * n)
    a = (ctypes.c_void_p * n)(*[buf.ptr + i * L for i in range(n)])
    b = (ctypes.c_size_t * n)(*([L] * n))
    out = (ctypes.c_ssize_t * (n * sc.slots))()
    for _ in range(4):
        t0 = time.perf_counter()
        assert sc.lib.sre_hip_scan_enqueue(sc.h, a, b, n, None) == 0
        t1 = time.perf_counter()
        assert sc.lib.sre_hip_scan_results(sc.h, out) == 0
        t2 = time.perf_counter()
        print("enqueue %.3f ms, results %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
