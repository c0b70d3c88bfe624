"""PCIe-inclusive rate of the unchanged C API: one sre_vm_pike_exec(buf, len, eof=1)
on a HOST buffer (pageable memory -> one H2D copy + scan), as bench/sregex.c does."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sregex_amd as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else (256 << 20)
data = S.gen_data_host(n, b" a@abc.cc ")
with S.Pool() as pool:
    re = S.parse(pool, [rb"[a-z]+@[a-z]+\.[a-z]+"])
    prog = S.compile(pool, re)
    for rep in range(3):
        with S.Pool() as ep:
            p = S.PikeCtx(ep, prog, re.ncaps)
            t0 = time.perf_counter()
            rc = p.exec(data, True)
            dt = time.perf_counter() - t0
            print("rep", rep, "rc", rc, list(p.ovector[:2]), "%.1f ms  %.2f GB/s" % (dt * 1e3, len(data) / dt / 1e9), flush=True)
