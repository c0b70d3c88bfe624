"""A stream fed to the unchanged sre_vm_pike_exec in chunks from host memory: per-call
times (for a rocprofv3 --kernel-trace --memory-copy-trace --hip-trace breakdown)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sregex_amd as S
total = int(sys.argv[1]) if len(sys.argv) > 1 else (64 << 20)
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 20)
data = S.gen_data_host(total, b" a@abc.cc ")
L = len(data)
buf = ctypes.create_string_buffer(data, L)
with S.Pool() as pool:
    re = S.parse(pool, [rb"[a-z]+@[a-z]+\.[a-z]+"])
    prog = S.compile(pool, re)
    for rep in range(3):
        with S.Pool() as ep:
            ctx = S.PikeCtx(ep, prog, re.ncaps)
            off, rc, calls, ts = 0, S.SRE_AGAIN, 0, []
            t0 = time.perf_counter()
            while rc == S.SRE_AGAIN:
                n = min(chunk, L - off)
                t1 = time.perf_counter()
                rc = ctx.exec(None, off + n >= L, want_pending=False, base=buf, offset=off, length=n)
                ts.append(time.perf_counter() - t1)
                off += n
                calls += 1
            dt = time.perf_counter() - t0
            ts.sort()
            print("rep %d rc %d calls %d  %.2f GB/s  per call: median %.1f us, min %.1f us" %
                  (rep, rc, calls, L / dt / 1e9, ts[len(ts) // 2] * 1e6, ts[0] * 1e6), flush=True)
