"""Many short streams in one call (log lines): time per call by number of streams, with and without a match per line."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sregex_amd as S
import harness
L = 96
lines = {"match": (b"GET /index.html user a@abc.cc " + b"x" * 96)[:L - 1] + b"\n",
         "nomatch": (b"GET /index.html user nobody " + b"x" * 96)[:L - 1] + b"\n"}
ora = harness.OracleEngine()
with S.Pool() as pool:
    re = S.parse(pool, [rb"([a-z]+)@([a-z]+)\.[a-z]+"])
    prog = S.compile(pool, re)
    for mode, name in ((S.HIP_PIKE_FIRST, "first"), (S.HIP_PIKE_COUNT, "count"), (S.HIP_THOMPSON, "thompson")):
        sc = S.Scanner(pool, prog, mode, S.ENGINE_AUTO)
        for kind, line in lines.items():
            want = harness.findall(ora, prog, re.ncaps, line)[0]
            for n in (1000, 10000, 100000, 1000000):
                buf = S.DeviceBuffer.from_bytes(line * n)
                ptrs = [buf.ptr + i * L for i in range(n)]
                lens = [L] * n
                sc.scan(ptrs[:10], lens[:10])
                # the C ABI alone: arrays prepared outside the timed region (the Python binding's list
                # handling costs 0.3-0.6 us per stream by itself)
                import ctypes
                a = (ctypes.c_void_p * n)(*ptrs)
                b = (ctypes.c_size_t * n)(*lens)
                out = (ctypes.c_ssize_t * (n * sc.slots))()
                for _ in range(2):
                    t0 = time.perf_counter()
                    assert sc.lib.sre_hip_scan_enqueue(sc.h, a, b, n, None) == 0
                    assert sc.lib.sre_hip_scan_results(sc.h, out) == 0
                    dt = time.perf_counter() - t0
                s_ = sc.slots
                recs = [list(out[i * s_:(i + 1) * s_]) for i in (0, n // 2, n - 1)]
                ok = all(r[0] == want[0] for r in recs)
                print("%-8s %-7s %8d streams x %d B: %.4f s = %.2f GB/s, %.0f ns/stream, fixups %d, rec0 %s %s" %
                      (name, kind, n, L, dt, n * L / dt / 1e9, dt / n * 1e9, sc.last_fixups, recs[0][:4], "ok" if ok else "MISMATCH"), flush=True)
                buf.free()
