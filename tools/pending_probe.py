"""COUNT with a pending match that changes from segment to segment behind a never-forgetting state: a.*b over a
newline-free stream (one a, a b every few bytes), and over lines.  Time and fix-up rounds by size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sregex_amd as S
import harness
ora = harness.OracleEngine()
for pat, head, body in ((rb"a.*b", b"a", b"xxbxyxx"), (rb"a.*b", b"", b"xa xxbxyxxbx\n"), (rb"a[^c]*b", b"a", b"xxbxyxx")):
    with S.Pool() as pool:
        re = S.parse(pool, [pat])
        prog = S.compile(pool, re)
        sc = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, S.ENGINE_AUTO)
        small = head + body * 2000
        want = harness.findall(ora, prog, re.ncaps, small)
        buf = S.DeviceBuffer.from_bytes(small)
        got = sc.scan([buf.ptr], [len(small)])[0]
        buf.free()
        print(pat, body, "engine", sc.engine, "small:", got[:4], "oracle n", len(want) - 1, want[-2][:3] if len(want) > 1 else None, "fixups", sc.last_fixups, flush=True)
        for mib in (1, 4, 16, 64):
            n = (mib << 20) // len(body)
            data = head + body * n
            buf = S.DeviceBuffer.from_bytes(data)
            t0 = time.perf_counter()
            rec = sc.scan([buf.ptr], [len(data)])[0]
            dt = time.perf_counter() - t0
            buf.free()
            print("   %3d MiB: %.3f s, fixups %d, rec %s" % (mib, dt, sc.last_fixups, rec[:4]), flush=True)
            if dt > 3:
                break
