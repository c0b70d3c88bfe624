"""An automaton that never forgets in COUNT mode: quoted strings — whether a lane is inside or outside a string
depends on the parity of the quotes in front of it.  Time and fix-up rounds by stream size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sregex_amd as S
import harness
ora = harness.OracleEngine()
for pat, body in ((rb'"[^"]*"', b'"ab" cde '), (rb'"[^"]*"', b'"abc" "d" e')):
    with S.Pool() as pool:
        re = S.parse(pool, [pat])
        prog = S.compile(pool, re)
        for mode, name in ((S.HIP_PIKE_COUNT, "count"), (S.HIP_PIKE_FIRST, "first")):
            sc = S.Scanner(pool, prog, mode, S.ENGINE_AUTO)
            print(pat, name, "engine", sc.engine, sc.kernel_name, flush=True)
            small = body * 3000
            want = harness.findall(ora, prog, re.ncaps, small)
            buf = S.DeviceBuffer.from_bytes(small)
            got = sc.scan([buf.ptr], [len(small)])[0]
            buf.free()
            print("   small:", got[:4], "oracle n", len(want) - 1, "fixups", sc.last_fixups, flush=True)
            for mib in (1, 4, 16, 64, 256):
                n = (mib << 20) // len(body) * len(body)
                buf = S.DeviceBuffer.from_bytes(body * (n // len(body)))
                t0 = time.perf_counter()
                rec = sc.scan([buf.ptr], [n])[0]
                dt = time.perf_counter() - t0
                buf.free()
                print("   %3d MiB: %.3f s, fixups %d, rec %s" % (mib, dt, sc.last_fixups, rec[:4]), flush=True)
                if dt > 4:
                    break
