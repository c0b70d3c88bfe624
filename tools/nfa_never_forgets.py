"""How the NFA tier copes with a program that never forgets: a thread that stays alive from an x far back
(x[^y]*y...) over streams of growing size — time per scan and fix-up rounds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sregex_amd as S
import harness
ora = harness.OracleEngine()
for pat in (rb"x[^y]*y(?:a|b)*a(?:a|b){7}@", rb"x.*y(?:a|b)*a(?:a|b){7}@"):
    with S.Pool() as pool:
        re = S.parse(pool, [pat])
        prog = S.compile(pool, re)
        for mode, name in ((S.HIP_PIKE_FIRST, "first"), (S.HIP_THOMPSON, "thompson")):
            sc = S.Scanner(pool, prog, mode, S.ENGINE_AUTO)
            print(pat, name, "engine", sc.engine, sc.kernel_name, flush=True)
            for mib in (1, 4, 16, 64):
                n = mib << 20
                body = b"x" + b"abccc" * ((n - 40) // 5)
                data = body + b" y abaabaabab@ "
                if mib == 1:
                    small = b"x" + b"abccc" * 2000 + b" y abaabaabab@ "
                    want = harness.findall(ora, prog, re.ncaps, small)[0]
                    buf = S.DeviceBuffer.from_bytes(small)
                    got = sc.scan([buf.ptr], [len(small)])[0]
                    buf.free()
                    print("   small:", got[:4], "oracle", want[:3], flush=True)
                buf = S.DeviceBuffer.from_bytes(data)
                t0 = time.perf_counter()
                rec = sc.scan([buf.ptr], [len(data)])[0]
                dt = time.perf_counter() - t0
                buf.free()
                print("   %3d MiB: %.3f s, fixups %d, rec %s" % (mib, dt, sc.last_fixups, rec[:4]), flush=True)
                if dt > 5:
                    break
