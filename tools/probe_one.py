"""One (pattern, body, mode) at a given size, a few scans: for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sregex_amd as S
pat, body, mode, n = sys.argv[1].encode(), sys.argv[2].encode().decode("unicode_escape").encode("latin-1"), int(sys.argv[3]), int(sys.argv[4])
n = n // len(body) * len(body)
with S.Pool() as pool:
    re = S.parse(pool, [pat])
    prog = S.compile(pool, re)
    buf = S.DeviceBuffer.from_bytes(body * (n // len(body)))
    sc = S.Scanner(pool, prog, mode, S.ENGINE_AUTO)
    for _ in range(4):
        rec = sc.scan([buf.ptr], [n])[0]
    print(pat, body, mode, n, rec[:4], "fixups", sc.last_fixups, sc.kernel_name)
