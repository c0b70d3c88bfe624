"""Time one scan of random subjects per (engine, mode) for the patterns given as hex on the command line
(gpurun_out/fuzz_slow.jsonl: the "re" field)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sregex_amd as S
pats = [bytes.fromhex(h) for h in sys.argv[1:]]
rng = random.Random(7)
datas = [bytes(rng.choice(b"abcx \n_.") for _ in range(n)) for n in (400, 1500, 5000)]
names = {S.ENGINE_SCAN: "scan", S.ENGINE_NFA: "nfa", S.ENGINE_VM: "vm"}
with S.Pool() as pool:
    re = S.parse(pool, pats)
    prog = S.compile(pool, re)
    bufs = [S.DeviceBuffer.from_bytes(d) for d in datas]
    for eng in (S.ENGINE_SCAN, S.ENGINE_NFA, S.ENGINE_VM):
        for mode in (S.HIP_THOMPSON, S.HIP_PIKE_FIRST, S.HIP_PIKE_COUNT):
            t0 = time.perf_counter()
            try:
                sc = S.Scanner(pool, prog, mode, eng)
            except RuntimeError:
                print("%-4s mode %d: not admitted (%.3f s)" % (names[eng], mode, time.perf_counter() - t0), flush=True)
                continue
            t1 = time.perf_counter()
            for b, d in zip(bufs, datas):
                t2 = time.perf_counter()
                r = sc.scan([b.ptr], [len(d)])
                print("%-4s mode %d: create %.3f s, %5d bytes in %.3f s -> %s fixups %d" %
                      (names[eng], mode, t1 - t0, len(d), time.perf_counter() - t2, r[0][:4], sc.last_fixups), flush=True)
