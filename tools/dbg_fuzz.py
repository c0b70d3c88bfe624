import sys, random
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sregex_amd as S, harness
import importlib.util
spec = importlib.util.spec_from_file_location("tg", "tests/test_gpu_parity.py"); tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
ora = harness.OracleEngine()
seg0 = 64
rng = random.Random(20261004 + seg0)
alphabet = b"abcx \n_."
target = rb'(?:([^a]{0,2}?){1,3}|c{1,3}[a-c]?)*(' + b'\\n' + rb'b{2}.+?)'
for _ in range(600):
    nre = 1 if rng.random() < 0.8 else rng.randrange(2, 4)
    pats = [harness.random_regex(rng) for _ in range(nre)]
    with S.Pool() as pool:
        re = S.parse(pool, pats); prog = S.compile(pool, re)
        ok = {}
        for mode in (S.HIP_THOMPSON, S.HIP_PIKE_FIRST, S.HIP_PIKE_COUNT):
            try:
                S.Scanner(pool, prog, mode, S.ENGINE_SCAN); ok[mode] = 1
            except RuntimeError:
                pass
        if not ok:
            continue
        datas = [bytes(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 7, 64, 65, 130, 400]))) for _ in range(6)]
        if pats != [target]:
            continue
        print("found", pats)
        for d in datas:
            first, cnt = tg._expect(ora, prog, re.ncaps, d)
            for seg in (64, 128, 0):
                sc = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, S.ENGINE_SCAN)
                if seg: sc.set_segment_bytes(seg)
                buf = S.DeviceBuffer.from_bytes(d)
                got = sc.scan([buf.ptr], [len(d)])[0]
                buf.free()
                if got != cnt:
                    print("seg", seg, "len", len(d), "got", got, "want", cnt, "fix", sc.last_fixups)
                    allm = harness.findall(ora, prog, re.ncaps, d)
                    print(" matches", [m[:3] for m in allm])
                    print(" data", d)
        break
