"""Re-run recorded differential-test failures (default: tests/golden/fuzz_regressions.jsonl; written by
tests/test_gpu_parity.py::test_scanner_random_patterns_vs_oracle) on the GPU.
usage: python tools/fuzz_repro.py [--file gpurun_out/fuzz_fail.jsonl] [index ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sregex_amd as S, harness
import importlib.util
spec = importlib.util.spec_from_file_location("tg", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
ora = harness.OracleEngine()
args = sys.argv[1:]
path = os.path.join(ROOT, "tests", "golden", "fuzz_regressions.jsonl")
if args[:1] == ["--file"]:
    path, args = args[1], args[2:]
cases = [json.loads(l) for l in open(path)]
want_idx = [int(a) for a in args] or range(len(cases))
for i in want_idx:
    c = cases[i]
    pats = [bytes.fromhex(x) for x in c["re"]]; d = bytes.fromhex(c["s"])
    with S.Pool() as pool:
        re = S.parse(pool, pats); prog = S.compile(pool, re)
        first, cnt = tg._expect(ora, prog, re.ncaps, d)
        want = first if c["mode"] == S.HIP_PIKE_FIRST else cnt
        eng = S.ENGINE_VM if c["engine"] == "vm" else S.ENGINE_SCAN
        for seg in ([c["seg"]] if c["engine"] == "vm" else [64, 128, 0]):
            sc = S.Scanner(pool, prog, c["mode"], eng)
            if seg and eng == S.ENGINE_SCAN: sc.set_segment_bytes(seg)
            buf = S.DeviceBuffer.from_bytes(d)
            got = sc.scan([buf.ptr], [len(d)])[0]
            buf.free()
            print(i, c["engine"], c["mode"], "seg", seg, "OK" if got == want else "BAD", pats, len(d), got[:6], want[:6],
                  "fix", sc.last_fixups if eng == S.ENGINE_SCAN else "-", flush=True)
