"""Re-run the LAST case of tests/golden/fuzz_regressions.jsonl (engine, mode, segment size as recorded) and
print what came back next to the oracle's answer; SRE_HIP_* knobs apply."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sregex_amd as S
import harness
c = json.loads(open(os.path.join(ROOT, "tests", "golden", "fuzz_regressions.jsonl")).read().strip().split("\n")[-1])
pats = [bytes.fromhex(x) for x in c["re"]]
d = bytes.fromhex(c["s"])
ora = harness.OracleEngine()
with S.Pool() as pool:
    re = S.parse(pool, pats)
    prog = S.compile(pool, re)
    allm = harness.findall(ora, prog, re.ncaps, d)
    sc = S.Scanner(pool, prog, c["mode"], {"vm": S.ENGINE_VM, "nfa": S.ENGINE_NFA}.get(c["engine"], S.ENGINE_SCAN))
    for seg in (c["seg"], 256, 512, 1024):
        if seg:
            sc.set_segment_bytes(seg)
        buf = S.DeviceBuffer.from_bytes(d)
        got = sc.scan([buf.ptr], [len(d)])[0]
        buf.free()
        print("seg", seg, "got", got[:6], "oracle last", allm[-2] if len(allm) > 1 else None, "n", len(allm) - 1, "fixups", sc.last_fixups, flush=True)
