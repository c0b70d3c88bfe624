"""Look for the table-driven scanner's worst case: find-all counts (and first matches) of small programs over
1 GiB of a repeated body, timed per scan; the count of a 64 KiB prefix is checked against the oracle.
Output: gpurun_out/floor_probe.json (profiles/rNN_floor_probe.json)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sregex_amd as S
import harness

CASES = [  # (patterns, body)
    ([rb"\bfoo\b"], b"foo "),               # round 3's floor: a look-ahead match every 4 bytes (POP, nothing outlives)
    ([rb"a+"], b"aaab"),                    # a pending match that grows: an event at every byte of the run
    ([rb"a+"], b"a"),                       # one match that spans the stream
    ([rb"a*"], b"ab"),                      # empty matches between the others
    ([rb"\b"], b"ab cd "),                  # empty matches at every word boundary
    ([rb"$"], b"ab\n"),                     # empty look-ahead matches
    ([rb"a|ab|abc"], b"abcabd"),            # alternation, first arm wins
    ([rb"(a)(b)?"], b"ab a"),               # optional group: the match is pending over one byte
    ([rb"[a-z]+"], b"foo bar "),            # words
    ([rb"\w+\s"], b"foo bar "),
    ([rb"foo$"], b"foo\n"),
    ([rb"^foo"], b"foo\n"),
    ([rb"x*"], b"abc"),                     # an empty match at every byte
    ([rb"(?:a|b)+c"], b"ababab"),           # never matches, list alive throughout
    ([rb"a", rb"ab", rb"c", rb"b"], b"abccc"),
]
N = int(os.environ.get("SRE_FLOOR_BYTES", str(1 << 30)))
rows = []
ora = harness.OracleEngine()
if os.environ.get("SRE_FLOOR_CASES"):
    CASES = [CASES[int(i)] for i in os.environ["SRE_FLOOR_CASES"].split(",")]
for pats, body in CASES:
    n = N // len(body) * len(body)
    host = body * (65536 // len(body))
    with S.Pool() as pool:
        re = S.parse(pool, pats)
        prog = S.compile(pool, re)
        small = S.DeviceBuffer.from_bytes(host)
        big = S.DeviceBuffer.from_bytes(body * (n // len(body))) if n <= (1 << 30) else None
        row = {"re": [p.decode() for p in pats], "body": body.decode(), "bytes": n}
        for mode, key in ((S.HIP_PIKE_COUNT, "count"), (S.HIP_PIKE_FIRST, "first")):
            try:
                sc = S.Scanner(pool, prog, mode, S.ENGINE_AUTO)
            except RuntimeError as e:
                row[key] = "not admitted: %s" % e
                continue
            print("  ", pats, body, key, "small scan", flush=True)
            got = sc.scan([small.ptr], [len(host)])[0]
            allm = harness.findall(ora, prog, re.ncaps, host)
            want_n = len(allm) - 1 if mode == S.HIP_PIKE_COUNT else min(1, len(allm) - 1)
            assert got[1] == want_n, (pats, body, key, got[:4], want_n)
            print("  ", pats, body, key, "oracle ok; big scan", flush=True)
            t0 = time.perf_counter()
            sc.scan([big.ptr], [n])
            first_s = time.perf_counter() - t0
            print("   first big scan %.3f s, fixups %d" % (first_s, sc.last_fixups), flush=True)
            if first_s > 2.0:
                row[key] = {"ms": first_s * 1e3, "GBps": n / first_s / 1e9, "frac": n / first_s / 8e12, "fixups": sc.last_fixups, "note": "one scan only"}
                continue
            t0 = time.perf_counter()
            for _ in range(3):
                rec = sc.scan([big.ptr], [n])[0]
            dt = (time.perf_counter() - t0) / 3
            row[key] = {"ms": dt * 1e3, "GBps": n / dt / 1e9, "frac": n / dt / 8e12, "engine": sc.engine, "kernel": sc.kernel_name,
                        "matches": rec[1], "fixups": sc.last_fixups}
        small.free()
        big.free()
        rows.append(row)
        print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "floor_probe.json"), "w") as f:
    json.dump(rows, f, indent=1)
