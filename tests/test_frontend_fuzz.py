"""Front end (parser + compiler) against the LIVE reference CLI on random and
randomly damaged patterns (build container only: needs oracle/_ref/sregex-cli,
the real reference compiled by oracle/Makefile).  Compared: AST dump, capture
count, program dump — or the syntax-error offset."""
import os
import random
import subprocess

import pytest

import sregex_amd as S
import harness

CLI = os.path.join(harness.ROOT, "oracle", "_ref", "sregex-cli")
pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="oracle/_ref/sregex-cli not built")


def _ours(pat, caseless=False):
    with S.Pool() as pool:
        try:
            re = S.parse(pool, [pat], [S.SRE_REGEX_CASELESS] if caseless else None)
        except S.SyntaxError_ as e:
            return {"err": str(e)}
        prog = S.compile(pool, re)
        return {"ast": re.dump().rstrip("\n"), "ncaps": re.ncaps, "prog": prog.dump().rstrip("\n")}


def _theirs(pat, caseless=False):
    p = subprocess.run([CLI] + (["--flags", "i"] if caseless else []) + [pat, b""], capture_output=True, timeout=20)
    if p.returncode < 0:
        return None                      # the reference crashed on this one
    err = p.stderr.decode("latin-1").strip()
    if "[error]" in err:
        return {"err": err.splitlines()[0]}
    if p.returncode != 0 and not p.stdout:
        return None                      # an option-like argument, not a pattern
    out = p.stdout.decode("latin-1").split("\n## ")[0].split("\n")
    return {"ast": out[0], "ncaps": int(out[1].split(":")[1]), "prog": "\n".join(out[2:]).rstrip("\n")}


def test_random_and_damaged_patterns_parse_like_the_reference(monkeypatch):
    monkeypatch.setenv("SRE_FUZZ_WIDE", "1")
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "99")))
    junk = b"()[]{}*+?|\\^$.,-0123456789abx:=!<>"
    bad = []
    n = crashed = rejected = 0
    for _ in range(1200):
        pat = harness.random_regex(rng)
        if rng.random() < 0.4:           # damage it: insert / delete / replace a byte
            b = bytearray(pat)
            for _ in range(rng.randrange(1, 3)):
                k = rng.randrange(0, len(b) + 1)
                op = rng.randrange(3)
                if op == 0 or not b:
                    b.insert(k, rng.choice(junk))
                elif op == 1:
                    del b[min(k, len(b) - 1)]
                else:
                    b[min(k, len(b) - 1)] = rng.choice(junk)
            pat = bytes(b)
        if not pat or b"\0" in pat or pat.startswith(b"-"):
            continue
        caseless = rng.random() < 0.3       # applied at parse time (sre_yyparser.y:244-279, sre_regex.c:170-214)
        try:
            want = _theirs(pat, caseless)
        except subprocess.TimeoutExpired:
            crashed += 1
            continue
        if want is None:
            crashed += 1
            continue
        got = _ours(pat, caseless)
        n += 1
        rejected += "err" in want
        if got != want:
            bad.append((pat, got.get("err") or got.get("ast"), want.get("err") or want.get("ast")))
    assert n > 1000 and rejected > 50, (n, rejected, crashed)
    assert not bad, (len(bad), bad[:5])


def test_oracle_matches_the_live_reference_on_random_patterns():
    """The oracle (oracle/*.c) against the real reference executors, through the
    reference CLI's own call sequence: thompson / splitted thompson / pike /
    splitted pike lines (captures, temp captures, pending matches) on random
    patterns — assertions, lazy quantifiers, counted repeats — and subjects."""
    ora = harness.OracleEngine()
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "99")) + 1000)
    alphabet = b"abcx \n_."
    bad = []
    n = crashed = 0
    for _ in range(500):
        pats = [harness.random_regex(rng) for _ in range(1 if rng.random() < 0.75 else rng.randrange(2, 4))]
        if any(p.startswith(b"-") for p in pats):
            continue
        subject = bytes(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 3, 9, 20, 33])))
        if subject.startswith(b"-"):
            continue
        pat = pats
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            argv = [CLI] + (["-n", str(len(pats))] if len(pats) > 1 else []) + pats + [subject]
            try:
                p = subprocess.run(argv, capture_output=True, timeout=20)
            except subprocess.TimeoutExpired:
                crashed += 1
                continue
            if p.returncode != 0:
                crashed += 1                # the reference crashes on some programs
                continue
            # the six engine lines close the output (the subject itself may hold newlines)
            lines = [l for l in p.stdout.decode("latin-1").split("\n") if l][-6:]
            want = [l for l in lines if "jitted" not in l]
            got = harness.cli_lines(ora, prog, subject, re.ncaps)
            n += 1
            if got != want:
                bad.append((pat, subject, got, want))
    assert n > 400, (n, crashed)
    assert not bad, (len(bad), bad[:3])
