"""Pin the CPU restatement (oracle/) against the reference before anything
trusts it:

  (a) every block of the reference's t/ suite, single and forced-multi form,
      whole-buffer and byte-at-a-time ("splitted") modes, Thompson and Pike —
      compared with the lines the reference's own CLI printed
      (tests/golden/t_blocks.jsonl.gz, made by tests/golden/make_goldens.py);
  (b) the explicit expectations stored in the .t files themselves
      (`--- cap`, `--- match_id`, `--- temp_cap`, `--- no_match`);
  (c) reference runs over gen-data streams and find-all iteration traces;
  (d) when oracle/_ref exists (build container, or shipped to the GPU box as a
      binary), live differential runs against the real reference library.
"""
import ctypes
import os
import random

import pytest

import sregex_amd as S
import harness

ENG = harness.OracleEngine()


def test_all_reference_blocks_whole_and_splitted(lib, blocks):
    bad, n = [], 0
    for blk in blocks:
        subject = bytes.fromhex(blk["s"])
        for name, regexes, flags, multi, ref in harness.block_variants(blk):
            if ref["rc"] != 0:
                continue
            with S.Pool() as pool:
                prog = S.compile(pool, S.parse(pool, regexes, flags, multi))
                got = harness.cli_lines(ENG, prog, subject, ref["ncaps"])
            n += 1
            if got != harness.ref_lines(ref):
                bad.append((blk["file"], blk["name"], name, got, harness.ref_lines(ref)))
    assert n == 3832
    assert not bad, bad[:5]


def test_explicit_expectations_in_the_t_files(lib, blocks):
    """`--- cap` / `--- match_id` / `--- temp_cap` / `--- no_match` sections, e.g.
    t/01-sanity-01.t:112-131, t/04-multi.t:9-121, t/01-sanity-05.t:287-344."""
    n = 0
    for blk in blocks:
        if not ({"cap", "match_id", "temp_cap", "no_match"} & set(blk)):
            continue
        _, regexes, flags, multi, ref = harness.block_variants(blk)[0]
        subject = bytes.fromhex(blk["s"])
        with S.Pool() as pool:
            re = S.parse(pool, regexes, flags, multi)
            prog = S.compile(pool, re)
            lines = harness.cli_lines(ENG, prog, subject, re.ncaps)
        pike, spl = lines[2], lines[3]
        if blk.get("no_match"):
            assert pike == "pike no match"
        if "cap" in blk:
            # the harness strips trailing (-1, -1) groups (t/SRegex.pm:401)
            caps = pike.split(" ", 3)[3] if pike.startswith("pike match") else ""
            while caps.endswith(" (-1, -1)"):
                caps = caps[:-len(" (-1, -1)")]
            assert caps == blk["cap"], (blk["file"], blk["name"], pike)
        if "match_id" in blk:
            assert pike.split()[2] == blk["match_id"], (blk["file"], blk["name"], pike)
        if "temp_cap" in blk:
            body = spl[len("splitted pike "):]
            cut = body.find("match") if "match" in body else body.find("no match")
            assert body[:cut].strip() == blk["temp_cap"].strip(), (blk["file"], blk["name"], spl)
        n += 1
    assert n >= 40


def test_gen_data_streams_match_reference(lib):
    for rec in harness.load_jsonl("gen_data.jsonl"):
        pats = [bytes.fromhex(h) for h in rec["re"]]
        data = S.gen_data_host(rec["n"], bytes.fromhex(rec["tail"]))
        assert len(data) == rec["len"]
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, pats))
            t = ENG.thompson(prog)
            assert t.exec(data, True) == rec["thompson"], rec
            t.close()
            p = ENG.pike(prog, rec["ncaps"])
            rc = p.exec(data, True, want_pending=False)
            assert rc == rec["pike_rc"], rec
            if rc >= 0:
                assert list(p.ovector) == rec["pike_ov"], rec
            p.close()


def test_findall_iteration_matches_reference(lib):
    for rec in harness.load_jsonl("findall.jsonl"):
        pats = [bytes.fromhex(h) for h in rec["re"]]
        data = bytes.fromhex(rec["s"])
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, pats))
            assert harness.findall(ENG, prog, rec["ncaps"], data) == rec["matches"], rec["re"]
            n, spans = ENG.count(prog, data, 2 * (rec["ncaps"] + 1), 1024)
            assert n == len(rec["matches"]) - 1
            assert spans == rec["matches"][:-1]
            assert ENG.final_rc == rec["matches"][-1][0]


# ---------------------------------------------------------------- live vs _ref

REF_SO = os.path.join(harness.ORACLE_DIR, "_ref", "libsregex_ref.so")


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_live_differential_against_reference_library(lib):
    """Random subjects over a small alphabet, random chunkings, a pattern zoo
    that leans on assertions, empty matches, nested stars and multi-regex."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_goldens", os.path.join(harness.GOLDEN, "make_goldens.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    ref = mg.RefLib()
    rng = random.Random(20261004)
    zoo = [
        [rb"(a*)*b"], [rb"(a*)+"], [rb"(a|b)*?c"], [rb"\bfoo\b"], [rb"^a|b$"], [rb"a$|ab"],
        [rb"(?:a$|a)b?"], [rb"\Ba\B|c\b"], [rb"(a+)(b+)?"], [rb"x*"], [rb"(|a)+"], [rb"a{2,3}b{0,2}"],
        [rb"\Aab|\n^b"], [rb"[^a]\z|a"], [rb"a", rb"ab", rb"\s+", rb"b$"], [rb"(a)|b", rb"(b)(c)?"],
        [rb"$"], [rb"^"], [rb"\b"], [rb"(a?)*?b"], [rb"((a)|b)+"], [rb"a.c"], [rb"[a-c]+\.[^b]"],
    ]
    alphabet = b"ab c\n.x"
    for pats in zoo:
        rpool, rprog, ncaps = ref.compile(pats)
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, pats))
            for _ in range(60):
                data = bytes(rng.choice(alphabet) for _ in range(rng.randrange(0, 24)))
                assert harness.findall(ENG, prog, ncaps, data) == ref.pike_findall(rprog, ncaps, data), (pats, data)
                t = ENG.thompson(prog)
                assert t.exec(data, True) == ref.thompson(rprog, data), (pats, data)
                t.close()
        ref.L.sre_destroy_pool(rpool)
