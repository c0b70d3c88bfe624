"""GPU parity: the HIP path, called through the C ABI, against the reference's
golden vectors and against the oracle on the same inputs.  Bit-exact: rc,
regex id, every ovector slot, temp/pending captures in streaming mode.
"""
import ctypes
import os

import pytest

import sregex_amd as S
import harness

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(lib):
    assert lib.sre_hip_device_count() >= 1, "no HIP device: the product has no CPU path"
    return lib


def test_compat_api_all_reference_blocks(gpu, blocks):
    """sre_vm_thompson_exec / sre_vm_pike_exec on the GPU, replaying the reference
    CLI's whole-buffer and byte-at-a-time call sequences for every t/ block
    (single and forced-multi forms): the printed lines must equal the
    reference CLI's."""
    eng = harness.ProductEngine()
    bad, n = [], 0
    for blk in blocks:
        subject = bytes.fromhex(blk["s"])
        for name, regexes, flags, multi, ref in harness.block_variants(blk):
            if ref["rc"] != 0:
                continue
            with S.Pool() as pool:
                prog = S.compile(pool, S.parse(pool, regexes, flags, multi))
                got = harness.cli_lines(eng, prog, subject, ref["ncaps"])
                eng.recycle()
            n += 1
            if got != harness.ref_lines(ref):
                bad.append((blk["file"], blk["name"], name, got, harness.ref_lines(ref)))
    assert n == 3832
    assert not bad, (len(bad), bad[:5])


def test_findall_iteration_on_one_context(gpu):
    eng = harness.ProductEngine()
    for rec in harness.load_jsonl("findall.jsonl"):
        pats = [bytes.fromhex(h) for h in rec["re"]]
        data = bytes.fromhex(rec["s"])
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, pats))
            assert harness.findall(eng, prog, rec["ncaps"], data) == rec["matches"], rec["re"]
            eng.recycle()


def _expect(ora, prog, ncaps, data):
    """(first-match record, count record) the batched API must return."""
    nov = 2 * (ncaps + 1)
    allm = harness.findall(ora, prog, ncaps, data)
    final = allm[-1][0]
    matches = allm[:-1]
    first = [matches[0][0], 1] + matches[0][1:] if matches else [S.SRE_DECLINED, 0] + [-1] * nov
    if matches:
        cnt = [S.SRE_ERROR if final == S.SRE_ERROR else matches[-1][0], len(matches)] + matches[-1][1:]
    else:
        cnt = [final, 0] + [-1] * nov
    return first, cnt


@pytest.mark.parametrize("engine", [S.ENGINE_VM, S.ENGINE_AUTO])
def test_batched_scan_gen_data_goldens(gpu, engine):
    """Device-resident batched API on gen-data streams vs reference results."""
    recs = harness.load_jsonl("gen_data.jsonl")
    if engine == S.ENGINE_VM:
        # the exact VM walks one stream per lane at MB/s: the small sizes pin it, the
        # 5 MiB + 8 (bench/gen-data.pl:9) and 16 MiB rows are for the throughput engines
        recs = [r for r in recs if r["n"] <= (1 << 20) + 8]
    by_cfg = {}
    for r in recs:
        by_cfg.setdefault(tuple(r["re"]), []).append(r)
    for pats_hex, rs in by_cfg.items():
        pats = [bytes.fromhex(h) for h in pats_hex]
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, pats))
            datas = [S.gen_data_host(r["n"], bytes.fromhex(r["tail"])) for r in rs]
            bufs = [S.DeviceBuffer.from_bytes(d) for d in datas]
            ptrs, lens = [b.ptr for b in bufs], [len(d) for d in datas]
            th = S.Scanner(pool, prog, S.HIP_THOMPSON, engine).scan(ptrs, lens)
            pk = S.Scanner(pool, prog, S.HIP_PIKE_FIRST, engine).scan(ptrs, lens)
            for r, t, p in zip(rs, th, pk):
                assert t[0] == r["thompson"], (r["cfg"], r["n"], t)
                assert p[0] == r["pike_rc"], (r["cfg"], r["n"], p)
                if r["pike_rc"] >= 0:
                    assert p[2:] == r["pike_ov"], (r["cfg"], r["n"], p)
            for b in bufs:
                b.free()


@pytest.mark.parametrize("engine", [S.ENGINE_VM, S.ENGINE_AUTO])
def test_batched_count_vs_oracle(gpu, engine):
    """PIKE_COUNT = the find-all iteration done on device; count, last regex id
    and last ovector must equal the oracle's iteration."""
    ora = harness.OracleEngine()
    cases = []
    for rec in harness.load_jsonl("findall.jsonl"):
        cases.append(([bytes.fromhex(h) for h in rec["re"]], bytes.fromhex(rec["s"])))
    cfg3 = [b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"]
    cases.append((cfg3, S.gen_data_host(20000, b"aaabbccb")))
    cases.append(([rb"[a-z]+@[a-z]+\.[a-z]+"], S.gen_data_host(30000, b"@abc.cc ") * 3))
    for pats, data in cases:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            first, cnt = _expect(ora, prog, re.ncaps, data)
            buf = S.DeviceBuffer.from_bytes(data)
            rec = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, engine).scan([buf.ptr], [len(data)])[0]
            buf.free()
            assert rec == cnt, (pats, rec, cnt)


@pytest.mark.parametrize("horizon", [0, 512])
def test_count_on_the_nfa_tier_vs_oracle(gpu, horizon, monkeypatch):
    """Find-all counting of programs the step automaton declines (ENGINE_NFA forced): rounds of
    first-match searches — set kernel to the next MATCH event, exact VM over its window, next search
    from the match's end (sre_vm_pike.c:179-196, :586-636).  Count, last regex id and last ovector
    equal the oracle's iteration; a tiny horizon forces the rounds that move a search's buffer to its
    last clean position and the ones that must let the horizon grow; several streams per batch."""
    import random
    if horizon:
        monkeypatch.setenv("SRE_HIP_COUNT_HORIZON", str(horizon))
    ora = harness.OracleEngine()
    rng = random.Random(2025 + horizon)
    zoo = [[rb"(?:a|b)*a(?:a|b){7}@"], [rb"(a|b)*a(a|b){5}(c)"], [rb"[ab]{3,9}c{2}(x)?"], [rb"x.{0,10}y"],
           [rb"(a|ab|abc){2,6}x"], [rb"a[^x]{20}x"], [rb"(\w+ ){3}(\w+)"], [rb"\Aa.{3}b", rb"\nc{2,4}"],
           [rb"[a-z]+@[a-z]+\.[a-z]+"], [rb"(a+)(b+)?"], [rb"(?:a.*b|a)"], [rb"\Ab+|xb"], [rb"(\n|a)b"],
           [b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"]]
    alphabets = [b"abc", b"ab c\n.x@:/?y,d", b"aaaaab", b"ab", b"abcx@ \n"]
    with S.Pool() as pool:
        # a re-armed search that does not start behind a newline skips over newlines (the leading-byte
        # skip fires on its bare initial list): programs whose seeded closure depends on ^ keep the VM
        for pat in (rb"^b+", rb"(^|a)b"):
            with pytest.raises(RuntimeError):
                S.Scanner(pool, S.compile(pool, S.parse(pool, [pat])), S.HIP_PIKE_COUNT, S.ENGINE_NFA)
    for pats in zoo:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            sc = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, S.ENGINE_NFA)
            assert sc.engine == S.ENGINE_NFA
            datas = []
            for i in range(7):
                alpha = alphabets[i % len(alphabets)]
                n = rng.choice([0, 1, 65, 300, 2000, 9000])
                datas.append(bytes(rng.choice(alpha) for _ in range(n)))
            # sparse matches far apart, and none at all
            datas.append((S.gen_data_host(5000, b" abaabaabab@ abccc") * 3)[:14000])
            datas.append(b"x" * 3000 + b"abababab@" + b"x" * 2500 + b"\nb" + b"y" * 900)
            datas.append(b"z" * 6000)
            bufs = [S.DeviceBuffer.from_bytes(d) for d in datas]
            recs = sc.scan([b.ptr for b in bufs], [len(d) for d in datas])
            for d, rec in zip(datas, recs):
                _, cnt = _expect(ora, prog, re.ncaps, d)
                assert rec == cnt, (pats, horizon, d[:60], len(d), rec, cnt)
            assert sc.last_count_rounds >= 1
            for b in bufs:
                b.free()


def test_count_with_lookahead_assertions_on_the_scanner(gpu):
    """Find-all counting of a program with $ \\z \\b \\B on the table-driven scanner
    (ENGINE_SCAN is forced: a decline raises).  A re-armed search starts from the initial
    list of its context's seen_newline / seen_word (SRE_DFA_INIT_RESTART_NL / _WORD); a match
    that ends in front of a byte by look-ahead (foo$) is followed by a search that reads
    that byte again; an empty one skips a byte (sre_vm_pike.c:179-196, 586-601)."""
    import random
    ora = harness.OracleEngine()
    rng = random.Random(77)
    zoo = [[rb"\b(\w+)\b"], [rb"a$"], [rb"\bfoo\b"], [rb"\B"], [rb"\b"], [rb"$"], [rb"(a+)\b(?:\s|$)"], [rb"x*\b"],
           [rb"(\w)\B(\w)"], [rb"^(\w+)$"], [rb"(?:$|a)(b|\b)"], [rb"[a-c]+$"], [rb"\Ba\B"], [rb"(\s*)\b([a-c]+)\B"]]
    alphabets = [b"ab c\n_x.", b"foo \nab", b"aaa\n "]
    n = 0
    for pats in zoo:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            for seg in (0, 64, 256):
                sc = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, S.ENGINE_SCAN)
                if seg:
                    sc.set_segment_bytes(seg)
                datas = [bytes(rng.choice(a) for _ in range(size)) for a in alphabets for size in (0, 1, 77, 3000, 20000)]
                bufs = [S.DeviceBuffer.from_bytes(d) for d in datas]
                got = sc.scan([b.ptr for b in bufs], [len(d) for d in datas])
                for d, g in zip(datas, got):
                    _, cnt = _expect(ora, prog, re.ncaps, d)
                    assert g == cnt, (pats, seg, len(d), g, cnt)
                    n += 1
                for b in bufs:
                    b.free()
    assert n == len(zoo) * 3 * 15


def test_many_ragged_streams_one_call(gpu):
    """Independent streams of different lengths (incl. empty) in one batch."""
    ora = harness.OracleEngine()
    pats = [rb"([a-z]+)@([a-z]+)\.[a-z]+"]
    tails = [b"", b"x", b"@abc.cc ", b" a@b.c", b"@@..", b"q@w.e!"]
    datas = [S.gen_data_host(n, tails[i % len(tails)]) for i, n in
             enumerate([0, 1, 5, 64, 65, 1000, 4096, 4097, 10000, 33333] * 7)]
    with S.Pool() as pool:
        re = S.parse(pool, pats)
        prog = S.compile(pool, re)
        bufs = [S.DeviceBuffer.from_bytes(d) for d in datas]
        got = S.Scanner(pool, prog, S.HIP_PIKE_FIRST).scan([b.ptr for b in bufs], [len(d) for d in datas])
        for d, g in zip(datas, got):
            p = ora.pike(prog, re.ncaps)
            rc = p.exec(d, True, want_pending=False)
            assert g[0] == rc, (d[-12:], g)
            if rc >= 0:
                assert g[2:] == list(p.ovector), (d[-12:], g, list(p.ovector))
            p.close()
        for b in bufs:
            b.free()


def test_thousands_of_short_streams_one_call(gpu):
    """A batch of log lines: 6000 short streams (0-300 bytes) in one call — at 4096 streams and more the capture
    walker takes the streams by LANES (64 walks per wave side by side; one walking lane per workgroup made 8.8 ms
    of a million lines with a match each, tools/many_small_probe.py).  First match + captures and find-all
    counts of every line against the oracle."""
    import random
    ora = harness.OracleEngine()
    rng = random.Random(5)
    words = [b"GET ", b"/index.html ", b"user ", b"a@abc.cc ", b"x@y.zz ", b"nobody ", b"@@ ", b"q@w ", b"\n", b"[abc] "]
    lines = [b"".join(rng.choice(words) for _ in range(rng.randrange(0, 24))) for _ in range(6000)]
    blob = b"".join(lines)
    offs, o = [], 0
    for ln in lines:
        offs.append(o)
        o += len(ln)
    with S.Pool() as pool:
        re = S.parse(pool, [rb"([a-z]+)@([a-z]+)\.[a-z]+", rb"\[(\w+)\]"])
        prog = S.compile(pool, re)
        buf = S.DeviceBuffer.from_bytes(blob)
        ptrs, lens = [buf.ptr + x for x in offs], [len(ln) for ln in lines]
        first = S.Scanner(pool, prog, S.HIP_PIKE_FIRST).scan(ptrs, lens)
        count = S.Scanner(pool, prog, S.HIP_PIKE_COUNT).scan(ptrs, lens)
        buf.free()
        cache = {}
        for ln, f, c in zip(lines, first, count):
            if ln not in cache:
                cache[ln] = _expect(ora, prog, re.ncaps, ln)
            wf, wc = cache[ln]
            assert f == wf and c == wc, (ln, f, wf, c, wc)


def test_gen_data_kernel_matches_host_generator(gpu):
    for n, tail in [(0, b""), (1, b""), (5, b""), (4098, b"aaabbccb"), (100003, b"@abc.cc "), (77, b"x" * 77)]:
        n = S.gen_data_length(n, len(tail)) if n >= len(tail) else len(tail)
        buf = S.DeviceBuffer(max(n, 1))
        assert gpu.sre_hip_gen_data(buf.ptr, n, tail, len(tail), None) == 0
        assert buf.to_bytes(n) == S.gen_data_host(n, tail)
        buf.free()


# ------------------------------------------------------------ table-driven scanner

def test_scanner_all_admitted_reference_blocks(gpu, blocks):
    """Every block of the reference suite whose automaton the scanner admits, through the scanner
    engine (device-resident, batched API): first match + captures, and
    Thompson's yes/no, against the reference CLI's lines."""
    bad, n = [], 0
    for blk in blocks:
        subject = bytes.fromhex(blk["s"])
        _, regexes, flags, multi, ref = harness.block_variants(blk)[0]
        if ref["rc"] != 0:
            continue
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, regexes, flags, multi))
            try:
                sc = S.Scanner(pool, prog, S.HIP_PIKE_FIRST, S.ENGINE_AUTO)
            except RuntimeError:
                continue
            if sc.engine != S.ENGINE_SCAN:
                continue
            buf = S.DeviceBuffer.from_bytes(subject)
            rec = sc.scan([buf.ptr], [len(subject)])[0]
            th = S.Scanner(pool, prog, S.HIP_THOMPSON, S.ENGINE_SCAN).scan([buf.ptr], [len(subject)])[0]
            buf.free()
            nov = 2 * (ref["ncaps"] + 1)
            line = ("pike match %d%s" % (rec[0], harness._fmt_caps(rec[2:], nov)) if rec[0] >= 0
                    else "pike no match")
            tl = "thompson " + ("match" if th[0] == 0 else "no match")
            n += 1
            if line != ref["res"][4] or tl != ref["res"][0]:
                bad.append((blk["file"], blk["name"], line, ref["res"][4], tl, ref["res"][0]))
    assert n > 1600, n
    assert not bad, (len(bad), bad[:5])


@pytest.mark.parametrize("seg", [64, 192, 4096])
def test_scanner_segments_vs_oracle(gpu, seg):
    """Small segments force many lanes per stream: speculative entry states,
    chain verification, fix-up rounds, cross-segment lineage walks."""
    import random
    ora = harness.OracleEngine()
    rng = random.Random(1234 + seg)
    zoo = [
        [rb"[a-z]+@[a-z]+\.[a-z]+"], [rb"([a-z]+)://([^/ ]+)(/[^ ?]*)?(\?[^ ]*)?"],
        [rb"a?a?a?aaa"], [rb"(a+)(b+)?"], [rb"(?:a.*b|a)"], [rb"x*"], [rb"(a|ab)(c|bcd)(d*)"],
        [b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"],
        [rb"(a*)*b"], [rb"a.c"], [rb"\Aab|\n^b"], [rb"(x+x+)+y"], [rb"[ab]c?"], [rb"(a|b)*?c"],
        # ^ in find-all counting: the initial list of every re-armed search depends on the byte in front
        [rb"^a|^c|c"], [rb"^b+"], [rb"^x*"], [rb"(^|a)b"],
        # look-ahead assertions, decided inside the automaton step (FIRST / Thompson only)
        # (not inside a loop — `(\B.)*?\b(x)`: there the VM's generation tags decide what a splice lists and
    # the tier declines, see test_nfa_tier_segments_vs_oracle)
    [rb"(\w+)\b(.)"], [rb"c$"], [rb"^(.*)$"], [rb"(a+)\b(?:\s|$)"], [rb"(b)\z"],
        [rb"a$", rb"\bb"], [rb"(?:$|a)(b|\b)"],
    ]
    alphabets = [b"abc", b"ab c\n.x@:/?y", b"aaaaab"]
    for pats in zoo:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            scs = {}
            for mode in (S.HIP_THOMPSON, S.HIP_PIKE_FIRST, S.HIP_PIKE_COUNT):
                try:
                    sc = S.Scanner(pool, prog, mode, S.ENGINE_SCAN)
                except RuntimeError:
                    continue
                sc.set_segment_bytes(seg)
                scs[mode] = sc
            assert S.HIP_PIKE_FIRST in scs, pats
            datas = []
            for i in range(12):
                alpha = alphabets[i % len(alphabets)]
                n = rng.choice([0, 1, 63, 64, 65, 200, 1000, 3000])
                datas.append(bytes(rng.choice(alpha) for _ in range(n)))
            datas.append(S.gen_data_host(2000, b"@abc.cc "))
            datas.append(S.gen_data_host(1500, b" abc://abc.cc/ab/c?a=b "))
            bufs = [S.DeviceBuffer.from_bytes(d) for d in datas]
            ptrs, lens = [b.ptr for b in bufs], [len(d) for d in datas]
            got = {m: sc.scan(ptrs, lens) for m, sc in scs.items()}
            for i, d in enumerate(datas):
                first, cnt = _expect(ora, prog, re.ncaps, d)
                assert got[S.HIP_PIKE_FIRST][i] == first, (pats, seg, d[:80], len(d))
                assert got[S.HIP_THOMPSON][i][0] == (0 if first[0] >= 0 else S.SRE_DECLINED), (pats, d[:80])
                if S.HIP_PIKE_COUNT in got:
                    assert got[S.HIP_PIKE_COUNT][i] == cnt, (pats, seg, d[:80], len(d))
            for b in bufs:
                b.free()


def test_scanner_fixup_rounds_are_reported(gpu):
    """A match early in a long stream leaves later segments assumed 'idle' while
    the truth is 'search over': the chain check must catch it."""
    with S.Pool() as pool:
        prog = S.compile(pool, S.parse(pool, [rb"(?:a.*b|a)"]))
        sc = S.Scanner(pool, prog, S.HIP_PIKE_FIRST, S.ENGINE_SCAN)
        sc.set_segment_bytes(64)
        data = b"xx a" + b"c" * 500 + b"b" + b"c" * 300
        buf = S.DeviceBuffer.from_bytes(data)
        rec = sc.scan([buf.ptr], [len(data)])[0]
        buf.free()
        assert rec == [0, 1, 3, 505]
        assert sc.last_fixups >= 1


def test_scanner_count_long_pending_match_converges_quickly(gpu):
    """COUNT: a match whose list lives on for the rest of the stream (a.*b with no
    b in sight) is pending at every segment boundary behind it.  Lanes of a
    fix-up round take the verified prefix's pending match as their belief, so
    the chain closes in a round or two instead of one segment per round."""
    ora = harness.OracleEngine()
    cases = [(rb"(?:a.*b|a)", b"xa" + b"c" * 20000),
             (rb"(?:a.*b|a)", b"xa" + b"c" * 9000 + b"b" + b"ca" + b"c" * 9000),
             (rb"(a)[^b]*(b)?", b"za" + b"c" * 30000 + b"aa")]
    for pat, data in cases:
        with S.Pool() as pool:
            re = S.parse(pool, [pat])
            prog = S.compile(pool, re)
            first, cnt = _expect(ora, prog, re.ncaps, data)
            sc = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, S.ENGINE_SCAN)
            sc.set_segment_bytes(64)
            buf = S.DeviceBuffer.from_bytes(data)
            got = sc.scan([buf.ptr], [len(data)])[0]
            buf.free()
            assert got == cnt, (pat, got[:6], cnt[:6])
            assert sc.last_fixups <= 4, (pat, sc.last_fixups)


def test_scanner_automaton_that_never_forgets_gets_exact_entry_states(gpu):
    """x(?:[^y]{3})*y behind an 'x': the state rotates with the input (period 3, which
    does not divide the segment size), so every speculative lane behind the x is wrong
    and a fix-up round is only sure to repair one segment.  After two rounds the
    scanner composes the segments' transition functions on the device and finishes in
    ONE exact pass (round-1 advisor finding: the rounds used to be linear in the
    number of segments, each with a host round trip)."""
    ora = harness.OracleEngine()
    cases = [([rb"x(?:[^y]{3})*y"], b"ab" * 50 + b"x" + b"abc" * 20000 + b"ab" + b"y" + b"zz"),
             ([rb"x(?:[^y]{3})*y"], b"x" + b"abc" * 20000 + b"y" + b"zz"),
             ([rb"(a)(?:[bc]{2})*(d)"], b"q" * 777 + b"a" + b"bc" * 30001 + b"bd" + b"bc" * 5000)]
    for seg in (256, 1280):
        for pats, data in cases:
            with S.Pool() as pool:
                re = S.parse(pool, pats)
                prog = S.compile(pool, re)
                first, cnt = _expect(ora, prog, re.ncaps, data)
                buf = S.DeviceBuffer.from_bytes(data)
                for mode, want in ((S.HIP_PIKE_FIRST, first), (S.HIP_THOMPSON, None)):
                    sc = S.Scanner(pool, prog, mode, S.ENGINE_SCAN)
                    sc.set_segment_bytes(seg)
                    rec = sc.scan([buf.ptr], [len(data)])[0]
                    if want is None:
                        assert rec[0] == (0 if first[0] >= 0 else S.SRE_DECLINED), (pats, seg, rec)
                    else:
                        assert rec == want, (pats, seg, rec, want)
                    assert sc.last_fixups <= 4, (pats, seg, mode, sc.last_fixups)
                buf.free()


def test_count_automaton_that_never_forgets_gets_exact_entry_states(gpu):
    """Find-all of quoted strings: whether a lane is inside or outside a string depends on the parity of the quotes
    in front of it, which 128 bytes of warm-up cannot tell — a third of the lanes guessed wrong, every fix-up round
    repaired one of them, and 64 MiB took 87 390 rounds = 11.4 s (tools/parity_probe.py).  COUNT now composes the
    segments' transition functions like FIRST does (the caller's restarts are part of the function, taken from the
    COUNT table and followed byte by byte where it leaves the fast path) and finishes in a few passes.  Counts and
    last matches against the oracle, rounds bounded; periods that do not divide the segment size."""
    import random
    ora = harness.OracleEngine()
    rng = random.Random(11)
    words = [b'"ab" cde ', b'"abc" "d" e', b'"" x', b'key: "va lue", ', b"'q' "]
    text = b"".join(rng.choice(words) for _ in range(9000))
    cases = [([rb'"[^"]*"'], b'"ab" cde ' * 9000), ([rb'"[^"]*"'], b'"abc" "d" e' * 7000), ([rb'"[^"]*"'], text),
             ([rb'"([^"]*)"'], text), ([rb'"[^"]*"', rb"'[^']*'"], text), ([rb"x(?:[^y]{3})*y"], b"xabcabcy z" * 8000)]
    for seg in (0, 256, 1280):
        for pats, data in cases:
            with S.Pool() as pool:
                re = S.parse(pool, pats)
                prog = S.compile(pool, re)
                first, cnt = _expect(ora, prog, re.ncaps, data)
                buf = S.DeviceBuffer.from_bytes(data)
                sc = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, S.ENGINE_SCAN)
                if seg:
                    sc.set_segment_bytes(seg)
                rec = sc.scan([buf.ptr], [len(data)])[0]
                buf.free()
                assert rec == cnt, (pats, seg, rec, cnt)
                assert sc.last_fixups <= 12, (pats, seg, sc.last_fixups)


@pytest.mark.parametrize("seg", [64, 4096])
def test_scanner_long_lineage_uses_ancestor_maps(gpu, seg):
    """A match that starts at offset 0 and ends at the far end of the stream:
    its captures cannot be found by walking a few segments back.  The scanner
    either crosses the stream in O(1) jumps over stable stretches (the thread list
    loops in place) or falls back to the parallel per-segment ancestor maps (and
    their 256-segment compositions) — bit-exact either way."""
    ora = harness.OracleEngine()
    cases = [
        ([rb"[a-z]+@[a-z]+\.[a-z]+"], S.gen_data_host(300000, b"@abc.cc ")),
        ([rb"([a-z]+)@([a-z]+)\.([a-z]+)"], S.gen_data_host(70000, b"@abc.cc ")),
        ([rb"(a|b|c)+(@)(x)?"], S.gen_data_host(50003, b"@")),
        ([rb"x(.*)y(.*)z"], b"..x" + b"ab" * 20000 + b"y" + b"cd" * 9000 + b"z.."),
    ]
    passes = []
    for pats, data in cases:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            first, cnt = _expect(ora, prog, re.ncaps, data)
            buf = S.DeviceBuffer.from_bytes(data)
            sc = S.Scanner(pool, prog, S.HIP_PIKE_FIRST, S.ENGINE_SCAN)
            sc.set_segment_bytes(seg)
            # several streams in one call, only some of them long-lineage
            other = S.DeviceBuffer.from_bytes(b"zz a@b.c zz")
            recs = sc.scan([other.ptr, buf.ptr, buf.ptr], [11, len(data), len(data)])
            assert recs[1] == first and recs[2] == first, (pats, seg, recs[1], first)
            passes.append(sc.last_lineage_passes)
            sc2 = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, S.ENGINE_SCAN)
            sc2.set_segment_bytes(seg)
            assert sc2.scan([buf.ptr], [len(data)])[0] == cnt, (pats, seg)
            buf.free()
            other.free()
    # a thread list that loops in place over the whole stream ([a-z]+ over letters) is crossed
    # by the stable-stretch jumps and needs no maps; the list of the third case changes
    # with every byte: it must take the ancestor maps
    assert passes[0] == 0 and passes[2] == 1, passes


def test_compat_api_large_buffers_take_the_scanner(gpu):
    """sre_vm_pike_exec / sre_vm_thompson_exec on large whole buffers (what the
    reference's bench/sregex.c does): routed through the scanner, same answers
    as the reference; the find-all iteration keeps working on the same context,
    including the hand-over to the VM kernel for the short remainder."""
    ora = harness.OracleEngine()
    eng = harness.ProductEngine()
    for rec in harness.load_jsonl("gen_data.jsonl"):
        if rec["n"] < 60000:
            continue
        pats = [bytes.fromhex(h) for h in rec["re"]]
        data = S.gen_data_host(rec["n"], bytes.fromhex(rec["tail"]))
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, pats))
            t = eng.thompson(prog)
            assert t.exec(data, True) == rec["thompson"], (rec["cfg"], rec["n"])
            p = eng.pike(prog, rec["ncaps"])
            rc = p.exec(data, True, want_pending=False)
            assert rc == rec["pike_rc"], (rec["cfg"], rec["n"], rc)
            if rc >= 0:
                assert list(p.ovector) == rec["pike_ov"], (rec["cfg"], rec["n"])
            eng.recycle()
    # iteration: many matches spread over a large buffer, empty matches, ^ after newline
    chunk = S.gen_data_host(50000, b" bob@example.com\n")
    cases = [
        ([rb"([a-z]+)@([a-z]+)\.[a-z]+"], chunk * 4 + b"x@y.z"),
        ([rb"^abc"], (b"abccc" * 9000 + b"\n") * 3),
        ([rb"x*"], b"ab" * 20000),
        ([rb"(a+)(b+)?"], b"b a\nca" + b" " * 40000 + b"a\nc" + b"." * 40000),
        # look-ahead assertions on a re-armed context: \b / \B go by its seen_word, $ ends a match
        # in front of a byte the next search reads again
        ([rb"\b(\w+)\b"], b"foo bar_1  baz\n" * 3000 + b"tail"),
        ([rb"(a+)$"], (b"xaa\n" + b"b" * 300 + b"a\n\n") * 200 + b"aaa"),
        ([rb"\B"], b"ab  c_d\n" * 2000),
        ([rb"x*\b"], b"xx yx\nz " * 2500),
    ]
    for pats, data in cases:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            limit = 200
            want = harness.findall(ora, prog, re.ncaps, data, limit)
            got = harness.findall(eng, prog, re.ncaps, data, limit)
            assert got == want, (pats, got[:3], want[:3], len(got), len(want))
            eng.recycle()


def _feed(ctx, data, sizes, nov):
    """Feed `data` in chunks of the given sizes (the last one with eof); one record per
    exec call: rc, the ovector the call defined, the pending match of an AGAIN."""
    out, off = [], 0
    sizes = list(sizes)
    while True:
        n = sizes.pop(0) if sizes else len(data) - off
        n = min(n, len(data) - off)
        eof = off + n >= len(data) and not sizes
        rc = ctx.exec(data[off:off + n], eof, want_pending=True)
        off += n
        if rc == S.SRE_AGAIN:
            out.append((rc, tuple(ctx.ovector[:2]), ctx.pending))
            continue
        out.append((rc, tuple(ctx.ovector[:nov]) if rc >= 0 else None, None))
        return out


def test_compat_api_search_after_a_vm_match_is_routed_afresh(gpu):
    """A context whose FIRST call is short (a header line) runs that call on the exact VM kernel.  When
    that search ends with a match the context is between two searches again (sre_vm_pike.c:624-628):
    the host takes its state back and the next call — a megabyte — goes through a throughput scanner
    instead of staying on the VM for the rest of the stream (round-2 advisor finding).  Results equal
    the oracle's for the same call sequence; the route counters say where the calls ran."""
    ora = harness.OracleEngine()
    eng = harness.ProductEngine()
    cases = [([rb"[a-z]+@[a-z]+\.[a-z]+"], b"hi a@b.c ", b" x@abc.cc "),
             ([rb"(a+)$"], b"xaa\n", b" baaa"),                      # look-ahead: seen_newline / seen_word travel
             ([rb"\bab\b"], b"ab ", b" ab"),
             ([b"a", b"ab", b"c"], b"zzc", b"aaabbccb")]
    for pats, head, tail in cases:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            data = head + S.gen_data_host(1 << 20, tail)

            def calls(ctx):
                out = []
                buf = ctypes.create_string_buffer(data, len(data))
                rc = ctx.exec(None, False, base=buf, offset=0, length=len(head))       # the short first call
                out.append((rc, list(ctx.ovector)[:2] if rc >= 0 else None))
                off = ctx.ovector[1] if rc >= 0 else len(head)
                for _ in range(3):                                                     # then whole remainders
                    rc = ctx.exec(None, True, base=buf, offset=off, length=len(data) - off)
                    out.append((rc, list(ctx.ovector) if rc >= 0 else None))
                    if rc < 0:
                        break
                    off = ctx.ovector[1]
                return out

            want = calls(ora.pike(prog, re.ncaps))
            before = S.compat_route_counts()
            got = calls(eng.pike(prog, re.ncaps))
            after = S.compat_route_counts()
            assert got == want, (pats, got, want)
            assert want[0][0] >= 0, (pats, want)            # the short call ends its search with a match
            # the short call ran on the VM, and so does a last call on a remainder below the scanners'
            # 16-byte floor; every other one — the megabyte first of all — on a throughput scanner
            assert after[0] - before[0] >= 1, (pats, before, after)
            assert after[2] - before[2] <= 2, (pats, before, after)
            assert (after[0] - before[0]) + (after[2] - before[2]) == len(got), (pats, before, after, len(got))
            eng.recycle()


def test_compat_api_chunked_streams_take_the_scanner(gpu):
    """sre_vm_pike_exec fed in CHUNKS (eof = 0, then a last chunk with eof): the chunks run
    on the table-driven scanner, the thread list travelling from chunk to chunk as automaton
    state + one capture vector per listed thread.  Every call must answer what the oracle
    answers to the same call sequence: SRE_AGAIN with its temporary match range and the
    pending match (sre_vm_pike.c:640-688, 692-735), then the match / SRE_DECLINED."""
    import random
    ora = harness.OracleEngine()
    eng = harness.ProductEngine()
    rng = random.Random(4242 + int(os.environ.get("SRE_FUZZ_SEED", "0")))
    cfg3 = [b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"]
    zoo = [[rb"[a-z]+@[a-z]+\.[a-z]+"], [rb"([a-z]+)://([^/ ]+)(/[^ ?]*)?(\?[^ ]*)?"], [rb"a?a?a?aaa"],
           [rb"(a+)(b+)?"], [rb"(?:a.*b|a)"], [rb"x(.*)y(.*)z"], [rb"(a|ab)(c|bcd)(d*)"], cfg3,
           [rb"\Aab|\n^b"], [rb"(x+x+)+y"], [rb"^b+"], [rb"q(\w+)@"],
           # look-ahead assertions: a splice at the first byte of a chunk goes by the context's
           # seen_newline / seen_word, not by the byte in front (sre_vm_pike.c:276-285, 492)
           [rb"\bab\b"], [rb"(a+)$"], [rb"(\w+)\b(.)"], [rb"c\B(.)"], [rb"^(\w+) \b"], [rb"x*\b y"]]
    tails = [b"@abc.cc ", b" abc://abc.cc/ab/c?a=b ", b"aaabbccb", b" a\nca", b"xabyabz", b"q"]
    n = 0
    on_vm = set()
    for pats in zoo:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            nov = 2 * (re.ncaps + 1)
            for trial in range(6):
                if trial < 3:
                    data = S.gen_data_host(rng.choice([9000, 40000, 150000]), tails[rng.randrange(len(tails))])
                else:
                    alpha = [b"abc", b"ab c\n.x@:/?y", b"aaaaab xy\nz"][trial - 3]
                    data = bytes(rng.choice(alpha) for _ in range(rng.choice([5000, 20000, 70000])))
                first = rng.choice([256, 300, 1000, 4096, 5000, 8192, 30000])
                sizes = [first] + [rng.choice([0, 1, 7, 64, 1000, 4096, 10000, 33333]) for _ in range(rng.randrange(0, 9))]
                want = _feed(ora.pike(prog, re.ncaps), data, sizes, nov)
                before = S.compat_route_counts()
                got = _feed(eng.pike(prog, re.ncaps), data, sizes, nov)
                after = S.compat_route_counts()
                n += 1
                assert got == want, (pats, len(data), sizes, got[-3:], want[-3:])
                # ... and every one of these calls ran as a chunk on the table-driven scanner
                # (or, for a program whose chunks it does not take, every one on the exact VM)
                if after[2] != before[2]:
                    assert after[1] == before[1], (pats, before, after)
                    if len(got) > 1:
                        on_vm.add(tuple(pats))
                else:
                    # (a subject that fits the first chunk is one whole-buffer call)
                    assert (after[0] - before[0]) + (after[1] - before[1]) == len(got), (pats, before, after, len(got))
                eng.recycle()
    assert n == 6 * len(zoo)
    assert not on_vm, on_vm     # no multi-chunk stream of this zoo fell back to the exact VM
    # sre_vm_thompson_exec in chunks: the list travels as the automaton state alone
    n = 0
    for pats in zoo:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            for trial in range(4):
                if trial < 2:
                    data = S.gen_data_host(rng.choice([9000, 40000, 150000]), tails[rng.randrange(len(tails))])
                else:
                    alpha = [b"abc", b"ab c\n.x@:/?y"][trial - 2]
                    data = bytes(rng.choice(alpha) for _ in range(rng.choice([5000, 20000, 70000])))
                    if trial == 3:
                        data = data.replace(b"a", b"d")     # mostly no match: AGAIN ... DECLINED
                first = rng.choice([256, 1000, 4096, 5000, 8192, 30000])
                sizes = [first] + [rng.choice([0, 1, 7, 64, 1000, 4096, 10000, 33333]) for _ in range(rng.randrange(0, 9))]
                res = []
                for e in (ora, eng):
                    ctx, off, rcs, todo = e.thompson(prog), 0, [], list(sizes)
                    while True:
                        k = min(todo.pop(0) if todo else len(data) - off, len(data) - off)
                        eof = off + k >= len(data) and not todo
                        rc = ctx.exec(data[off:off + k], eof)
                        rc = rc[0] if isinstance(rc, (list, tuple)) else rc
                        rcs.append(rc)
                        off += k
                        if rc != S.SRE_AGAIN:
                            break
                    res.append(rcs)
                n += 1
                assert res[0] == res[1], (pats, len(data), sizes, res)
                eng.recycle()
    assert n == 4 * len(zoo)
    # ... where chunking changes this VM's answer: its \A / ^ / \b are local to the buffer of a call
    before = S.compat_route_counts()
    for pats, data, sizes in (([rb"$\A\nb"], b"#" * 5000 + b"\nb", [5000]), ([rb"\B\Ax"], b"_" * 4097 + b"x", [4097]),
                              ([rb"a\B\bc"], b"#" * 4095 + b"ac", [4096]), ([rb"$^\nb"], b"#" * 4200 + b"\nb", [4200])):
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            res = []
            for e in (ora, eng):
                ctx, off, rcs, todo = e.thompson(prog), 0, [], list(sizes)
                while True:
                    k = min(todo.pop(0) if todo else len(data) - off, len(data) - off)
                    rc = ctx.exec(data[off:off + k], off + k >= len(data) and not todo)
                    rc = rc[0] if isinstance(rc, (list, tuple)) else rc
                    rcs.append(rc)
                    off += k
                    if rc != S.SRE_AGAIN:
                        break
                res.append(rcs)
            whole = ora.thompson(prog).exec(data, True)
            assert res[0] == res[1] and res[0][-1] != whole, (pats, res, whole)
            eng.recycle()
    after = S.compat_route_counts()
    assert after[1] == before[1], (before, after)      # look-ahead programs: this VM's chunks stay on the exact VM kernel
    # a long thread list carried from chunk to chunk: BASELINE configs[2] (12 regexes, 37 list-able
    # threads) over a subject that keeps the search alive across several chunks
    with S.Pool() as pool:
        re = S.parse(pool, cfg3)
        prog = S.compile(pool, re)
        nov = 2 * (re.ncaps + 1)
        for data in (b"#" * 9000 + b"e" + b"#" * 3000 + b"ef#gh BLAH", b"#" * 5000 + b"g" + b"#" * 4500 + b"BLA" + b"#" * 100 + b"abcd",
                     b"#" * 13000):
            for sizes in ([4096, 4096, 1000, 1, 4096], [5000, 7000], [4096] * 6):
                want = _feed(ora.pike(prog, re.ncaps), data, sizes, nov)
                before = S.compat_route_counts()
                got = _feed(eng.pike(prog, re.ncaps), data, sizes, nov)
                after = S.compat_route_counts()
                assert got == want, (len(data), sizes, got[-2:], want[-2:])
                assert after[2] == before[2], (before, after)       # none of it on the exact VM
                eng.recycle()
    # every call refreshes seen_newline / seen_word from a match reached DURING it, a pending one
    # too (sre_vm_pike.c:586-601): here the pending "a" ends one byte before the end of the first
    # chunk, and the \B / \b thread listed behind the blank is decided by the next chunk's first byte
    for pats in ([rb"a(\s\B.)?"], [rb"a(\s\b.)?"], [rb"a(\n^b|\s\B\B.)?"]):
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            nov = 2 * (re.ncaps + 1)
            for nxt in (b"b", b".", b"\n"):
                for blank in (b" ", b"\n"):
                    data = b"#" * 4094 + b"a" + blank + nxt + b"b# a b"
                    for sizes in ([4096], [4096, 1], [4095, 1, 1], [4097]):
                        want = _feed(ora.pike(prog, re.ncaps), data, sizes, nov)
                        got = _feed(eng.pike(prog, re.ncaps), data, sizes, nov)
                        assert got == want, (pats, nxt, blank, sizes, got, want)
                        eng.recycle()
    # random patterns (all constructs, all assertions) over sparse subjects — filler bytes with
    # short bursts of the patterns' alphabet, so that searches live across several chunks
    taken = 0
    for _ in range(60):
        pats = [harness.random_regex(rng)]
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            nov = 2 * (re.ncaps + 1)
            try:
                S.Scanner(pool, prog, S.HIP_PIKE_FIRST, S.ENGINE_SCAN)
            except RuntimeError:
                if prog_has_lookahead(pats):
                    continue        # the reference VM itself may diverge on these
            for trial in range(2):
                filler = rng.choice([b"#", b"c", b" ", b"\n"])
                data = bytearray(filler * rng.choice([9000, 30000]))
                for _ in range(rng.randrange(0, 12)):
                    at = rng.randrange(0, len(data))
                    burst = bytes(rng.choice(b"abcx \n_.") for _ in range(rng.randrange(1, 6)))
                    data[at:at + len(burst)] = burst
                data = bytes(data[:rng.choice([9000, 30000])])
                sizes = [rng.choice([256, 257, 700, 4096, 4097, 6000])] + [rng.choice([0, 1, 2, 64, 1000, 4096, 5000]) for _ in range(rng.randrange(0, 7))]
                want = _feed(ora.pike(prog, re.ncaps), data, sizes, nov)
                before = S.compat_route_counts()
                got = _feed(eng.pike(prog, re.ncaps), data, sizes, nov)
                after = S.compat_route_counts()
                assert got == want, (pats, data.hex() if len(data) < 200 else len(data), sizes, got[-3:], want[-3:])
                taken += after[1] - before[1]
                eng.recycle()
    assert taken > 100, taken
    # chunks of the size that travels through the pinned staging buffer (64 KiB .. 2 MiB), a
    # different content every time: the buffer is reused from call to call
    with S.Pool() as pool:
        re = S.parse(pool, [rb"q(\w+)@(x*)"])
        prog = S.compile(pool, re)
        nov = 2 * (re.ncaps + 1)
        for trial in range(3):
            data = bytearray(rng.choice(b"abc xyz\n") for _ in range(3 << 20))
            at = [len(data) - 40, (2 << 20) + 17, len(data) - 70000][trial]
            data[at:at + 8] = b"qabc@xxx"
            data = bytes(data)
            sizes = [[1 << 20, 70000, 1 << 20], [65536, 2 << 20, 100000], [300000] * 6][trial]
            want = _feed(ora.pike(prog, re.ncaps), data, sizes, nov)
            got = _feed(eng.pike(prog, re.ncaps), data, sizes, nov)
            assert got == want, (trial, got[-2:], want[-2:])
            eng.recycle()
    # find-all over a chunked stream: after each match the caller re-feeds from the match end
    chunk = S.gen_data_host(50000, b" bob@example.com\n")
    for pats, data in (([rb"([a-z]+)@([a-z]+)\.[a-z]+"], chunk * 3 + b"x@y.z"), ([rb"^abc"], (b"abccc" * 3000 + b"\n") * 3)):
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            nov = 2 * (re.ncaps + 1)
            res = []
            for e in (ora, eng):
                ctx, off, found = e.pike(prog, re.ncaps), 0, []
                while len(found) < 50:
                    end = min(off + 16384, len(data))
                    rc = ctx.exec(data[off:end], end == len(data), want_pending=True)
                    if rc == S.SRE_AGAIN:
                        off = end
                        continue
                    if rc < 0:
                        found.append((rc,))
                        break
                    found.append((rc,) + tuple(ctx.ovector[:nov]))
                    off = ctx.ovector[1]
                res.append(found)
            eng.recycle()
            assert res[0] == res[1], (pats, res[0][:3], res[1][:3])


def test_compat_api_large_chunks_arrive_intact(gpu):
    """Chunks of 512 KiB and more travel through the pinned ring (sre_vm_api.cpp ring_upload:
    units copied by several threads, fetched by the GPU as they complete, slots reused above
    16 MiB).  A 40 MiB stream with one marker per 256 KiB unit at an offset that differs from
    unit to unit — the first and last byte of a unit among them — is searched to its end through
    sre_vm_pike_exec (re-fed from every match end, like the reference's clients): every match
    position must be the marker's, for chunk sizes that are no multiple of anything, with
    helper threads and without."""
    import numpy as np
    total, unit = 40 << 20, 256 << 10
    arr = np.full(total, ord("a"), dtype=np.uint8)
    want = []
    for k in range(total // unit):
        o = [0, unit - 1, (k * 7919 + 13) % unit][k % 3] if k % 5 else (k * 104729) % unit
        if k * unit + o == 0:
            o = 5
        arr[k * unit + o] = ord("q")
        want.append(k * unit + o)
    data = arr.tobytes()
    buf = ctypes.create_string_buffer(data, total)
    for step in ((512 << 10) + 3, (5 << 20) + 1, (17 << 20) + 5, total):
        with S.Pool() as pool:
            re = S.parse(pool, [b"q"])
            prog = S.compile(pool, re)
            got = []
            with S.Pool() as ep:
                ctx = S.PikeCtx(ep, prog, re.ncaps)
                off = 0                 # absolute offset of the next byte to feed
                edge = min(step, total) # end of the caller's current chunk
                while True:
                    eof = edge >= total
                    rc = ctx.exec(None, eof, want_pending=False, base=buf, offset=off, length=edge - off)
                    if rc == S.SRE_AGAIN:
                        off, edge = edge, min(edge + step, total)
                        continue
                    if rc == S.SRE_DECLINED:
                        break
                    assert rc == 0, rc
                    got.append(ctx.ovector[0])
                    assert ctx.ovector[1] == ctx.ovector[0] + 1
                    off = ctx.ovector[1]
                    if off >= edge and not eof:
                        edge = min(edge + step, total)
            assert got == want, (step, len(got), len(want), [(a, b) for a, b in zip(got, want) if a != b][:4])
        if step == (5 << 20) + 1:
            # sre_hip_compat_trim() frees the parked streams AND stops the ring's helper threads and frees
            # its pinned memory; the next large chunk starts them again
            gpu.sre_hip_compat_trim()


def test_compat_api_contexts_on_several_host_threads(gpu):
    """The reference has no globals: separate programs and pools may be used from different threads
    (SURVEY.md 8b).  Four host threads each compile their own program and stream their own 24 MiB
    subject through sre_vm_pike_exec in chunks that travel through the ONE process-wide pinned ring
    (1 MiB and 5 MiB + 1: ring_begin serialises the chunks and waits for the fetch of another
    context's chunk), plus byte-sized calls on the exact VM in between.  Every thread must see its
    own stream's answer."""
    import threading
    total = 24 << 20
    errors, results = [], {}

    def work(k):
        try:
            tail = b" a%d@abc.cc " % k
            data = S.gen_data_host(total, tail)
            L = len(data)
            buf = ctypes.create_string_buffer(data, L)
            with S.Pool() as pool:
                re = S.parse(pool, [rb"[a-z]+[0-9]@[a-z]+\.[a-z]+" if k % 2 else rb"([a-z]+)%d@([a-z]+)\.[a-z]+" % k])
                prog = S.compile(pool, re)
                for step in ((1 << 20), (5 << 20) + 1):
                    with S.Pool() as ep:
                        ctx = S.PikeCtx(ep, prog, re.ncaps)
                        off, rc = 0, S.SRE_AGAIN
                        while rc == S.SRE_AGAIN:
                            n = min(step, L - off)
                            rc = ctx.exec(None, off + n >= L, want_pending=False, base=buf, offset=off, length=n)
                            off += n
                        results[(k, step)] = (rc, list(ctx.ovector[:2]), L)
                    # a short subject in byte-sized calls: the exact VM's kernels next to the others' chunks
                    with S.Pool() as ep:
                        ctx = S.PikeCtx(ep, prog, re.ncaps)
                        small = b"xx" + tail
                        rc = S.SRE_AGAIN
                        for i in range(len(small)):
                            rc = ctx.exec(small[i:i + 1], i + 1 == len(small), want_pending=False)
                            if rc != S.SRE_AGAIN:
                                break
                        results[(k, "bytes", step)] = (rc, list(ctx.ovector[:2]))
        except Exception as e:      # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not any(t.is_alive() for t in threads), "a thread is stuck"
    assert not errors, errors
    for k in range(4):
        for step in ((1 << 20), (5 << 20) + 1):
            rc, ov, L = results[(k, step)]
            assert rc == 0 and ov == [L - 10, L - 1], (k, step, rc, ov, L)
            rc, ov = results[(k, "bytes", step)]
            assert rc == 0 and ov == [3, 12], (k, step, rc, ov)


def test_compat_api_chunked_stream_rate(gpu):
    """A 256 MiB stream through sre_vm_pike_exec in 1 MiB chunks (host memory): same final
    answer as one whole-buffer call; the sustained rate is printed and written to
    gpurun_out/stream_rate.json (PCIe- and host-memcpy-inclusive: never bench.py's value)."""
    import json
    import time
    pats, tail = [rb"[a-z]+@[a-z]+\.[a-z]+"], b" a@abc.cc "
    total, chunk = 256 << 20, 1 << 20
    data = S.gen_data_host(total, tail)
    L = len(data)
    rows = {}
    with S.Pool() as pool:
        re = S.parse(pool, pats)
        prog = S.compile(pool, re)
        # each shape twice: the first pass pays for the context's device buffers
        for name, step in (("1 MiB chunks (cold)", chunk), ("1 MiB chunks", chunk),
                           ("16 MiB chunks (cold)", 16 << 20), ("16 MiB chunks", 16 << 20),
                           ("64 KiB chunks", 64 << 10), ("4 MiB chunks", 4 << 20), ("64 MiB chunks", 64 << 20)):
            with S.Pool() as ep:
                ctx = S.PikeCtx(ep, prog, re.ncaps)
                buf = ctypes.create_string_buffer(data, L)
                t0 = time.perf_counter()
                off, rc, calls, per_call = 0, S.SRE_AGAIN, 0, []
                while rc == S.SRE_AGAIN:
                    n = min(step, L - off)
                    t1 = time.perf_counter()
                    rc = ctx.exec(None, off + n >= L, want_pending=False, base=buf, offset=off, length=n)
                    per_call.append(time.perf_counter() - t1)
                    off += n
                    calls += 1
                dt = time.perf_counter() - t0
                assert rc == 0 and list(ctx.ovector) == [L - 9, L - 1], (name, rc, list(ctx.ovector))
                per_call.sort()
                med = per_call[len(per_call) // 2]
                # GBps: the whole stream, the context's set-up (its first call) included; the
                # steady state of a long stream is the median call
                rows[name] = {"calls": calls, "seconds": dt, "GBps": L / dt / 1e9, "median_call_us": med * 1e6,
                              "first_call_us_max": per_call[-1] * 1e6, "GBps_median_call": step / med / 1e9}
                print(name, rows[name])
    out = os.path.join(harness.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "stream_rate.json"), "w") as f:
        json.dump(rows, f, indent=1)
    assert rows["1 MiB chunks"]["GBps"] > 0.5, rows


def test_full_size_streams_closed_form_properties(gpu):
    """BASELINE.json's full-size configurations (4 GiB gen-data streams).  The
    oracle cannot run there, so the expected records are closed forms in the
    stream length — each closed form is first checked against the oracle at a
    size the oracle handles, then the scanner must reproduce it at 4 GiB."""
    ora = harness.OracleEngine()
    cfg3 = [b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"]
    uri_tail = b" abc://abc.cc/ab/c?a=b "
    cases = [
        # (patterns, mode, tail, expected(L) -> record)
        ([rb"[a-z]+@[a-z]+\.[a-z]+"], S.HIP_PIKE_FIRST, b"aaabbccb", lambda L: [S.SRE_DECLINED, 0, -1, -1]),
        ([rb"[a-z]+@[a-z]+\.[a-z]+"], S.HIP_PIKE_FIRST, b"@abc.cc ", lambda L: [0, 1, 0, L - 1]),
        ([rb"([a-z]+)://([^/ ]+)(/[^ ?]*)?(\?[^ ]*)?"], S.HIP_PIKE_FIRST, uri_tail,
         lambda L: [0, 1, L - 22, L - 1, L - 22, L - 19, L - 16, L - 10, L - 10, L - 5, L - 5, L - 1]),
        (cfg3, S.HIP_PIKE_COUNT, b"aaabbccb", lambda L: [7, L, L - 1, L, -1, -1]),
        ([rb"a?a?a?aaa"], S.HIP_THOMPSON, b"aaabbccb", lambda L: [0, 1, -1, -1]),
        ([rb"a?a?a?aaa"], S.HIP_PIKE_FIRST, b"aaabbccb", lambda L: [0, 1, L - 8, L - 5]),
    ]
    big = 4 << 30
    buf = S.DeviceBuffer(big)
    for pats, mode, tail, expect in cases:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            # the closed form against the oracle
            small = S.gen_data_host(20000, tail)
            first, cnt = _expect(ora, prog, re.ncaps, small)
            want_small = {S.HIP_PIKE_FIRST: first, S.HIP_PIKE_COUNT: cnt,
                          S.HIP_THOMPSON: [0 if first[0] >= 0 else S.SRE_DECLINED, 1 if first[0] >= 0 else 0]
                          + [-1] * (2 * (re.ncaps + 1))}[mode]
            assert expect(len(small)) == want_small, (pats, tail, expect(len(small)), want_small)
            # the real thing
            L = S.gen_data_length(big, len(tail))
            assert gpu.sre_hip_gen_data(buf.ptr, L, tail, len(tail), None) == 0
            sc = S.Scanner(pool, prog, mode, S.ENGINE_SCAN)
            rec = sc.scan([buf.ptr], [L])[0]
            assert rec == expect(L), (pats, tail, rec, expect(L))
            assert sc.last_fixups == 0
    buf.free()


def test_configs4_per_gpu_shape_128_streams_of_64MiB(gpu):
    """BASELINE.json configs[4] as ONE GPU sees it: 128 independent 64 MiB streams
    (stream g of the 1024 lives on rank g mod 8), tails alternating matching /
    non-matching => 64 matches on this GPU.  Expected records are closed forms in
    the stream length, first checked against the oracle on short streams with the
    same tails; the same batch also runs through COUNT and Thompson."""
    ora = harness.OracleEngine()
    pats = [rb"[a-z]+@[a-z]+\.[a-z]+"]
    tails = [b" a@abc.cc ", b"aaabbccb"]

    def expect(L, tail, mode):
        hit = b"@" in tail
        if mode == S.HIP_THOMPSON:
            return [0 if hit else S.SRE_DECLINED, 1 if hit else 0, -1, -1]
        return [0, 1, L - 9, L - 1] if hit else [S.SRE_DECLINED, 0, -1, -1]

    with S.Pool() as pool:
        re = S.parse(pool, pats)
        prog = S.compile(pool, re)
        for tail in tails:                                  # the closed forms against the oracle
            small = S.gen_data_host(30000, tail)
            first, cnt = _expect(ora, prog, re.ncaps, small)
            assert first == expect(len(small), tail, S.HIP_PIKE_FIRST)
            assert cnt == expect(len(small), tail, S.HIP_PIKE_COUNT)
        per, nstreams = 64 << 20, 128
        lens = [S.gen_data_length(per, len(tails[g % 2])) for g in range(nstreams)]
        bufs = [S.DeviceBuffer(per) for _ in range(nstreams)]
        for g, b in enumerate(bufs):
            t = tails[g % 2]
            assert gpu.sre_hip_gen_data(b.ptr, lens[g], t, len(t), None) == 0
        ptrs = [b.ptr for b in bufs]
        for mode in (S.HIP_PIKE_FIRST, S.HIP_PIKE_COUNT, S.HIP_THOMPSON):
            sc = S.Scanner(pool, prog, mode, S.ENGINE_SCAN)
            recs = sc.scan(ptrs, lens)
            assert sum(1 for r in recs if r[0] >= 0) == 64
            for g, r in enumerate(recs):
                assert r == expect(lens[g], tails[g % 2], mode), (mode, g, r)
            assert sc.last_fixups == 0
        for b in bufs:
            b.free()


# ------------------------------------------------------------ bit-parallel NFA tier

NFA_ZOO = [
    # ordered-list automata beyond the table-driven scanner's 55 states
    [rb"(?:a|b)*a(?:a|b){7}@"], [rb"(a|b)*a(a|b){5}(c)"], [rb"[ab]{3,9}c{2}(x)?"], [rb"x.{0,10}y"],
    [rb"(a|ab|abc){2,6}x"], [rb"(?:[^,]*,){3,}d"], [rb"a[^x]{20}x"], [rb"(\w+ ){3}(\w+)"],
    [rb"^a.{3}b", rb"\nc{2,4}"], [rb"(?:a|b)*a(?:a|b){12}"],
    # and ones it would take anyway: the tier must agree with everything else
    [rb"[a-z]+@[a-z]+\.[a-z]+"], [rb"([a-z]+)://([^/ ]+)(/[^ ?]*)?(\?[^ ]*)?"], [rb"a?a?a?aaa"],
    [rb"(a+)(b+)?"], [rb"(?:a.*b|a)"], [rb"\Aab|\n^b"], [rb"^b+"], [rb"(^|a)b"], [rb"(x+x+)+y"],
    [b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"],
    # look-ahead assertions: decided by the next byte inside the set step (expansion tables)
    # (not inside a loop — `(\B.)*?\b(x)`: there the VM's generation tags decide what a splice lists and
    # the tier declines, see test_nfa_tier_segments_vs_oracle)
    [rb"(\w+)\b(.)"], [rb"c$"], [rb"^(.*)$"], [rb"(a+)\b(?:\s|$)"], [rb"(b)\z"],
    [rb"a$", rb"\bb"], [rb"(?:$|a)(b|\b)"], [rb"\bab\b"], [rb"x\B[ab]{2,8}\b"], [rb"^\xe7\xab\xa0$", rb"(a|b)*a(a|b){5}c"],
]


@pytest.mark.parametrize("seg", [64, 192, 0])
def test_nfa_tier_segments_vs_oracle(gpu, seg):
    """The bit-parallel NFA tier (thread set as a mask per lane, exact VM over the
    window from the last clean position), forced, against the oracle: first match
    + captures and Thompson; small segments force speculative entry sets, the
    chain check and fix-up rounds."""
    _nfa_zoo_vs_oracle(gpu, seg)


# sre_nfa.h SRE_NFA_SA_*: every option set that selects another variant of the shift-and kernel
# (masked shift, events from the consumed set, 64-bit masks with and without a carry, explicit ".*?"
# thread without merging), and 128 = the plain slices only (the round-2 kernel)
@pytest.mark.parametrize("sa", [1, 2, 4, 12, 48, 15, 100, 128])
@pytest.mark.parametrize("seg", [64, 0])
def test_nfa_tier_shift_and_variants_vs_oracle(gpu, seg, sa, monkeypatch):
    monkeypatch.setenv("SRE_HIP_NFA_SA", str(sa))
    kernels = _nfa_zoo_vs_oracle(gpu, seg)
    print("sa option", sa, "kernels:", sorted(kernels))
    if sa == 128:
        assert not any("nfa_sa" in k for k in kernels), kernels
    else:
        assert any("nfa_sa" in k for k in kernels), kernels


def _nfa_zoo_vs_oracle(gpu, seg):
    import random
    kernels = set()
    ora = harness.OracleEngine()
    rng = random.Random(99 + seg)
    alphabets = [b"abc", b"ab c\n.x@:/?y,d", b"aaaaab", b"ab"]
    with S.Pool() as pool:
        # a look-ahead assertion inside a loop has no set form (sre_nfa.cpp): declined, the exact VM takes it
        prog = S.compile(pool, S.parse(pool, [rb"(\B.)*?\b(x)"]))
        with pytest.raises(RuntimeError):
            S.Scanner(pool, prog, S.HIP_PIKE_FIRST, S.ENGINE_NFA)
    for pats in NFA_ZOO:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            scs = {m: S.Scanner(pool, prog, m, S.ENGINE_NFA) for m in (S.HIP_THOMPSON, S.HIP_PIKE_FIRST)}
            for sc in scs.values():
                assert sc.engine == S.ENGINE_NFA
                kernels.add(sc.kernel_name)
                if seg:
                    sc.set_segment_bytes(seg)
            datas = []
            for i in range(12):
                alpha = alphabets[i % len(alphabets)]
                n = rng.choice([0, 1, 63, 64, 65, 200, 1000, 3000, 20000])
                datas.append(bytes(rng.choice(alpha) for _ in range(n)))
            datas.append(S.gen_data_host(2000, b"@abc.cc "))
            datas.append(S.gen_data_host(1500, b" abc://abc.cc/ab/c?a=b "))
            datas.append(S.gen_data_host(70000, b" abaabaabab@ "))
            datas.append(b"ab" * 30000 + b"a" + b"ba" * 6 + b"c")
            bufs = [S.DeviceBuffer.from_bytes(d) for d in datas]
            ptrs, lens = [b.ptr for b in bufs], [len(d) for d in datas]
            got = {m: sc.scan(ptrs, lens) for m, sc in scs.items()}
            for i, d in enumerate(datas):
                first, _ = _expect(ora, prog, re.ncaps, d)
                assert got[S.HIP_PIKE_FIRST][i] == first, (pats, seg, d[:60], len(d), got[S.HIP_PIKE_FIRST][i], first)
                assert got[S.HIP_THOMPSON][i][:2] == [0 if first[0] >= 0 else S.SRE_DECLINED,
                                                     1 if first[0] >= 0 else 0], (pats, seg, d[:60])
            for b in bufs:
                b.free()
    return kernels


@pytest.mark.parametrize("sa", [0, 128, 4 + 8, 1 + 64])
def test_nfa_tier_program_that_never_forgets_gets_exact_entry_sets(gpu, sa, monkeypatch):
    """A thread that stays alive from an x far back (`x[^y]*y...`, `x.*y...` over a stream without newlines): no lane
    can know from its warm-up that the thread is alive, and a fix-up round carried the knowledge ONE segment further
    (64 MiB: 262 143 rounds, 16 s; tools/nfa_never_forgets.py).  After two speculative rounds the tier now walks
    every remaining segment with a wave whose lanes enter with the singleton sets, folds those exit sets into every
    lane's exact entry set (the step is a union-homomorphism) and finishes in one exact pass — under every table
    form (shift-and 32 / 64 bits with carry / masked, plain slices, look-ahead).  Results against the oracle,
    rounds bounded."""
    monkeypatch.setenv("SRE_HIP_NFA_SA", str(sa))
    ora = harness.OracleEngine()
    tail = b" abaabaabab@ "
    cases = [([rb"x[^y]*y(?:a|b)*a(?:a|b){7}@"], b"ab" * 33 + b"x" + b"abccc" * 30000 + b"y" + b"abaabaabab@ zz"),
             ([rb"x[^y]*y(?:a|b)*a(?:a|b){7}@"], b"x" + b"abccc" * 30000 + tail),                # no y: no match
             ([rb"x.*y(?:a|b)*a(?:a|b){7}@"], b"q" * 700 + b"x" + b"abccc" * 25000 + b"yabaabaabab@ " + b"abccc" * 3000),
             ([rb"x.*y(?:a|b)*a(?:a|b){7}@$"], b"x" + b"abccc" * 25000 + b"yabaabaabab@"),       # look-ahead form
             ([rb"x[^y]*y(?:a|b)*a[ab]{20}c[^x]{30}@"], b"x" + b"abccc" * 20000 + b"y" + b"a" * 22 + b"c" + b"b" * 30 + b"@ ")]
    for seg in (256, 1280):
        for pats, data in cases:
            with S.Pool() as pool:
                re = S.parse(pool, pats)
                prog = S.compile(pool, re)
                first, cnt = _expect(ora, prog, re.ncaps, data)
                buf = S.DeviceBuffer.from_bytes(data)
                for mode, want in ((S.HIP_PIKE_FIRST, first), (S.HIP_THOMPSON, None)):
                    try:
                        sc = S.Scanner(pool, prog, mode, S.ENGINE_NFA)
                    except RuntimeError:
                        continue
                    sc.set_segment_bytes(seg)
                    rec = sc.scan([buf.ptr], [len(data)])[0]
                    if want is None:
                        assert rec[0] == (0 if first[0] >= 0 else S.SRE_DECLINED), (pats, seg, rec, sc.kernel_name)
                    else:
                        assert rec == want, (pats, seg, rec, want, sc.kernel_name)
                    assert sc.last_fixups <= 6, (pats, seg, mode, sc.last_fixups, sc.kernel_name)
                buf.free()


def test_nfa_tier_takes_what_the_step_automaton_declines(gpu, blocks):
    """ENGINE_AUTO: every reference block whose ordered-list automaton is too large
    for the table-driven scanner and whose program has a bit-parallel form runs
    on the NFA tier — first match + captures and Thompson equal the reference
    CLI's lines."""
    bad, n, engines = [], 0, {S.ENGINE_VM: 0, S.ENGINE_SCAN: 0, S.ENGINE_NFA: 0}
    kernels = {}
    for blk in blocks:
        subject = bytes.fromhex(blk["s"])
        for name, regexes, flags, multi, ref in harness.block_variants(blk):
            if ref["rc"] != 0:
                continue
            with S.Pool() as pool:
                prog = S.compile(pool, S.parse(pool, regexes, flags, multi))
                sc = S.Scanner(pool, prog, S.HIP_PIKE_FIRST, S.ENGINE_AUTO)
                engines[sc.engine] += 1
                if sc.engine != S.ENGINE_NFA:
                    continue
                kernels[sc.kernel_name] = kernels.get(sc.kernel_name, 0) + 1
                buf = S.DeviceBuffer.from_bytes(subject)
                rec = sc.scan([buf.ptr], [len(subject)])[0]
                th = S.Scanner(pool, prog, S.HIP_THOMPSON, S.ENGINE_AUTO).scan([buf.ptr], [len(subject)])[0]
                buf.free()
                nov = 2 * (ref["ncaps"] + 1)
                line = ("pike match %d%s" % (rec[0], harness._fmt_caps(rec[2:], nov)) if rec[0] >= 0
                        else "pike no match")
                tl = "thompson " + ("match" if th[0] == 0 else "no match")
                n += 1
                if line != ref["res"][4] or tl != ref["res"][0]:
                    bad.append((blk["file"], blk["name"], name, line, ref["res"][4], tl, ref["res"][0]))
    print("engine admission over the reference runs (AUTO, first match):", engines)
    print("NFA-tier kernels of those runs:", sorted(kernels.items(), key=lambda kv: -kv[1]))
    assert not bad, (len(bad), bad[:5])
    assert n > 30, (n, engines)
    assert engines[S.ENGINE_VM] < 40, engines


def test_nfa_tier_large_stream_closed_form(gpu):
    """A program the step automaton declines (19 list-able threads, > 256 ordered
    lists) over a 1 GiB gen-data stream: ENGINE_AUTO must pick the NFA tier; the
    expected record is a closed form in the length, first checked against the
    oracle on a short stream."""
    ora = harness.OracleEngine()
    pats, tail = [rb"(?:a|b)*a(?:a|b){7}@"], b" abaabaabab@ "
    expect = lambda L: [0, 1, L - 12, L - 1]
    with S.Pool() as pool:
        re = S.parse(pool, pats)
        prog = S.compile(pool, re)
        small = S.gen_data_host(30000, tail)
        first, _ = _expect(ora, prog, re.ncaps, small)
        assert first == expect(len(small)), (first, expect(len(small)))
        big = 1 << 30
        L = S.gen_data_length(big, len(tail))
        buf = S.DeviceBuffer(big)
        assert gpu.sre_hip_gen_data(buf.ptr, L, tail, len(tail), None) == 0
        for mode in (S.HIP_PIKE_FIRST, S.HIP_THOMPSON):
            sc = S.Scanner(pool, prog, mode, S.ENGINE_AUTO)
            assert sc.engine == S.ENGINE_NFA
            rec = sc.scan([buf.ptr], [L])[0]
            assert rec == (expect(L) if mode == S.HIP_PIKE_FIRST else [0, 1, -1, -1]), (mode, rec)
            assert sc.last_fixups == 0
        # no match anywhere: the whole stream is scanned
        assert gpu.sre_hip_gen_data(buf.ptr, L, b"aaabbccb", 8, None) == 0
        rec = S.Scanner(pool, prog, S.HIP_PIKE_FIRST, S.ENGINE_AUTO).scan([buf.ptr], [L])[0]
        assert rec == [S.SRE_DECLINED, 0, -1, -1], rec
        buf.free()


def test_bench_floor_variant_closed_form_vs_oracle(gpu):
    """bench.py's `floor` (a pending match left behind every 3 bytes: the scan kernel's exact path), `floorla`
    (a look-ahead match every 4 bytes) and `words` (a greedy class over words) variants — the last two folded
    into the COUNT table since round 3: the closed forms they assert at 4 GiB, against the oracle at small sizes."""
    ora = harness.OracleEngine()
    for pat, body, last in ((rb"\bfoo\b", b"foo ", (4, 1)), (rb"a(?:bc)?", b"ab ", (3, 2)), (rb"[a-z]+", b"foo bar ", (4, 1))):
        with S.Pool() as pool:
            re = S.parse(pool, [pat])
            prog = S.compile(pool, re)
            sc = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, S.ENGINE_AUTO)
            assert sc.engine == S.ENGINE_SCAN
            for k in (1, 17, 5000):
                data = body * k
                _, cnt = _expect(ora, prog, re.ncaps, data)
                n = len(data)
                per = 2 if body == b"foo bar " else 1
                assert cnt == [0, per * k, n - last[0], n - last[1]], (pat, k, cnt)
                buf = S.DeviceBuffer.from_bytes(data)
                rec = sc.scan([buf.ptr], [n])[0]
                buf.free()
                assert rec == cnt, (pat, k, rec, cnt)


def test_count_restarts_that_depend_on_the_byte_in_front_settle_at_once(gpu):
    """Find-all counts whose every search ends at once and restarts from the list the byte in front selects
    (`\\b` over words, `$` over lines, empty matches at every byte): the lanes' warm-up follows the caller's
    restarts, so the chain check holds on the first pass — with a fixed warm-up seed 64 MiB of "ab cd " took
    131 086 fix-up rounds (151 s; tools/floor_probe.py).  Counts against the oracle at small sizes, closed
    forms and the number of rounds at 16 MiB."""
    ora = harness.OracleEngine()
    cases = [(rb"\b", b"ab cd "), (rb"$", b"ab\n"), (rb"x*", b"abc"), (rb"a*", b"ab"), (rb"^a", b"ab\nab\n"),
             (rb"\bfoo\b", b"foo "), (rb"a+", b"aaab")]
    for pat, body in cases:
        with S.Pool() as pool:
            re = S.parse(pool, [pat])
            prog = S.compile(pool, re)
            sc = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, S.ENGINE_AUTO)
            assert sc.engine == S.ENGINE_SCAN
            for seg in (0, 256):
                if seg:
                    sc.set_segment_bytes(seg)
                for k in (1, 50, 3000):
                    data = body * k
                    _, cnt = _expect(ora, prog, re.ncaps, data)
                    buf = S.DeviceBuffer.from_bytes(data)
                    rec = sc.scan([buf.ptr], [len(data)])[0]
                    buf.free()
                    assert rec == cnt, (pat, seg, k, rec, cnt)
            # 16 MiB: the count scales linearly from the 3000-period sample, and the rounds stay few
            small = body * 3000
            _, c_small = _expect(ora, prog, re.ncaps, small)
            big_k = (16 << 20) // len(body)
            buf = S.DeviceBuffer.from_bytes(body * big_k)
            rec = sc.scan([buf.ptr], [big_k * len(body)])[0]
            buf.free()
            per = (c_small[1] - _expect(ora, prog, re.ncaps, body * 2999)[1][1])
            assert rec[1] == c_small[1] + per * (big_k - 3000), (pat, rec[:4], c_small[:4], per)
            assert sc.last_fixups <= 4, (pat, sc.last_fixups)


def test_newline_flag_exec_vs_reference_goldens(gpu):
    """programs parsed with SRE_REGEX_NEWLINE through the compat API: Thompson status, Pike rc and
    ovector equal the real reference library's (tests/golden/newline_flag.jsonl)."""
    n = 0
    for r in harness.load_jsonl("newline_flag.jsonl"):
        pats = [bytes.fromhex(h) for h in r["re"]]
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, pats, r["flags"]))
            for run in r["runs"]:
                subj = bytes.fromhex(run["s"])
                t = S.ThompsonCtx(pool, prog)
                assert t.exec(subj, True) == run["thompson"], (pats, r["flags"], subj)
                p = S.PikeCtx(pool, prog, r["ncaps"])
                rc = p.exec(subj, True)
                assert rc == run["pike_rc"], (pats, r["flags"], subj, rc)
                if rc >= 0:
                    assert list(p.ovector) == run["pike_ov"], (pats, r["flags"], subj, list(p.ovector))
                n += 1
    assert n >= 125, n


def test_bench_nfa_variants_closed_forms_vs_oracle(gpu):
    """bench.py's NFA-tier variants (nfa37: configs[2]'s program forced onto the tier over a body that
    never matches; nfa60 / nfa57w: declined programs with ~60 list-able threads): the closed forms
    bench.py asserts at 4 GiB, checked against the oracle at a size it finishes, on the kernels the
    bench measures."""
    import bench
    ora = harness.OracleEngine()
    cases = [
        ("nfa37", bench.CFG3, bench.NFA37_BODY, bench.NFA37_TAIL, S.ENGINE_NFA, lambda n, t: [8, 1, n - 4, n, -1, -1]),
        ("nfa60", [bench.NFA60_PAT], b"abccc", bench.NFA60_TAIL, S.ENGINE_AUTO, lambda n, t: [0, 1, n - (len(t) - 1), n - 1]),
        ("nfala", [bench.NFA_PAT + b"$"], b"abccc", b" abaabaabab@", S.ENGINE_AUTO, lambda n, t: [0, 1, n - 11, n]),
        ("nfa57w", [bench.NFA57_PAT], b"abccc", bench.NFA57_TAIL, S.ENGINE_AUTO, lambda n, t: [0, 1, n - (len(t) - 1), n - 1]),
    ]
    for name, pats, body, tail, engine, want in cases:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            for size in (3000, 200000):
                data = body * ((size - len(tail)) // len(body)) + tail
                first, _ = _expect(ora, prog, re.ncaps, data)
                assert first == want(len(data), tail), (name, size, first, want(len(data), tail))
                sc = S.Scanner(pool, prog, S.HIP_PIKE_FIRST, engine)
                assert sc.engine == S.ENGINE_NFA and "nfa_sa" in sc.kernel_name, (name, sc.engine, sc.kernel_name)
                buf = S.DeviceBuffer.from_bytes(data)
                rec = sc.scan([buf.ptr], [len(data)])[0]
                buf.free()
                assert rec == first, (name, size, rec, first)
            print(name, sc.kernel_name)


# ------------------------------------------------------------ randomised differential

@pytest.mark.parametrize("seg", [64, 0])
def test_scanner_random_patterns_vs_oracle(gpu, seg):
    """Differential test: random patterns x random subjects, every mode the
    scanner admits (tiny segments force speculation) and, for seg == 0, the exact
    VM kernels too (SRE_FUZZ_VM=0 leaves them out), against the oracle.  Failing cases are also written to
    gpurun_out/fuzz_fail.jsonl (pattern / subject in hex) for reproduction."""
    import json
    import random
    ora = harness.OracleEngine()
    # SRE_FUZZ_SEED=<n> runs another sequence (exploration outside the fixed suite)
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) + seg)
    alphabet = b"abcx \n_."
    modes = (S.HIP_THOMPSON, S.HIP_PIKE_FIRST, S.HIP_PIKE_COUNT)
    tested = admitted = nfa_admitted = 0
    bad = []
    import time
    slow_log = os.environ.get("SRE_FUZZ_SLOW")          # seconds: patterns slower than that go to gpurun_out/fuzz_slow.jsonl
    for _ in range(600):
        nre = 1 if rng.random() < 0.8 else rng.randrange(2, 4)
        pats = [harness.random_regex(rng) for _ in range(nre)]
        t_pat = time.perf_counter()
        if slow_log and bad is not None and getattr(test_scanner_random_patterns_vs_oracle, "_last", None):
            last_pats, last_t = test_scanner_random_patterns_vs_oracle._last
            if t_pat - last_t > float(slow_log):
                os.makedirs(os.path.join(harness.ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(harness.ROOT, "gpurun_out", "fuzz_slow.jsonl"), "a") as f:
                    f.write(json.dumps({"seg": seg, "seconds": t_pat - last_t, "re": [p.hex() for p in last_pats],
                                        "text": [p.decode("latin-1") for p in last_pats]}) + "\n")
        test_scanner_random_patterns_vs_oracle._last = (pats, t_pat)
        with S.Pool() as pool:
            try:
                re = S.parse(pool, pats)
            except Exception:
                continue
            prog = S.compile(pool, re)
            engines = {}
            for mode in modes:
                try:
                    sc = S.Scanner(pool, prog, mode, S.ENGINE_SCAN)
                except RuntimeError:
                    continue
                if seg:
                    sc.set_segment_bytes(int(os.environ.get("SRE_FUZZ_SEG", seg)))
                engines[("scan", mode)] = sc
            for mode in (S.HIP_THOMPSON, S.HIP_PIKE_FIRST, S.HIP_PIKE_COUNT):
                # the bit-parallel NFA tier, forced (it is chosen by itself only when the
                # step automaton declines): set pass + exact VM window.  (Not the look-ahead
                # programs the builder declines: the reference VM itself may diverge on them,
                # DESIGN.md 5.)
                if ("scan", S.HIP_PIKE_FIRST) not in engines and prog_has_lookahead(pats):
                    continue
                try:
                    sc = S.Scanner(pool, prog, mode, S.ENGINE_NFA)
                except RuntimeError:
                    continue
                if seg:
                    sc.set_segment_bytes(int(os.environ.get("SRE_FUZZ_SEG", seg)))
                engines[("nfa", mode)] = sc
                nfa_admitted += 1
            tested += 1
            admitted += 1 if engines else 0
            if seg == 0 and os.environ.get("SRE_FUZZ_VM", "1") != "0" and \
                    (("scan", S.HIP_PIKE_FIRST) in engines or not prog_has_lookahead(pats)):
                # the exact VM kernels take every program: same subjects, same expectations
                # (not the look-ahead programs the builder declines: the reference VM itself may
                # diverge on them)
                for mode in modes:
                    engines[("vm", mode)] = S.Scanner(pool, prog, mode, S.ENGINE_VM)
            if not engines:
                continue
            sizes = [0, 1, 7, 64, 65, 130, 400] + ([1500, 5000] if os.environ.get("SRE_FUZZ_BIG") else [])
            datas = [bytes(rng.choice(alphabet) for _ in range(rng.choice(sizes))) for _ in range(6)]
            # streams start at arbitrary byte offsets (unaligned rows in the staging loads)
            offs = [rng.randrange(0, 16) for _ in datas]
            bufs = [S.DeviceBuffer.from_bytes(b"#" * o + d) for o, d in zip(offs, datas)]
            ptrs, lens = [b.ptr + o for b, o in zip(bufs, offs)], [len(d) for d in datas]
            got = {key: sc.scan(ptrs, lens) for key, sc in engines.items()}
            for i, d in enumerate(datas):
                first, cnt = _expect(ora, prog, re.ncaps, d)
                t = ora.thompson(prog)
                th = t.exec(d, True)
                t.close()
                for (eng, mode), recs in got.items():
                    want = first if mode == S.HIP_PIKE_FIRST else cnt if mode == S.HIP_PIKE_COUNT else None
                    if want is None and th == S.SRE_ERROR:
                        continue            # the reference's Thompson list overflows here (oracle guard)
                    ok = recs[i][0] == th if want is None else recs[i] == want
                    if not ok:
                        bad.append({"engine": eng, "mode": mode, "seg": seg, "re": [p.hex() for p in pats],
                                    "s": d.hex(), "got": recs[i], "want": want if want is not None else [th]})
            for b in bufs:
                b.free()
    if bad:
        os.makedirs(os.path.join(harness.ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(harness.ROOT, "gpurun_out", "fuzz_fail.jsonl"), "a") as f:
            for rec in bad:
                f.write(json.dumps(rec) + "\n")
    assert admitted > 400, (tested, admitted)
    assert nfa_admitted > 300, nfa_admitted
    assert not bad, (len(bad), [(b["engine"], b["mode"], bytes.fromhex(b["re"][0]), b["got"][:4], b["want"][:4])
                               for b in bad[:6]])


def test_exact_vm_stable_runs_vs_oracle(gpu):
    """Stable runs on the exact VM (sre_hip_pwave.hip / thompson_wave_run): subjects made of long runs,
    over which most lists loop in place and the wave skips 512 bytes at a time — ENGINE_VM scans in all
    three modes, and sre_vm_pike_exec / sre_vm_thompson_exec fed in chunks that stay on the VM (a short
    first chunk), against the oracle.  tests/test_pwave_model.py checks the same on the CPU model."""
    import random
    ora = harness.OracleEngine()
    eng = harness.ProductEngine()
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) + 77)
    alphabet = b"abcx \n_."

    def runs(total):
        out = bytearray()
        while len(out) < total:
            out += bytes([rng.choice(alphabet)]) * rng.choice([1, 1, 2, 3, 9, 30, 70, 150, 700])
        return bytes(out[:total])

    zoo = [[rb"[a-z]+@[a-z]+\.[a-z]+"], [rb"([a-z]+)@([a-z]+)\.[a-z]+"], [rb"a+b"], [rb"(a+)(b+)?"], [rb"x(.*)y(.*)z"],
           [rb"^a+c"], [rb"(?:a|b)*c"], [rb"a.*b"], [rb"(a*)*x"], [rb"\n+a"], [rb"[^x]+x", rb"a+_"], [rb"(?:aa)+b"],
           [rb"(a|ab)(c|bcd)(d*)"], [rb"a{3,}b"], [rb"(?:.|\n)*x"], [rb"\s+\S"], [rb"(?:a|b)*a(?:a|b){7}_"]]
    progs = zoo + [[harness.random_regex(rng) for _ in range(1 if rng.random() < 0.8 else 2)] for _ in range(220)]
    modes = (S.HIP_THOMPSON, S.HIP_PIKE_FIRST, S.HIP_PIKE_COUNT)
    bad, n, fed = [], 0, 0
    for k, pats in enumerate(progs):
        if prog_has_lookahead(pats):
            continue            # the one-lane kernels (no stable runs there)
        with S.Pool() as pool:
            try:
                re = S.parse(pool, pats)
            except Exception:
                continue
            prog = S.compile(pool, re)
            nov = 2 * (re.ncaps + 1)
            datas = [runs(rng.choice([300, 2000, 6000])) for _ in range(4)]
            bufs = [S.DeviceBuffer.from_bytes(d) for d in datas]
            got = {m: S.Scanner(pool, prog, m, S.ENGINE_VM).scan([b.ptr for b in bufs], [len(d) for d in datas]) for m in modes}
            for i, d in enumerate(datas):
                first, cnt = _expect(ora, prog, re.ncaps, d)
                t = ora.thompson(prog)
                th = t.exec(d, True)
                t.close()
                n += 1
                if got[S.HIP_PIKE_FIRST][i] != first or got[S.HIP_PIKE_COUNT][i] != cnt or \
                        (th != S.SRE_ERROR and got[S.HIP_THOMPSON][i][0] != th):
                    bad.append((pats, d[:80], got[S.HIP_PIKE_FIRST][i][:4], first[:4], got[S.HIP_PIKE_COUNT][i][:4], cnt[:4],
                                got[S.HIP_THOMPSON][i][0], th))
            for b in bufs:
                b.free()
            if k < 60:
                # chunks on the VM: the first one is short, so the stream starts (and stays) there
                d = datas[1]
                sizes = [rng.choice([60, 100, 200]), 700, 64, 1500, 1, 900]
                want = _feed(ora.pike(prog, re.ncaps), d, sizes, nov)
                gotf = _feed(eng.pike(prog, re.ncaps), d, sizes, nov)
                fed += 1
                if gotf != want:
                    bad.append((pats, "pike chunks", sizes, gotf[-2:], want[-2:]))
                to, tp = ora.thompson(prog), eng.thompson(prog)
                off, rcs = 0, []
                for sz in sizes + [len(d)]:
                    chunk = d[off:off + sz]
                    off += len(chunk)
                    eof = off >= len(d)
                    rcs.append((to.exec(chunk, eof), tp.exec(chunk, eof)))
                    if rcs[-1][0] != S.SRE_AGAIN or eof:
                        break
                to.close()
                if any(a != b for a, b in rcs if a != S.SRE_ERROR):
                    bad.append((pats, "thompson chunks", sizes, rcs))
                eng.recycle()
    eng.pool.destroy()
    assert not bad, (len(bad), bad[:3])
    assert n > 400 and fed > 30, (n, fed)


def test_recorded_fuzz_regressions(gpu):
    """Cases the randomised tests once failed on (tests/golden/fuzz_regressions.jsonl:
    engine, mode, segment size, patterns and subject in hex) stay fixed."""
    import json
    ora = harness.OracleEngine()
    n = 0
    for line in open(os.path.join(harness.ROOT, "tests", "golden", "fuzz_regressions.jsonl")):
        c = json.loads(line)
        pats = [bytes.fromhex(x) for x in c["re"]]
        d = bytes.fromhex(c["s"])
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            first, cnt = _expect(ora, prog, re.ncaps, d)
            want = first if c["mode"] == S.HIP_PIKE_FIRST else cnt
            sc = S.Scanner(pool, prog, c["mode"], {"vm": S.ENGINE_VM, "nfa": S.ENGINE_NFA}.get(c["engine"], S.ENGINE_SCAN))
            if c["engine"] != "vm" and c["seg"]:
                sc.set_segment_bytes(c["seg"])
            buf = S.DeviceBuffer.from_bytes(d)
            got = sc.scan([buf.ptr], [len(d)])[0]
            buf.free()
            assert got == want, (c["engine"], c["mode"], pats, d.hex())
            n += 1
    assert n >= 40


def test_compat_api_random_patterns_vs_oracle(gpu):
    """The unchanged C API (sre_vm_pike_exec / sre_vm_thompson_exec) on random
    patterns: whole-buffer and byte-at-a-time calls exactly as the reference CLI
    makes them (temp captures and pending matches of every AGAIN included),
    against the oracle driven through the same sequence."""
    import random
    ora = harness.OracleEngine()
    eng = harness.ProductEngine()
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) * 7 + 1)
    alphabet = b"abcx \n_."
    bad = []
    n = 0
    for _ in range(120):
        nre = 1 if rng.random() < 0.8 else rng.randrange(2, 4)
        pats = [harness.random_regex(rng) for _ in range(nre)]
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            # programs on which the reference VM itself diverges have no expectation
            try:
                S.Scanner(pool, prog, S.HIP_PIKE_FIRST, S.ENGINE_SCAN)
            except RuntimeError:
                if prog_has_lookahead(pats):
                    continue
            for _ in range(2):
                d = bytes(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 5, 17, 40])))
                want = harness.cli_lines(ora, prog, d, re.ncaps)
                got = harness.cli_lines(eng, prog, d, re.ncaps)
                n += 1
                if got != want:
                    bad.append((pats, d, got, want))
            eng.recycle()
    assert n > 150, n
    assert not bad, (len(bad), bad[:3])


def prog_has_lookahead(pats):
    return any(tok in p for p in pats for tok in (b"$", b"\\b", b"\\B", b"\\z"))
