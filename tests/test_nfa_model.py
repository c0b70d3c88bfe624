"""The bit-parallel form of a program (sregex_amd/csrc/sre_nfa.cpp), checked on the
CPU through a test-only sequential model (tests/nfa_sim.cpp) against the
reference goldens and the oracle.  Pins the ALGORITHM of the NFA tier: thread
sets decide Thompson exactly, and for Pike the first MATCH event and the clean
position in front of it bracket the reference's match."""
import ctypes
import os
import random
import subprocess

import pytest

import sregex_amd as S
import harness

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_vp, _i64 = ctypes.c_void_p, ctypes.c_int64


@pytest.fixture(scope="module")
def sim(lib):
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libnfasim.so")
    srcs = [os.path.join(HERE, "nfa_sim.cpp"), os.path.join(ROOT, "sregex_amd", "csrc", "sre_nfa.cpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-shared", "-fPIC", "-o", so] + srcs +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "sregex_amd", "csrc")])
    L = ctypes.CDLL(so)
    L.nfa_sim_build.restype = _vp
    L.nfa_sim_build.argtypes = [_vp, ctypes.POINTER(ctypes.c_char_p)]
    L.nfa_sim_free.argtypes = [_vp]
    L.nfa_sim_nbits.argtypes = [_vp]
    L.nfa_sim_nbits.restype = ctypes.c_uint32
    L.nfa_sim_run.argtypes = [_vp, ctypes.c_char_p, _i64, ctypes.c_int, ctypes.POINTER(_i64)]
    L.nfa_sim_build2.restype = _vp
    L.nfa_sim_build2.argtypes = [_vp, ctypes.c_uint, ctypes.POINTER(ctypes.c_char_p)]
    L.nfa_sim_sa_info.argtypes = [_vp, ctypes.POINTER(ctypes.c_int32)]
    L.nfa_sim_run_sa.argtypes = [_vp, ctypes.c_char_p, _i64, ctypes.c_int, ctypes.POINTER(_i64)]
    L.nfa_sim_walk_set.restype = ctypes.c_uint64
    L.nfa_sim_walk_set.argtypes = [_vp, ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p, _i64, ctypes.c_uint32]
    L.nfa_sim_valid_bits.restype = ctypes.c_uint64
    L.nfa_sim_valid_bits.argtypes = [_vp, ctypes.c_int]
    L.nfa_sim_thompson_wave.restype = _i64
    L.nfa_sim_thompson_wave.argtypes = [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, _i64, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(_i64)]
    L.nfa_sim_init0.restype = ctypes.c_uint64
    L.nfa_sim_init0.argtypes = [_vp]
    L.nfa_sim_nassert.argtypes = [_vp]
    return L


def _run(sim, h, data):
    out = (_i64 * 3)()
    sim.nfa_sim_run(h, bytes(data), len(data), 0, out)
    return out[0], out[1]


def _check(sim, ora, prog, ncaps, data):
    """-> None when the program has no bit-parallel form, else a list of complaints"""
    why = ctypes.c_char_p()
    h = sim.nfa_sim_build(prog.h, ctypes.byref(why))
    if not h:
        return None
    bad = []
    ev, clean = _run(sim, h, data)
    t = ora.thompson(prog)
    th = t.exec(data, True)
    t.close()
    if (ev >= 0) != (th == 0):
        bad.append(("thompson", ev, th))
    p = ora.pike(prog, ncaps)
    rc = p.exec(data, True, want_pending=False)
    ov = list(p.ovector)
    p.close()
    if (rc >= 0) != (ev >= 0):
        bad.append(("pike rc", ev, rc))
    if rc >= 0 and ev >= 0:
        # the reference's match starts at or behind the clean position and ends
        # behind the first event; the oracle restarted AT the clean position (a
        # fresh search sees the same list there) finds the same match
        if not (clean <= ov[0] and ev < max(ov[1], ev + 1) and ev + 1 <= max(ov[1], ev + 1)):
            bad.append(("bracket", ev, clean, ov[:2]))
        if clean > ov[0]:
            bad.append(("clean behind the match start", clean, ov[:2]))
    sim.nfa_sim_free(h)
    return bad


def test_sets_decide_thompson_and_bracket_pike_on_reference_blocks(sim, blocks):
    ora = harness.OracleEngine()
    admitted, bad = 0, []
    for blk in blocks:
        subject = bytes.fromhex(blk["s"])
        for name, regexes, flags, multi, ref in harness.block_variants(blk):
            if ref["rc"] != 0:
                continue
            with S.Pool() as pool:
                prog = S.compile(pool, S.parse(pool, regexes, flags, multi))
                r = _check(sim, ora, prog, ref["ncaps"], subject)
                if r is None:
                    continue
                admitted += 1
                if r:
                    bad.append((blk["file"], blk["name"], name, r))
    assert admitted > 3000, admitted
    assert not bad, (len(bad), bad[:5])


def test_sets_random_patterns_vs_oracle(sim):
    ora = harness.OracleEngine()
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) + 5)
    alphabet = b"abcx \n_."
    admitted, bad = 0, []
    for _ in range(1500):
        nre = 1 if rng.random() < 0.8 else rng.randrange(2, 4)
        pats = [harness.random_regex(rng) for _ in range(nre)]
        with S.Pool() as pool:
            try:
                re = S.parse(pool, pats)
            except Exception:
                continue
            prog = S.compile(pool, re)
            for _ in range(4):
                d = bytes(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 7, 40, 130, 400])))
                r = _check(sim, ora, prog, re.ncaps, d)
                if r is None:
                    break
                admitted += 1
                if r:
                    bad.append((pats, d, r))
    assert admitted > 2000, admitted
    assert not bad, (len(bad), bad[:5])


# ---------------------------------------------------------------- the shift-and form

SA_MASKED, SA_EVACC, SA_W64, SA_CARRY, SA_NO_MERGE, SA_EXPLICIT_ANY, SA_NO_EVACC = 1, 2, 4, 8, 16, 32, 64
SA_OPTIONS = [0, SA_MASKED, SA_EVACC, SA_NO_EVACC, SA_W64, SA_W64 | SA_CARRY, SA_NO_MERGE | SA_EXPLICIT_ANY,
              SA_MASKED | SA_EVACC | SA_W64 | SA_CARRY, SA_NO_EVACC | SA_W64 | SA_EXPLICIT_ANY]


def _sa_info(sim, h):
    info = (ctypes.c_int32 * 10)()
    sim.nfa_sim_sa_info(h, info)
    return dict(zip(("has", "nbits", "w64", "carry", "masked", "evacc", "nlut", "cost", "threads", "nassert"), info))


def _sa_check(sim, prog, datas, options):
    """-> (info of the default form or None, complaints): the shift-and form against the plain form, step by step"""
    bad, first = [], None
    for opt in options:
        why = ctypes.c_char_p()
        h = sim.nfa_sim_build2(prog.h, opt, ctypes.byref(why))
        if not h:
            return None, bad
        info = _sa_info(sim, h)
        if opt == 0:
            first = info
        if info["has"]:
            for d in datas:
                for variant in (0, 1, 2):
                    out = (_i64 * 3)()
                    sim.nfa_sim_run_sa(h, bytes(d), len(d), variant, out)
                    if out[2] >= 0:
                        bad.append((opt, variant, info, d[:80], out[2]))
                    ref = (_i64 * 3)()
                    sim.nfa_sim_run(h, bytes(d), len(d), variant, ref)
                    if (ref[0], ref[1]) != (out[0], out[1]):
                        bad.append(("result", opt, variant, d[:80], tuple(ref)[:2], tuple(out)[:2]))
        sim.nfa_sim_free(h)
    return first, bad


def test_shift_and_form_equals_the_plain_form_on_reference_blocks(sim, blocks):
    """Every reference program without look-ahead assertions, every build option that selects
    another kernel variant: the shift-and step lists exactly the threads the plain slices list
    (through the bit map), sees the same events and the same clean positions."""
    rng = random.Random(7)
    alphabet = b"abcx \n_.@/:"
    stats, bad, n = {}, [], 0
    seen = set()
    for blk in blocks:
        subject = bytes.fromhex(blk["s"])
        for name, regexes, flags, multi, ref in harness.block_variants(blk):
            if ref["rc"] != 0 or (tuple(regexes), tuple(flags)) in seen:
                continue
            seen.add((tuple(regexes), tuple(flags)))
            with S.Pool() as pool:
                prog = S.compile(pool, S.parse(pool, regexes, flags, multi))
                datas = [subject, subject * 3 + b"\n" + subject,
                         bytes(rng.choice(alphabet) for _ in range(200))]
                info, r = _sa_check(sim, prog, datas, SA_OPTIONS)
                if info is None:
                    continue
                n += 1
                key = (info["has"], info["w64"], info["nlut"]) if info["has"] else (0,)
                stats[key] = stats.get(key, 0) + 1
                if r:
                    bad.append((regexes, r[:2]))
    print("shift-and forms over the reference programs (has, w64, nlut):", sorted(stats.items()))
    assert n > 1000, n
    assert not bad, (len(bad), bad[:3])


def test_shift_and_form_random_patterns(sim):
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) + 11)
    alphabet = b"abcx \n_."
    n, bad, forms = 0, [], 0
    for _ in range(1200):
        nre = 1 if rng.random() < 0.8 else rng.randrange(2, 4)
        pats = [harness.random_regex(rng) for _ in range(nre)]
        with S.Pool() as pool:
            try:
                re = S.parse(pool, pats)
            except Exception:
                continue
            prog = S.compile(pool, re)
            datas = [bytes(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 7, 40, 130, 400]))) for _ in range(4)]
            info, r = _sa_check(sim, prog, datas, SA_OPTIONS)
            if info is None:
                continue
            n += 1
            forms += info["has"]
            if r:
                bad.append((pats, r[:2]))
    assert n > 500 and forms > 300, (n, forms)
    assert not bad, (len(bad), bad[:3])


def test_shift_and_form_of_the_bench_programs(sim):
    """the programs of bench.py's NFA variants: what the builder makes of them"""
    cases = {
        "nfa": ([rb"(?:a|b)*a(?:a|b){7}@"], dict(w64=0, nlut=1)),
        "nfa37": ([b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"], dict(w64=0, nlut=0)),
    }
    for name, (pats, want) in cases.items():
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, pats))
            why = ctypes.c_char_p()
            h = sim.nfa_sim_build2(prog.h, 0, ctypes.byref(why))
            info = _sa_info(sim, h)
            sim.nfa_sim_free(h)
            print(name, info)
            assert info["has"], name
            for k, v in want.items():
                assert info[k] <= v, (name, k, info)


def test_a_segment_is_a_union_homomorphism_of_its_entry_set(sim):
    """What the exact-entry fallback of the NFA tier rests on (sre_hip_nfa.hip sre_k_nfa_seg_matrix /
    sre_k_nfa_exact_entries): stepping a thread set over a segment distributes over unions, so a segment's effect on
    ANY entry set follows from its effect on the singletons — F(B u M) = F(B) u U_{i in M} F({i}).  Checked for the
    plain slices and the shift-and form under every build option (32 / 64 bits, carry, masked, event accumulation,
    explicit `.*?` thread), with and without look-ahead assertions, over random sets and segments."""
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) + 41)
    alphabet = b"abcx \n_.y@"
    zoo = [[rb"(?:a|b)*a(?:a|b){7}@"], [rb"x[^y]*y(?:a|b)*a(?:a|b){7}@"], [rb"x.*y(?:a|b)*a(?:a|b){7}@$"],
           [rb"(?:a|b)*a[ab]{20}c[^x]{30}@"], [rb"\b(?:a|b)*a(?:a|b){5}\b"], [rb"^x[^y]*y[ab]{9}"],
           [rb"a", rb"ab", rb"c", rb"a(bc)", rb"e(f)", rb"gh", rb"A", rb"b", rb"BLAH", rb"\s+", rb"abcd", rb"bc"]]
    progs = zoo + [[harness.random_regex(rng) for _ in range(1 if rng.random() < 0.8 else 2)] for _ in range(300)]
    n = forms = 0
    bad = []
    for pats in progs:
        with S.Pool() as pool:
            try:
                re = S.parse(pool, pats)
            except Exception:
                continue
            prog = S.compile(pool, re)
            for opts in (0, 128, 1, 2 + 16, 4, 4 + 8, 32 + 64):
                why = ctypes.c_char_p()
                h = sim.nfa_sim_build2(prog.h, opts, ctypes.byref(why))
                if not h:
                    continue
                info = (ctypes.c_int32 * 10)()
                sim.nfa_sim_sa_info(h, info)
                for sa in ((0, 1) if info[0] else (0,)):
                    valid = sim.nfa_sim_valid_bits(h, sa)
                    forms += 1
                    for _ in range(6):
                        seg = bytes(rng.choice(alphabet) for _ in range(rng.choice([1, 5, 64, 200])))
                        prev = rng.randrange(4)
                        B = rng.getrandbits(64) & valid & rng.getrandbits(64)
                        M = rng.getrandbits(64) & valid & rng.getrandbits(64) & rng.getrandbits(64)
                        whole = sim.nfa_sim_walk_set(h, sa, B | M, seg, len(seg), prev)
                        parts = sim.nfa_sim_walk_set(h, sa, B, seg, len(seg), prev)
                        for i in range(64):
                            if (M >> i) & 1:
                                parts |= sim.nfa_sim_walk_set(h, sa, 1 << i, seg, len(seg), prev)
                        n += 1
                        if whole != parts:
                            bad.append((pats, opts, sa, seg[:30], hex(B), hex(M), hex(whole), hex(parts)))
                sim.nfa_sim_free(h)
    assert not bad, (len(bad), bad[:3])
    assert n > 3000 and forms > 400, (n, forms)


def test_thompson_wave_with_stable_runs_vs_oracle(sim):
    """The Thompson wave kernel (sre_hip_vm.hip thompson_wave_run: lanes = threads, the live set a 64-bit mask, one
    ballot per byte) with its stable runs — a byte that maps the set to itself joins a byte set kept for that set,
    the bytes of the set that follow are skipped inside the block and 512 at a time across blocks — modelled on the
    CPU (tests/nfa_sim.cpp nfa_sim_thompson_wave) with the skipping on and off, against the oracle's Thompson VM:
    whole buffers, and buffers fed in chunks with the mask as the context (programs without ^: this VM's ^ is
    local to the buffer of a call)."""
    ora = harness.OracleEngine()
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) + 43)
    alphabet = b"abcx \n_."

    def runs(total):
        out = bytearray()
        while len(out) < total:
            out += bytes([rng.choice(alphabet)]) * rng.choice([1, 1, 2, 3, 9, 30, 70, 150, 700])
        return bytes(out[:total])

    zoo = [[rb"[a-z]+@[a-z]+\.[a-z]+"], [rb"a+b"], [rb"x(.*)y(.*)z"], [rb"(?:a|b)*c"], [rb"a.*b"], [rb"(a*)*x"], [rb"\n+a"],
           [rb"[^x]+x", rb"a+_"], [rb"(?:aa)+b"], [rb"a{3,}b"], [rb"(?:.|\n)*x"], [rb"\s+\S"], [rb"(?:a|b)*a(?:a|b){7}_"]]
    progs = zoo + [[harness.random_regex(rng) for _ in range(1 if rng.random() < 0.8 else 2)] for _ in range(500)]
    n = chunked = 0
    skipped = _i64(0)
    bad = []
    for pats in progs:
        with S.Pool() as pool:
            try:
                re = S.parse(pool, pats)
            except Exception:
                continue
            prog = S.compile(pool, re)
            why = ctypes.c_char_p()
            h = sim.nfa_sim_build(prog.h, ctypes.byref(why))
            if not h:
                continue
            if sim.nfa_sim_nassert(h):
                sim.nfa_sim_free(h)
                continue                # (look-ahead programs keep the scalar VM)
            for _ in range(4):
                d = runs(rng.choice([1, 40, 300, 2000, 5000])) if rng.random() < 0.8 else b""
                t = ora.thompson(prog)
                want = t.exec(d, True)
                t.close()
                if want == S.SRE_ERROR:
                    continue            # the reference's thread list overflows here (oracle guard)
                for ff in (1, 0):
                    m = ctypes.c_uint64(sim.nfa_sim_init0(h))
                    got = sim.nfa_sim_thompson_wave(h, ctypes.byref(m), d, len(d), 1, ff, ctypes.byref(skipped))
                    n += 1
                    if got != want:
                        bad.append((pats, d[:60], ff, got, want))
                if not any(b"^" in p or b"\\A" in p for p in pats):
                    sizes = [rng.choice([0, 1, 5, 64, 65, 200, 700]) for _ in range(rng.randrange(1, 8))]
                    t = ora.thompson(prog)
                    m = ctypes.c_uint64(sim.nfa_sim_init0(h))
                    off, rcs = 0, []
                    for sz in sizes + [len(d)]:
                        chunk = d[off:off + sz]
                        off += len(chunk)
                        eof = 1 if off >= len(d) else 0
                        a = t.exec(chunk, bool(eof))
                        b = sim.nfa_sim_thompson_wave(h, ctypes.byref(m), chunk, len(chunk), eof, 1, ctypes.byref(skipped))
                        rcs.append((a, b))
                        if a != S.SRE_AGAIN or eof:
                            break
                    t.close()
                    chunked += 1
                    if any(a != b for a, b in rcs if a != S.SRE_ERROR):
                        bad.append((pats, d[:60], "chunks", sizes, rcs))
            sim.nfa_sim_free(h)
    assert not bad, (len(bad), bad[:3])
    assert n > 1500 and chunked > 500, (n, chunked)
    assert skipped.value > 100000, skipped.value
