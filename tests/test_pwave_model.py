"""The wavefront Pike step's ALGORITHM (sregex_amd/csrc/sre_pwave.cpp static closure lists +
sre_hip_pwave.hip's step), checked on the CPU through a test-only sequential model
(tests/pwave_sim.cpp) against the reference goldens and the oracle: rc, regex id and every ovector
slot of a whole-buffer exec, for every reference program that has the wave form."""
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
    so = os.path.join(out, "libpwavesim.so")
    srcs = [os.path.join(HERE, "pwave_sim.cpp"), os.path.join(ROOT, "sregex_amd", "csrc", "sre_pwave.cpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-shared", "-fPIC", "-o", so] + srcs +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "sregex_amd", "csrc")])
    L = ctypes.CDLL(so)
    L.pwave_sim_build.restype = _vp
    L.pwave_sim_build.argtypes = [_vp]
    L.pwave_sim_free.argtypes = [_vp]
    L.pwave_sim_exec.restype = _i64
    L.pwave_sim_exec.argtypes = [_vp, ctypes.c_char_p, _i64, ctypes.POINTER(_i64), ctypes.c_uint32, ctypes.POINTER(ctypes.c_int)]
    L.pwave_sim_count.argtypes = [_vp, ctypes.c_char_p, _i64, ctypes.POINTER(_i64), ctypes.c_uint32]
    L.pwave_sim_ctx_new.restype = _vp
    L.pwave_sim_ctx_new.argtypes = [_vp]
    L.pwave_sim_ctx_free.argtypes = [_vp]
    L.pwave_sim_ctx_exec.restype = _i64
    L.pwave_sim_ctx_exec.argtypes = [_vp, ctypes.c_char_p, _i64, ctypes.c_int, ctypes.POINTER(_i64), ctypes.c_uint32,
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_i64)]
    L.pwave_sim_set_ff.argtypes = [ctypes.c_int]
    L.pwave_sim_ff_bytes.restype = _i64
    return L


def _exec(sim, h, data, ncaps):
    nov = 2 * (ncaps + 1)
    ov = (_i64 * nov)(*([-7] * nov))
    poisoned = ctypes.c_int(0)
    rc = sim.pwave_sim_exec(h, bytes(data), len(data), ov, nov, ctypes.byref(poisoned))
    return rc, list(ov) if rc >= 0 else None, poisoned.value


def _oracle(ora, prog, ncaps, data):
    p = ora.pike(prog, ncaps)
    rc = p.exec(data, True, want_pending=False)
    ov = list(p.ovector) if rc >= 0 else None
    # a poisoned context answers SRE_ERROR to the next exec (sre_vm_pike.c:616-622)
    poisoned = 1 if rc >= 0 and p.exec(b"", True, want_pending=False) == S.SRE_ERROR else 0
    p.close()
    return rc, ov, poisoned


def test_wave_step_equals_the_reference_on_reference_blocks(sim, blocks):
    ora = harness.OracleEngine()
    n, bad = 0, []
    for blk in blocks:
        subject = bytes.fromhex(blk["s"])
        for name, regexes, flags, multi, ref in harness.block_variants(blk):
            if ref["rc"] != 0:
                continue
            with S.Pool() as pool:
                prog = S.compile(pool, S.parse(pool, regexes, flags, multi))
                h = sim.pwave_sim_build(prog.h)
                if not h:
                    continue            # look-ahead assertions, > 64 threads or slots: the one-lane VM
                got = _exec(sim, h, subject, ref["ncaps"])
                sim.pwave_sim_free(h)
                nov = 2 * (ref["ncaps"] + 1)
                line = ("pike match %d%s" % (got[0], harness._fmt_caps(got[1], nov)) if got[0] >= 0 else "pike no match")
                n += 1
                if line != ref["res"][4]:
                    bad.append((blk["file"], blk["name"], name, line, ref["res"][4]))
    assert not bad, (len(bad), bad[:5])
    assert n > 1400, n


def test_wave_step_random_patterns_vs_oracle(sim):
    ora = harness.OracleEngine()
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) + 21)
    alphabet = b"abcx \n_."
    n, bad = 0, []
    for _ in range(1500):
        nre = 1 if rng.random() < 0.8 else rng.randrange(2, 4)
        pats = [harness.random_regex(rng) for _ in range(nre)]
        with S.Pool() as pool:
            try:
                re = S.parse(pool, pats)
            except Exception:
                continue
            prog = S.compile(pool, re)
            h = sim.pwave_sim_build(prog.h)
            if not h:
                continue
            for _ in range(4):
                d = bytes(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 7, 40, 130, 400])))
                got = _exec(sim, h, d, re.ncaps)
                want = _oracle(ora, prog, re.ncaps, d)
                n += 1
                if got != want:
                    bad.append((pats, d, got, want))
            sim.pwave_sim_free(h)
    assert not bad, (len(bad), bad[:3])
    assert n > 1500, n


def _count_want(ora, prog, ncaps, data):
    """the record the batched API's PIKE_COUNT returns (tests/test_gpu_parity.py _expect)"""
    nov = 2 * (ncaps + 1)
    allm = harness.findall(ora, prog, ncaps, data)
    final, matches = allm[-1][0], allm[:-1]
    if matches:
        return [S.SRE_ERROR if final == S.SRE_ERROR else matches[-1][0], len(matches)] + matches[-1][1:]
    return [final, 0] + [-1] * nov


def test_wave_find_all_iteration_vs_oracle(sim):
    """the re-armed context of the find-all iteration (empty matches skip a byte, ^ goes by the byte in
    front of the previous match's end, a poisoned context ends it with SRE_ERROR)"""
    ora = harness.OracleEngine()
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) + 22)
    alphabet = b"abcx \n_."
    cases = []
    for rec in harness.load_jsonl("findall.jsonl"):
        cases.append(([bytes.fromhex(h) for h in rec["re"]], bytes.fromhex(rec["s"])))
    for _ in range(600):
        nre = 1 if rng.random() < 0.8 else rng.randrange(2, 4)
        pats = [harness.random_regex(rng) for _ in range(nre)]
        cases.append((pats, bytes(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 7, 40, 130])))))
    n, bad = 0, []
    for pats, data in cases:
        with S.Pool() as pool:
            try:
                re = S.parse(pool, pats)
            except Exception:
                continue
            prog = S.compile(pool, re)
            h = sim.pwave_sim_build(prog.h)
            if not h:
                continue
            nov = 2 * (re.ncaps + 1)
            rec = (_i64 * (2 + nov))()
            sim.pwave_sim_count(h, bytes(data), len(data), rec, nov)
            sim.pwave_sim_free(h)
            want = _count_want(ora, prog, re.ncaps, data)
            n += 1
            if list(rec) != want:
                bad.append((pats, data, list(rec), want))
    assert not bad, (len(bad), bad[:3])
    assert n > 300, n


def _feed_sim(sim, h, data, sizes, nov):
    """the chunked call sequence of tests/test_gpu_parity.py _feed on the model"""
    c = sim.pwave_sim_ctx_new(h)
    out, off = [], 0
    sizes = list(sizes)
    ov = (_i64 * max(nov, 2))(*([0] * max(nov, 2)))
    pend = (_i64 * 2)()
    hp = ctypes.c_int(0)
    keep = []
    while True:
        n = sizes.pop(0) if sizes else len(data) - off
        n = min(n, len(data) - off)
        eof = off + n >= len(data) and not sizes
        chunk = bytes(data[off:off + n])
        keep.append(chunk)
        rc = sim.pwave_sim_ctx_exec(c, chunk, n, 1 if eof else 0, ov, nov, ctypes.byref(hp), pend)
        off += n
        if rc == S.SRE_AGAIN:
            out.append((rc, tuple(ov[:2]), (pend[0], pend[1]) if hp.value else None))
            continue
        out.append((rc, tuple(ov[:nov]) if rc >= 0 else None, None))
        break
    sim.pwave_sim_ctx_free(c)
    return out


def _feed_oracle(ctx, data, sizes, nov):
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


def test_wave_step_chunked_feeding_vs_oracle(sim, blocks):
    """sre_vm_pike_exec fed in chunks down to one byte per call (the CLI's "splitted pike" mode and
    random chunkings): SRE_AGAIN with its temporary match range and pending match (sre_vm_pike.c:640-735),
    then the match — on the model of the wavefront step, against the oracle fed the same way."""
    ora = harness.OracleEngine()
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) + 23)
    n, bad = 0, []
    for blk in blocks[::3]:
        subject = bytes.fromhex(blk["s"])
        for name, regexes, flags, multi, ref in harness.block_variants(blk):
            if ref["rc"] != 0:
                continue
            with S.Pool() as pool:
                prog = S.compile(pool, S.parse(pool, regexes, flags, multi))
                h = sim.pwave_sim_build(prog.h)
                if not h:
                    continue
                nov = 2 * (ref["ncaps"] + 1)
                for sizes in ([1] * len(subject), [rng.choice([0, 1, 2, 3, 7]) for _ in range(12)]):
                    want = _feed_oracle(ora.pike(prog, ref["ncaps"]), subject, sizes, nov)
                    got = _feed_sim(sim, h, subject, sizes, nov)
                    n += 1
                    if got != want:
                        bad.append((regexes, subject[:40], sizes[:8], got[-2:], want[-2:]))
                sim.pwave_sim_free(h)
    alphabet = b"abcx \n_."
    for _ in range(500):
        pats = [harness.random_regex(rng) for _ in range(1 if rng.random() < 0.8 else 2)]
        with S.Pool() as pool:
            try:
                re = S.parse(pool, pats)
            except Exception:
                continue
            prog = S.compile(pool, re)
            h = sim.pwave_sim_build(prog.h)
            if not h:
                continue
            nov = 2 * (re.ncaps + 1)
            for _ in range(3):
                d = bytes(rng.choice(alphabet) for _ in range(rng.choice([1, 7, 40, 130])))
                sizes = [rng.choice([0, 1, 2, 5, 9, 20]) for _ in range(rng.randrange(1, 10))]
                want = _feed_oracle(ora.pike(prog, re.ncaps), d, sizes, nov)
                got = _feed_sim(sim, h, d, sizes, nov)
                n += 1
                if got != want:
                    bad.append((pats, d, sizes, got[-2:], want[-2:]))
            sim.pwave_sim_free(h)
    assert not bad, (len(bad), bad[:3])
    assert n > 1500, n


def _runs(rng, alphabet, total):
    """a subject made of runs: the lists of most programs loop in place over them"""
    out = bytearray()
    while len(out) < total:
        out += bytes([rng.choice(alphabet)]) * rng.choice([1, 1, 2, 3, 9, 30, 70, 150])
    return bytes(out[:total])


def test_wave_stable_runs_vs_oracle(sim):
    """Stable runs (sre_hip_pwave.hip): a byte whose step left the list, every capture column and
    seen_start_state as they were joins a per-list byte set, and the following bytes of that set are
    skipped 64 at a time.  The model with and without the fast-forward, whole buffers, the find-all
    iteration and chunked feeding, against the oracle over subjects made of runs; the headline
    program over a gen-data stream must skip nearly everything."""
    ora = harness.OracleEngine()
    rng = random.Random(int(os.environ.get("SRE_FUZZ_SEED", "20261004")) + 24)
    alphabet = b"abcx \n_."
    zoo = [[rb"[a-z]+@[a-z]+\.[a-z]+"], [rb"([a-z]+)@([a-z]+)\.[a-z]+"], [rb"a+b"], [rb"(a+)(b+)?"], [rb"x(.*)y(.*)z"],
           [rb"^a+c"], [rb"(?:a|b)*c"], [rb"a.*b"], [rb"(a*)*x"], [rb"\n+a"], [rb"[^x]+x", rb"a+_"], [rb"(?:aa)+b"],
           [rb"(a|ab)(c|bcd)(d*)"], [rb"a{3,}b"], [rb"(?:.|\n)*x"], [rb"\s+\S"]]
    cases = [(p, _runs(rng, alphabet, rng.choice([40, 300, 900]))) for p in zoo for _ in range(6)]
    for _ in range(2500):
        nre = 1 if rng.random() < 0.8 else 2
        cases.append(([harness.random_regex(rng) for _ in range(nre)], _runs(rng, alphabet, rng.choice([40, 300, 900]))))
    n, bad = 0, []
    skipped0 = sim.pwave_sim_ff_bytes()
    for pats, data in cases:
        with S.Pool() as pool:
            try:
                re = S.parse(pool, pats)
            except Exception:
                continue
            prog = S.compile(pool, re)
            h = sim.pwave_sim_build(prog.h)
            if not h:
                continue
            nov = 2 * (re.ncaps + 1)
            want = _oracle(ora, prog, re.ncaps, data)
            want_count = _count_want(ora, prog, re.ncaps, data)
            sizes = [rng.choice([0, 1, 5, 64, 65, 200]) for _ in range(rng.randrange(1, 8))]
            want_feed = _feed_oracle(ora.pike(prog, re.ncaps), data, sizes, nov)
            for ff in (1, 0):
                sim.pwave_sim_set_ff(ff)
                got = _exec(sim, h, data, re.ncaps)
                rec = (_i64 * (2 + nov))()
                sim.pwave_sim_count(h, bytes(data), len(data), rec, nov)
                got_feed = _feed_sim(sim, h, data, sizes, nov)
                n += 1
                if got != want or list(rec) != want_count or got_feed != want_feed:
                    bad.append((pats, data[:60], ff, got, want, list(rec)[:4], want_count[:4], got_feed[-1:], want_feed[-1:]))
            sim.pwave_sim_set_ff(1)
            sim.pwave_sim_free(h)
    assert not bad, (len(bad), bad[:3])
    assert n > 2500, n
    skipped = sim.pwave_sim_ff_bytes() - skipped0
    assert skipped > 50000, skipped
    # the headline program over gen-data: everything but a few learning steps is skipped
    data = S.gen_data_host(1 << 16, b" x@abc.cc ")
    with S.Pool() as pool:
        re = S.parse(pool, [rb"[a-z]+@[a-z]+\.[a-z]+"])
        prog = S.compile(pool, re)
        h = sim.pwave_sim_build(prog.h)
        before = sim.pwave_sim_ff_bytes()
        got = _exec(sim, h, data, re.ncaps)
        assert got == _oracle(ora, prog, re.ncaps, data), got
        assert sim.pwave_sim_ff_bytes() - before > len(data) - 64, sim.pwave_sim_ff_bytes() - before
        sim.pwave_sim_free(h)
