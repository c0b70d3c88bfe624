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
