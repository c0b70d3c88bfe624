"""Small-input behaviour of the unchanged C API: one
sre_vm_pike_exec / sre_vm_thompson_exec call on a host buffer of 64 B .. 1 MiB,
this library against the real reference (oracle/_ref/libsregex_ref.so) on the same
box.  Writes the measured table to gpurun_out/crossover.json (copied to profiles/)
and checks that at no size the library is slower than the reference's single core
by more than a launch + sync latency budget."""
import ctypes
import json
import os
import sys
import time

import pytest

import sregex_amd as S
import harness

pytestmark = pytest.mark.gpu

SIZES = [64, 1024, 32 * 1024, 1 << 20]
LATENCY_BUDGET_US = 150.0       # one H2D copy + kernel launches + one stream sync of a synchronous exec()


def _ref():
    sys.path.insert(0, os.path.join(harness.HERE, "golden"))
    import make_goldens
    if not os.path.exists(os.path.join(make_goldens.REFDIR, "libsregex_ref.so")):
        pytest.skip("oracle/_ref not built")
    return make_goldens.RefLib()


def _time(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def test_small_input_crossover_vs_reference(lib):
    assert lib.sre_hip_device_count() >= 1
    ref = _ref()
    rows = []
    cases = [("cfg2 no match", [rb"[a-z]+@[a-z]+\.[a-z]+"], b"aaabbccb"),
             ("cfg2 match at the end", [rb"[a-z]+@[a-z]+\.[a-z]+"], b" a@abc.cc "),
             ("leak-free bench pattern", [rb"(?:a|b)aa(?:aa|bb)cc(?:a|b)"], b"aaabbccb")]
    for name, pats, tail in cases:
        rpool, rprog, rncaps = ref.compile(pats)
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            for n in SIZES:
                data = S.gen_data_host(n, tail)
                reps = 200 if n <= 1024 else 40 if n <= 32768 else 5
                want = ref.pike_first(rprog, rncaps, data)

                def ours():
                    with S.Pool() as ep:
                        p = S.PikeCtx(ep, prog, re.ncaps)
                        rc = p.exec(data, True, want_pending=False)
                        return rc, list(p.ovector)

                rc, ov = ours()
                assert rc == want[0] and (rc < 0 or ov == want[1]), (name, n, rc, ov, want)
                t_ours = _time(ours, reps)
                t_ref = _time(lambda: ref.pike_first(rprog, rncaps, data), max(2, reps // 4))
                def ours_th():
                    with S.Pool() as ep:
                        return S.ThompsonCtx(ep, prog).exec(data, True)

                assert ours_th() == ref.thompson(rprog, data)
                t_ours_th = _time(ours_th, reps)
                t_ref_th = _time(lambda: ref.thompson(rprog, data), max(2, reps // 4))
                rows.append({"case": name, "bytes": len(data),
                             "pike_us": t_ours * 1e6, "ref_pike_us": t_ref * 1e6,
                             "thompson_us": t_ours_th * 1e6, "ref_thompson_us": t_ref_th * 1e6,
                             "pike_MBps": len(data) / t_ours / 1e6, "ref_pike_MBps": len(data) / t_ref / 1e6})
        ref.L.sre_destroy_pool(rpool)
    out = os.path.join(harness.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "crossover.json"), "w") as f:
        json.dump({"latency_budget_us": LATENCY_BUDGET_US, "rows": rows}, f, indent=1)
    for r in rows:
        print(r)
    slow = [r for r in rows if r["pike_us"] > r["ref_pike_us"] + LATENCY_BUDGET_US
            or r["thompson_us"] > r["ref_thompson_us"] + LATENCY_BUDGET_US]
    assert not slow, slow


def test_exact_vm_per_stream_rate(lib):
    """Row * of the round-2 verdict: the exact VM's rate on ONE stream (a 1 MiB device buffer, one
    wavefront) — the wave-cooperative Pike step and the wave Thompson kernel, for a thin list (the
    headline program, 2-3 threads listed) and a wide one (configs[2]'s 12 regexes).  The records must
    equal the table-driven scanner's; the rates go to gpurun_out/vm_rate.json (profiles/)."""
    n = 1 << 20
    rows = []
    cfg3 = [b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"]
    cases = [("headline program, no match", [rb"[a-z]+@[a-z]+\.[a-z]+"], b"aaabbccb"),
             ("headline program, match at the end", [rb"[a-z]+@[a-z]+\.[a-z]+"], b" a@abc.cc "),
             ("x(.*)y(.*)z, lists of 3-5", [rb"x(.*)y(.*)z"], b"aaabbccb")]
    buf = S.DeviceBuffer(n)
    for name, pats, tail in cases:
        with S.Pool() as pool:
            re = S.parse(pool, pats)
            prog = S.compile(pool, re)
            L = S.gen_data_length(n, len(tail))
            assert lib.sre_hip_gen_data(buf.ptr, L, tail, len(tail), None) == 0
            want = S.Scanner(pool, prog, S.HIP_PIKE_FIRST, S.ENGINE_SCAN).scan([buf.ptr], [L])[0]
            row = {"case": name, "bytes": L}
            for mode, key in ((S.HIP_PIKE_FIRST, "pike"), (S.HIP_THOMPSON, "thompson")):
                sc = S.Scanner(pool, prog, mode, S.ENGINE_VM)
                got = sc.scan([buf.ptr], [L])[0]
                if mode == S.HIP_PIKE_FIRST:
                    assert got == want, (name, got, want)
                else:
                    assert (got[0] >= 0) == (want[0] >= 0), (name, got, want)
                t = _time(lambda: sc.scan([buf.ptr], [L]), 3)
                row[key + "_ms"] = t * 1e3
                row[key + "_MBps"] = L / t / 1e6
            rows.append(row)
            print(row)
    # the other end: a subject whose list changes with nearly every byte (no stable runs to skip) —
    # the rate of the bare wave step
    import random
    rng = random.Random(7)
    noise = bytes(rng.choice(b"ab@. c") for _ in range(1 << 18))
    nbuf = S.DeviceBuffer.from_bytes(noise)
    with S.Pool() as pool:
        re = S.parse(pool, [rb"([a-z]+)@([a-z]+)\.[a-z]+x"])
        prog = S.compile(pool, re)
        want = S.Scanner(pool, prog, S.HIP_PIKE_FIRST, S.ENGINE_SCAN).scan([nbuf.ptr], [len(noise)])[0]
        row = {"case": "([a-z]+)@([a-z]+)\\.[a-z]+x over random 'ab@. c' (the list changes with nearly every byte)", "bytes": len(noise)}
        for mode, key in ((S.HIP_PIKE_FIRST, "pike"), (S.HIP_THOMPSON, "thompson")):
            sc = S.Scanner(pool, prog, mode, S.ENGINE_VM)
            got = sc.scan([nbuf.ptr], [len(noise)])[0]
            assert got == want if mode == S.HIP_PIKE_FIRST else (got[0] >= 0) == (want[0] >= 0), (got, want)
            t = _time(lambda: sc.scan([nbuf.ptr], [len(noise)]), 2)
            row[key + "_ms"] = t * 1e3
            row[key + "_MBps"] = len(noise) / t / 1e6
        rows.append(row)
        print(row)
    nbuf.free()
    out = os.path.join(harness.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "vm_rate.json"), "w") as f:
        json.dump(rows, f, indent=1)
