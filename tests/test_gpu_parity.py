"""GPU parity: the HIP path, called through the C ABI, against the reference's
golden vectors and against the oracle on the same inputs.  Bit-exact: rc,
regex id, every ovector slot, temp/pending captures in streaming mode.
"""
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


@pytest.mark.parametrize("engine", [S.ENGINE_VM, S.ENGINE_AUTO])
def test_batched_scan_gen_data_goldens(gpu, engine):
    """Device-resident batched API on gen-data streams vs reference results."""
    recs = harness.load_jsonl("gen_data.jsonl")
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
            nov = 2 * (re.ncaps + 1)
            n, spans = ora.count(prog, data, nov, 1 << 16)
            buf = S.DeviceBuffer.from_bytes(data)
            rec = S.Scanner(pool, prog, S.HIP_PIKE_COUNT, engine).scan([buf.ptr], [len(data)])[0]
            buf.free()
            assert rec[1] == n, (pats, rec, n)
            if n:
                assert rec[0] == spans[-1][0] and rec[2:] == spans[-1][1:], (pats, rec, spans[-1])
            else:
                assert rec[0] == S.SRE_DECLINED


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


def test_gen_data_kernel_matches_host_generator(gpu):
    for n, tail in [(0, b""), (1, b""), (5, b""), (4098, b"aaabbccb"), (100003, b"@abc.cc "), (77, b"x" * 77)]:
        n = S.gen_data_length(n, len(tail)) if n >= len(tail) else len(tail)
        buf = S.DeviceBuffer(max(n, 1))
        assert gpu.sre_hip_gen_data(buf.ptr, n, tail, len(tail), None) == 0
        assert buf.to_bytes(n) == S.gen_data_host(n, tail)
        buf.free()
