"""The reference's UNCHANGED clients — src/sre_cli.c and bench/sregex.c, compiled from the reference's
sources in the build container and linked against THIS library (oracle/Makefile `clients`; the binaries
travel to the GPU box under oracle/_ref/clients/, no reference source does) — RUN on the GPU: the
"CLI and bench/sregex.c link unchanged" sentence of BASELINE.json's north_star, for a run and not only
for the link.  Their stdout is compared with what the reference's own build of the same clients
printed (tests/golden/t_blocks.jsonl.gz, gen_data.jsonl); the JIT lines differ by design
(sre_vm_thompson_jit_compile answers SRE_DECLINED: "jitted thompson disabled")."""
import os
import subprocess

import pytest

import sregex_amd as S
import harness

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "oracle", "_ref", "clients", "sregex-cli")
BENCH = os.path.join(ROOT, "oracle", "_ref", "clients", "sregex-bench")
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "sregex_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(CLI) and os.path.exists(BENCH)),
                                 reason="client binaries not built (needs the reference's sources: make -C oracle clients)")]


@pytest.fixture(scope="module")
def gpu(lib):
    assert lib.sre_hip_device_count() >= 1, "no HIP device: the product has no CPU path"
    return lib


def _run_cli(res, flags, subject, multi):
    # how t/SRegex.pm:73-84 drives the CLI (tests/golden/make_goldens.py run_cli)
    args = [CLI, "--stdin"]
    if flags:
        args += ["--flags", flags]
    if multi:
        args += ["-n", str(len(res))]
    args += [r.split(b"\0")[0] for r in res]
    stdin = str(len(subject)).encode() + b"\n" + subject
    return subprocess.run(args, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=ENV, timeout=120)


def test_sregex_cli_unchanged_on_reference_blocks(gpu, blocks):
    """every 8th block of the reference's t/ suite (250 of 1999): the whole stdout of the unchanged CLI —
    AST dump, capture count, program dump, thompson / splitted thompson / pike / splitted pike lines with
    every capture and temporary capture — equals the reference build's; syntax errors print the same line."""
    n = bad = 0
    errors = []
    for blk in blocks[::8]:
        res = [bytes.fromhex(h) for h in blk["re"]]
        subject = bytes.fromhex(blk["s"])
        ref = blk["ref"]
        p = _run_cli(res, blk.get("flags", ""), subject, bool(blk["multi"]))
        n += 1
        if ref["rc"] != 0:
            if p.returncode == 0 or p.stderr.decode("latin-1") != ref["err"]:
                errors.append((blk["file"], blk["name"], p.returncode, p.stderr[:100], ref["err"][:100]))
            continue
        out = p.stdout.decode("latin-1")
        head, sep, tail = out.partition("\n## ")
        lines = head.split("\n")
        got = {"ast": lines[0], "ncaps": int(lines[1].split(": ")[1]), "prog": "\n".join(lines[2:]),
               "res": tail.rstrip("\n").split("\n")[-6:] if sep else None}
        want_res = list(ref["res"])
        # the JIT is dropped: src/sre_cli.c:441-446 prints these two lines when jit_compile declines
        want_res[2], want_res[3] = "jitted thompson disabled", "splitted jitted thompson disabled"
        if p.returncode != 0 or got["ast"] != ref["ast"] or got["ncaps"] != ref["ncaps"] or got["prog"] != ref["prog"] \
                or got["res"] != want_res:
            errors.append((blk["file"], blk["name"], p.returncode, got["res"], want_res))
    assert n >= 240, n
    assert not errors, (len(errors), errors[:3])


def test_sregex_bench_unchanged_on_configs0(gpu, tmp_path):
    """BASELINE configs[0]: bench/sregex.c, pattern a?a?a?aaa, gen-data 1 MiB, Thompson and Pike — and the
    other config patterns at the same size: the unchanged client's verdict lines equal the reference's
    (tests/golden/gen_data.jsonl)."""
    n = 0
    for rec in harness.load_jsonl("gen_data.jsonl"):
        if rec["n"] != (1 << 20) + 8 or len(rec["re"]) != 1:
            continue                    # bench/sregex.c takes one regex
        pat = bytes.fromhex(rec["re"][0])
        data = S.gen_data_host(rec["n"], bytes.fromhex(rec["tail"]))
        assert len(data) == rec["len"]
        path = tmp_path / "gen.txt"
        path.write_bytes(data)
        p = subprocess.run([BENCH, "--thompson", "--pike", pat.decode("latin-1"), str(path)], capture_output=True,
                           text=True, env=ENV, timeout=300)
        assert p.returncode == 0, (pat, p.stdout, p.stderr)
        lines = [l.split(": ")[0] for l in p.stdout.strip().split("\n")]
        want_t = "sregex Thompson " + {0: "match", -5: "no match", -2: "again", -1: "error"}[rec["thompson"]]
        if rec["pike_rc"] >= 0:
            ov = rec["pike_ov"]
            want_p = "sregex Pike match" + "".join(" (%d, %d)" % (ov[i], ov[i + 1]) for i in range(0, len(ov), 2))
        else:
            want_p = "sregex Pike " + {-5: "no match", -2: "again", -1: "error"}[rec["pike_rc"]]
        assert lines == [want_t, want_p], (pat, rec["tail"], lines, want_t, want_p)
        n += 1
    assert n >= 12, n
