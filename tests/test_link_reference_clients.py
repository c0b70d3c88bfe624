"""The reference's UNCHANGED clients compile and link against this library
(build container only: needs the reference's sources to compile them; nothing
is copied).  Without a GPU the executors answer SRE_ERROR, so only the parts of
the CLI's output that come from the front end are compared here; the engine
lines are compared on the GPU through the same C ABI by test_gpu_parity.py."""
import os
import subprocess

import pytest

import sregex_amd as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
# binaries built from reference sources live under oracle/_ref/ (git-ignored), like the oracle's
OUT = os.path.join(ROOT, "oracle", "_ref", "clients")

pytestmark = pytest.mark.skipif(not os.path.exists(REF + "/src/sre_cli.c"),
                                reason="reference sources not present on this machine")


def _cc(src, exe, extra=()):
    # oracle/Makefile `clients`: gcc on the reference's file where it lies, -lsregex = the product library
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "clients"])
    return os.path.join(OUT, exe)


def test_sre_cli_links_and_prints_the_reference_dumps(lib):
    exe = _cc(REF + "/src/sre_cli.c", "sregex-cli")
    p = subprocess.run([exe, "a|ab", "blab"], capture_output=True, text=True)
    lines = p.stdout.splitlines()
    assert lines[0] == "Cat(NgStar(Dot), TOPLEVEL(0, Paren(0, Alt(Lit(97), Cat(Lit(97), Lit(98))))))"
    assert lines[1] == "captures: 0"
    assert lines[2:13] == [" 0. split 3, 1", " 1. any", " 2. jmp 0", " 3. save 0", " 4. split 5, 7",
                           " 5. char 97", " 6. jmp 9", " 7. char 97", " 8. char 98", " 9. save 1",
                           "10. match 0"]
    assert "jitted thompson disabled" in lines
    if lib.sre_hip_device_count() == 0:
        assert "pike error" in lines        # loud failure, no CPU fallback
    else:
        assert "pike match 0 (2, 3)" in lines
    p = subprocess.run([exe, "(ab"], capture_output=True, text=True)
    assert "[error] syntax error at pos 3" in p.stderr


def test_bench_sregex_links(lib):
    exe = _cc(REF + "/bench/sregex.c", "sregex-bench", ["-lrt"])
    path = os.path.join(OUT, "abc.txt")
    with open(path, "wb") as f:
        f.write(S.gen_data_host(4096, b"aaabbccb"))
    p = subprocess.run([exe, "--thompson", "--pike", "a?a?a?aaa", path], capture_output=True, text=True)
    assert p.returncode == 0 or lib.sre_hip_device_count() == 0
    assert "sregex" in p.stdout or "error" in (p.stdout + p.stderr).lower()
