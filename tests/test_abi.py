"""The C-ABI shared library loads and exports every symbol that include/*.h
declares (no compute calls: this runs without a GPU)."""
import os
import re

import sregex_amd as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"SRE_API[^;(]*?\b(sre_[a-z0-9_]+)\s*\(", text))


def test_every_declared_symbol_is_exported(lib):
    names = _declared("sregex/sregex.h") | _declared("sregex_hip.h")
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(lib, n), n
    # and the Python mirror binds exactly that surface
    assert names == set(S.API), names ^ set(S.API)


def test_pool_and_front_end_work_without_a_gpu(lib):
    with S.Pool() as pool:
        re = S.parse(pool, [b"a|ab"])
        prog = S.compile(pool, re)
        assert prog.dump().splitlines()[0] == " 0. split 3, 1"
        pool.reset()


def test_jit_entry_points_decline(lib):
    """reference clients treat SRE_DECLINED as "JIT disabled" (src/sre_cli.c:419-424)"""
    import ctypes
    with S.Pool() as pool:
        prog = S.compile(pool, S.parse(pool, [b"a"]))
        code = ctypes.c_void_p()
        assert lib.sre_vm_thompson_jit_compile(pool.p, prog.h, ctypes.byref(code)) == S.SRE_DECLINED
        assert lib.sre_vm_thompson_jit_free(None) == S.SRE_OK
