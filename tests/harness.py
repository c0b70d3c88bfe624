"""Shared test plumbing: golden loaders, the oracle binding, and a driver that
replays the reference CLI's per-subject call sequence (reference
src/sre_cli.c:298-660 `process_string`) against any engine.

`oracle/` is test infrastructure; this module (under tests/) is one of the few
places allowed to load it.
"""
import ctypes
import gzip
import json
import os
import subprocess

import sregex_amd as S

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")
FORCE_RE = "^章亦春$".encode("utf-8")

_vp, _sz, _ssz = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_ssize_t
_pssz = ctypes.POINTER(ctypes.c_ssize_t)


# ------------------------------------------------------------------ goldens

def load_blocks():
    with gzip.open(os.path.join(GOLDEN, "t_blocks.jsonl.gz"), "rt") as f:
        return [json.loads(l) for l in f]


def load_jsonl(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return [json.loads(l) for l in f]


def parse_flags(s, n):
    """reference src/sre_cli.c:680-711: space advances to the next regex, 'i' = caseless"""
    fl = [0] * n
    i = 0
    for ch in s or "":
        if ch == " ":
            i += 1
        elif ch == "i":
            fl[i] |= S.SRE_REGEX_CASELESS
    return fl


def block_variants(blk):
    """(variant name, regex list, flags list, multi?, reference record)"""
    res = [bytes.fromhex(h) for h in blk["re"]]
    flags = blk.get("flags", "")
    out = [("ref", res, parse_flags(flags, len(res)), bool(blk["multi"]), blk["ref"])]
    if "ref_multi" in blk:
        mres = [FORCE_RE] + res
        out.append(("ref_multi", mres, parse_flags((" " + flags) if flags else "", len(mres)),
                    True, blk["ref_multi"]))
    return out


# ------------------------------------------------------------------ oracle

_oracle = None


def oracle_lib():
    """liboracle.so = CPU restatement of the reference VMs (oracle/*.c)."""
    global _oracle
    if _oracle is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "port"], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(so)
        L.sre_oracle_pike_create_ctx.restype = _vp
        L.sre_oracle_pike_create_ctx.argtypes = [_vp, _pssz, _sz]
        L.sre_oracle_pike_exec.restype = _ssz
        L.sre_oracle_pike_exec.argtypes = [_vp, _vp, _sz, ctypes.c_uint, ctypes.POINTER(_pssz)]
        L.sre_oracle_pike_free.argtypes = [_vp]
        L.sre_oracle_thompson_create_ctx.restype = _vp
        L.sre_oracle_thompson_create_ctx.argtypes = [_vp]
        L.sre_oracle_thompson_exec.restype = _ssz
        L.sre_oracle_thompson_exec.argtypes = [_vp, _vp, _sz, ctypes.c_uint]
        L.sre_oracle_thompson_free.argtypes = [_vp]
        L.sre_oracle_pike_count.restype = _ssz
        L.sre_oracle_pike_count.argtypes = [_vp, _vp, _sz, _pssz, _sz, _sz, _pssz]
        _oracle = L
    return _oracle


class _OraclePike:
    def __init__(self, prog, ncaps):
        self.L = oracle_lib()
        n = 2 * (ncaps + 1)
        self.ovector = (ctypes.c_ssize_t * n)(*([0] * n))
        self.h = self.L.sre_oracle_pike_create_ctx(prog.h, self.ovector, n * 8)
        self.pending = None

    def exec(self, data, eof, want_pending=True, base=None, offset=0, length=None):
        if base is not None:
            ptr, n, keep = ctypes.cast(ctypes.addressof(base) + offset, _vp), length, base
        else:
            ptr, n, keep = S._as_buffer(data)
        pend = _pssz()
        rc = self.L.sre_oracle_pike_exec(self.h, ptr, n, 1 if eof else 0,
                                         ctypes.byref(pend) if want_pending else None)
        self.pending = (pend[0], pend[1]) if (want_pending and rc == S.SRE_AGAIN and pend) else None
        del keep
        return rc

    def close(self):
        self.L.sre_oracle_pike_free(self.h)
        self.h = None


class _OracleThompson:
    def __init__(self, prog):
        self.L = oracle_lib()
        self.h = self.L.sre_oracle_thompson_create_ctx(prog.h)

    def exec(self, data, eof):
        ptr, n, keep = S._as_buffer(data)
        rc = self.L.sre_oracle_thompson_exec(self.h, ptr, n, 1 if eof else 0)
        del keep
        return rc

    def close(self):
        self.L.sre_oracle_thompson_free(self.h)
        self.h = None


class OracleEngine:
    name = "oracle"

    def pike(self, prog, ncaps):
        return _OraclePike(prog, ncaps)

    def thompson(self, prog):
        return _OracleThompson(prog)

    def count(self, prog, data, nov, max_spans=0):
        L = oracle_lib()
        spans = (ctypes.c_ssize_t * (max(max_spans, 1) * (nov + 1)))()
        buf = ctypes.create_string_buffer(bytes(data), max(len(data), 1))
        final = ctypes.c_ssize_t(0)
        n = L.sre_oracle_pike_count(prog.h, ctypes.cast(buf, _vp), len(data), spans, nov, max_spans,
                                    ctypes.byref(final))
        got = [list(spans[i * (nov + 1):(i + 1) * (nov + 1)]) for i in range(min(n, max_spans))]
        self.final_rc = final.value
        return n, got


class ProductEngine:
    """The product through its C ABI (HIP kernels behind sre_vm_*_exec)."""
    name = "product"

    def __init__(self):
        self.pool = S.Pool()

    def pike(self, prog, ncaps):
        c = S.PikeCtx(self.pool, prog, ncaps)
        c.close = lambda: None
        return c

    def thompson(self, prog):
        c = S.ThompsonCtx(self.pool, prog)
        c.close = lambda: None
        return c

    def recycle(self):
        """Free per-context device state between subjects (the CLI destroys its
        exec pool per subject, src/sre_cli.c:657)."""
        self.pool.destroy()
        self.pool = S.Pool()


# ------------------------------------------------------------------ CLI replay

_RC_WORD = {S.SRE_AGAIN: "again", S.SRE_DECLINED: "no match", S.SRE_ERROR: "error"}


def _fmt_caps(ov, n):
    return "".join(" (%d, %d)" % (ov[i], ov[i + 1]) for i in range(0, n, 2))


def cli_lines(engine, prog, subject, ncaps):
    """The four engine lines the reference CLI prints for one subject that do
    not involve the (dropped) JIT: thompson, splitted thompson, pike, splitted
    pike — same call sequence and formatting as src/sre_cli.c:313-656."""
    out = []
    nov = 2 * (ncaps + 1)

    # thompson, whole buffer (:327-355)
    t = engine.thompson(prog)
    rc = t.exec(subject, True)
    out.append("thompson " + ("match" if rc == S.SRE_OK else _RC_WORD.get(rc, "rc=%d" % rc)))
    t.close()

    # splitted thompson: alternate an empty chunk and a 1-byte chunk (:361-411)
    t = engine.thompson(prog)
    line = None
    for i in range(len(subject)):
        for chunk in (b"", subject[i:i + 1]):
            rc = t.exec(chunk, False)
            if rc != S.SRE_AGAIN:
                break
        if rc != S.SRE_AGAIN:
            break
    else:
        rc = t.exec(b"", True)
    line = "match" if rc == S.SRE_OK else _RC_WORD.get(rc, "rc=%d" % rc)
    out.append("splitted thompson " + line)
    t.close()

    # pike, whole buffer (:537-566)
    p = engine.pike(prog, ncaps)
    rc = p.exec(subject, True, want_pending=False)
    if rc >= 0:
        out.append("pike match %d%s" % (rc, _fmt_caps(p.ovector, nov)))
    else:
        out.append("pike " + _RC_WORD.get(rc, "unknown (%d)" % rc))
    p.close()

    # splitted pike (:570-656): temp captures "[(a, b)]" and pending "(a, b) "
    # are printed for every 1-byte chunk that answers AGAIN
    p = engine.pike(prog, ncaps)
    s = ""
    done = False
    for i in range(len(subject)):
        rc = p.exec(b"", False, want_pending=False)
        if rc == S.SRE_AGAIN:
            rc = p.exec(subject[i:i + 1], False, want_pending=True)
            if rc == S.SRE_AGAIN:
                s += "[(%d, %d)]" % (p.ovector[0], p.ovector[1])
                s += ("(%d, %d) " % p.pending) if p.pending else " "
                continue
        done = True
        break
    if not done:
        rc = p.exec(b"", True, want_pending=True)
    if rc >= 0:
        s += "match %d%s" % (rc, _fmt_caps(p.ovector, nov))
    else:
        s += _RC_WORD.get(rc, "unknown (%d)" % rc)
    out.append("splitted pike " + s)
    p.close()
    return out


def ref_lines(ref):
    """The same four lines out of a golden record (res[] = the CLI's six lines)."""
    r = ref["res"]
    return [r[0], r[1], r[4], r[5]]


def findall(engine, prog, ncaps, data, limit=1 << 30):
    """Iterate exec on ONE context, re-feeding from each match end."""
    p = engine.pike(prog, ncaps)
    nov = 2 * (ncaps + 1)
    base = ctypes.create_string_buffer(bytes(data), max(len(data), 1))
    off, out = 0, []
    while len(out) < limit:
        rc = p.exec(None, True, want_pending=False, base=base, offset=off, length=len(data) - off)
        if rc < 0:
            out.append([rc])
            break
        out.append([rc] + list(p.ovector[:nov]))
        off = p.ovector[1]
    p.close()
    return out


# ------------------------------------------------------------------ random patterns

def random_regex(rng, depth=0):
    """A random pattern over a small alphabet, from the constructs the reference
    parser accepts: literals, classes, '.', groups, alternation, greedy and lazy
    quantifiers, counted repeats, and every assertion."""
    atoms = [b"a", b"b", b"c", b"x", b".", b"[ab]", b"[^a]", b"\\w", b"\\s", b"\\n", b"[a-c]"]
    if os.environ.get("SRE_FUZZ_WIDE"):
        # more of the lexer: escapes, negated shorthands, upper case (for caseless flags), high bytes
        atoms += [b"A", b"X", b"\\.", b"\\W", b"\\S", b"\\d", b"\\D", b"[^\\s]", b"[\\w.]", b"\\x41",
                  b"\\xff", b"[\\x80-\\xff]", b"_", b"[A-C]"]
    asserts = [b"^", b"$", b"\\b", b"\\B", b"\\A", b"\\z"]
    n = rng.randrange(1, 5 if depth == 0 else 4)
    seq = []
    for _ in range(n):
        r = rng.random()
        if r < 0.12:
            piece = rng.choice(asserts)
        elif r < 0.30 and depth < 2:
            inner = random_regex(rng, depth + 1)
            if rng.random() < 0.5:
                inner += b"|" + random_regex(rng, depth + 1)
            piece = (b"(" if rng.random() < 0.7 else b"(?:") + inner + b")"
        else:
            piece = rng.choice(atoms)
        if piece not in asserts and rng.random() < 0.45:
            piece += rng.choice([b"*", b"+", b"?", b"*?", b"+?", b"??", b"{2}", b"{1,3}", b"{0,2}?"])
        seq.append(piece)
    return b"".join(seq)
