#!/usr/bin/env python3
"""Generate the golden vectors from the REAL reference built in oracle/_ref.

Runs only in the build container (needs /root/reference + `make -C oracle ref`).
The GPU box gets the committed outputs only:

  tests/golden/t_blocks.jsonl.gz   one record per reference test block
      file, name, re[] (hex), flags, s (hex)         <- data from t/*.t
      cap / match_id / temp_cap / err / ...          <- explicit expectations in t/*.t
      ref: {rc, err, ast, ncaps, prog, res[6]}       <- reference sregex-cli output
      ref_multi: {...}  same, with the never-matching regex '^章亦春$' prepended
                        (t/SRegex.pm:45-47, TEST_SREGEX_FORCE_MULTI_REGEXES)
  tests/golden/gen_data.jsonl      bench/gen-data.pl-style streams x config patterns:
                                   reference Thompson status + Pike rc/ovector
  tests/golden/findall.jsonl       iterated-exec (find-all) traces from the reference

How the reference is driven mirrors t/SRegex.pm:73-84:
   ./sregex-cli --stdin [--flags F] [-n N] RE...   with stdin "<len>\\n<bytes>"
"""
import ctypes
import gzip
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
CLI = os.path.join(REFDIR, "sregex-cli")
FORCE_RE = "^章亦春$".encode("utf-8")


def run_cli(res, flags, subject, multi):
    args = [CLI, "--stdin"]
    if flags:
        args += ["--flags", flags]
    if multi:
        args += ["-n", str(len(res))]
    # execve() cannot carry NUL bytes: perl's exec truncates there too
    args += [r.split(b"\0")[0] for r in res]
    stdin = str(len(subject)).encode() + b"\n" + subject
    p = subprocess.run(args, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr


def split_stdout(out):
    """AST dump / captures / program dump / six engine lines (src/sre_cli.c:174-202, 313-656)."""
    head, sep, tail = out.partition(b"\n## ")
    lines = head.split(b"\n")
    rec = {"ast": lines[0].decode("latin-1")}
    rec["ncaps"] = int(lines[1].split(b": ")[1]) if len(lines) > 1 else None
    rec["prog"] = "\n".join(l.decode("latin-1") for l in lines[2:])
    if sep:
        res = tail.rstrip(b"\n").split(b"\n")[-6:]
        rec["res"] = [l.decode("latin-1") for l in res]
    else:
        rec["res"] = None
    return rec


def one_variant(res, flags, subject, multi):
    rc, out, err = run_cli(res, flags, subject, multi)
    rec = {"rc": rc, "err": err.decode("latin-1")}
    if rc == 0:
        rec.update(split_stdout(out))
    return rec


def do_block(blk):
    res = [bytes.fromhex(h) for h in blk["re"]]
    subject = bytes.fromhex(blk["s"])
    flags = blk.get("flags", "")
    multi = bool(blk["multi"])
    blk["ref"] = one_variant(res, flags, subject, multi)
    if not multi:
        # forced multi-regex variant: flags shift by one regex (t/SRegex.pm:63-69)
        mflags = (" " + flags) if flags else ""
        r = one_variant([FORCE_RE] + res, mflags, subject, True)
        r.pop("ast", None)     # keep the file small: the program + results are the pin
        blk["ref_multi"] = r
    return blk


# ---------------------------------------------------------------- gen-data goldens

def gen_stream(n, tail):
    """bench/gen-data.pl:9 restated: "abccc" x k . tail, total length n."""
    k = (n - len(tail)) // 5
    return b"abccc" * k + tail


class RefLib:
    """ctypes view of oracle/_ref/libsregex_ref.so (public API of src/sregex/sregex.h)."""

    def __init__(self):
        L = ctypes.CDLL(os.path.join(REFDIR, "libsregex_ref.so"))
        vp, sz, ip = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_ssize_t)
        L.sre_create_pool.restype = vp
        L.sre_create_pool.argtypes = [sz]
        L.sre_destroy_pool.argtypes = [vp]
        L.sre_regex_parse.restype = vp
        L.sre_regex_parse.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ip]
        L.sre_regex_parse_multi.restype = vp
        L.sre_regex_parse_multi.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_ssize_t,
                                            ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int), ip, ip]
        L.sre_regex_compile.restype = vp
        L.sre_regex_compile.argtypes = [vp, vp]
        L.sre_vm_pike_create_ctx.restype = vp
        L.sre_vm_pike_create_ctx.argtypes = [vp, vp, ip, sz]
        L.sre_vm_pike_exec.restype = ctypes.c_ssize_t
        L.sre_vm_pike_exec.argtypes = [vp, ctypes.c_void_p, sz, ctypes.c_uint, ctypes.c_void_p]
        L.sre_vm_thompson_create_ctx.restype = vp
        L.sre_vm_thompson_create_ctx.argtypes = [vp, vp]
        L.sre_vm_thompson_exec.restype = ctypes.c_ssize_t
        L.sre_vm_thompson_exec.argtypes = [vp, ctypes.c_void_p, sz, ctypes.c_uint]
        self.L = L

    def compile(self, regexes, flags=None):
        L = self.L
        pool = L.sre_create_pool(4096)
        ncaps = ctypes.c_size_t(0)
        eo = ctypes.c_ssize_t(-1)
        if len(regexes) == 1:
            re = L.sre_regex_parse(pool, regexes[0], ctypes.byref(ncaps), (flags or [0])[0], ctypes.byref(eo))
        else:
            arr = (ctypes.c_char_p * len(regexes))(*regexes)
            fl = (ctypes.c_int * len(regexes))(*(flags or [0] * len(regexes)))
            ei = ctypes.c_ssize_t(-1)
            re = L.sre_regex_parse_multi(pool, arr, len(regexes), ctypes.byref(ncaps), fl,
                                         ctypes.byref(eo), ctypes.byref(ei))
        assert re, "parse failed"
        prog = L.sre_regex_compile(pool, re)
        assert prog
        return pool, prog, ncaps.value

    def thompson(self, prog, data):
        L = self.L
        pool = L.sre_create_pool(4096)
        ctx = L.sre_vm_thompson_create_ctx(pool, prog)
        buf = ctypes.create_string_buffer(data, len(data))
        rc = L.sre_vm_thompson_exec(ctx, ctypes.addressof(buf), len(data), 1)
        L.sre_destroy_pool(pool)
        return rc

    def pike_first(self, prog, ncaps, data):
        L = self.L
        pool = L.sre_create_pool(4096)
        n = 2 * (ncaps + 1)
        ov = (ctypes.c_ssize_t * n)()
        ctx = L.sre_vm_pike_create_ctx(pool, prog, ov, n * 8)
        buf = ctypes.create_string_buffer(data, len(data))
        rc = L.sre_vm_pike_exec(ctx, ctypes.addressof(buf), len(data), 1, None)
        L.sre_destroy_pool(pool)
        return rc, list(ov)

    def pike_findall(self, prog, ncaps, data, limit=1 << 30):
        """Iterate exec on ONE ctx, re-feeding from the match end (SURVEY.md 8b;
        src/sregex/sre_vm_pike.c:179-196, 624-628)."""
        L = self.L
        pool = L.sre_create_pool(4096)
        n = 2 * (ncaps + 1)
        ov = (ctypes.c_ssize_t * n)()
        ctx = L.sre_vm_pike_create_ctx(pool, prog, ov, n * 8)
        buf = ctypes.create_string_buffer(data, len(data))
        base = ctypes.addressof(buf)
        off = 0
        out = []
        while len(out) < limit:
            rc = L.sre_vm_pike_exec(ctx, base + off, len(data) - off, 1, None)
            if rc < 0:
                out.append([rc])
                break
            out.append([rc] + list(ov))
            off = ov[1]
        L.sre_destroy_pool(pool)
        return out


CFG_PATTERNS = {
    "cfg1": [b"a?a?a?aaa"],
    "cfg2": [rb"[a-z]+@[a-z]+\.[a-z]+"],
    "cfg3": [b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"],
    "cfg4": [rb"([a-z]+)://([^/ ]+)(/[^ ?]*)?(\?[^ ]*)?"],
    "benchmk": [b"(?:a|b)aa(?:aa|bb)cc(?:a|b)"],
    "d": [b"d"],
}
TAILS = {
    "plain": b"aaabbccb",
    "email": b"@abc.cc ",
    "uri": b" abc://abc.cc/ab/c?a=b ",
    "none": b"",
}
# ... up to the reference's own benchmark size (bench/gen-data.pl:9 writes 5 242 888 bytes) and 16 MiB,
# the largest the leaking reference Pike VM (SURVEY.md note L) is asked to do
SIZES = [4096 + 3, 1 << 16, (1 << 20) + 8, 5 * (1 << 20) + 8, 1 << 24]


def gen_data_goldens(path):
    ref = RefLib()
    with open(path, "w") as f:
        for cname, pats in CFG_PATTERNS.items():
            pool, prog, ncaps = ref.compile(pats)
            for tname, tail in TAILS.items():
                for n in SIZES:
                    data = gen_stream(n, tail)
                    rec = {"cfg": cname, "re": [p.hex() for p in pats], "tail": tail.hex(), "n": n,
                           "len": len(data), "ncaps": ncaps,
                           "thompson": ref.thompson(prog, data)}
                    rc, ov = ref.pike_first(prog, ncaps, data)
                    rec["pike_rc"] = rc
                    rec["pike_ov"] = ov if rc >= 0 else None
                    f.write(json.dumps(rec) + "\n")
            ref.L.sre_destroy_pool(pool)


FINDALL_CASES = [
    ([b"ab+"], b"xabbyabzab"),
    ([b"a*"], b"baac"),
    ([b"a*?"], b"baac"),
    ([b""], b"abc"),
    ([rb"\b"], b"ab cd"),
    ([b"^a"], b"aa\naa"),
    ([b"a$"], b"aa\naa"),
    ([rb"\w+"], b"hello, world foo_bar 42"),
    ([rb"(\d+)-(\d+)"], b"tel 555-1234 or 12-34-56"),
    (CFG_PATTERNS["cfg3"], b"abccc" * 40 + b"aaabbccb"),
    (CFG_PATTERNS["cfg3"], b"gh BLAH abcd ef \t\n e f a bc"),
    (CFG_PATTERNS["cfg2"], b"mail bob@example.com, al@b.c; x@y"),
    (CFG_PATTERNS["cfg4"], b"see http://a.b/c?d=e and ftp://host/ and x://y "),
    (CFG_PATTERNS["cfg1"], b"abccc" * 10 + b"aaabbccbaaaaaaa"),
    # the leading-byte skip re-seeding a search that already holds a match
    # (sre_vm_pike.c:256-309: the initial-state test ignores the last thread)
    ([rb"(a+)(b+)?"], b"b a\nca"),
    ([rb"(a+)(b+)?"], b" a\nc"),
    ([rb"(a+)(b+)?"], b"\n \n b a\nca.xx b.bcxaa\nbaa\n"),
    ([rb"ab?"], b"xa  ab a"),
    ([rb"a(b|c)?"], b"a a ab ac a"),
    ([rb"[ab]c?"], b"a.b.bc..a"),
]


def findall_goldens(path):
    ref = RefLib()
    with open(path, "w") as f:
        for pats, data in FINDALL_CASES:
            pool, prog, ncaps = ref.compile(pats)
            out = ref.pike_findall(prog, ncaps, data)
            f.write(json.dumps({"re": [p.hex() for p in pats], "s": data.hex(), "ncaps": ncaps,
                                "matches": out}) + "\n")
            ref.L.sre_destroy_pool(pool)


# ---------------------------------------------------------------- SRE_REGEX_NEWLINE goldens

# flag 2 = SRE_REGEX_NEWLINE ('.' and \C do not match "\n", src/sregex/sre_yyparser.y:293-297, :865-869),
# 3 = with SRE_REGEX_CASELESS.  The reference CLI has no switch for it, so the library is driven directly:
# AST dump, program dump and the first match over a subject with newlines.
NEWLINE_CASES = [
    ([b"."], [2]), ([b"a.b"], [2]), ([rb"\C"], [2]), ([rb"a\Cb"], [2]), ([b".*"], [2]), ([b".+x"], [2]),
    ([b"(.)(.)"], [2]), ([b"[^a]"], [2]), ([b"[^a]."], [2]), ([rb"\N."], [2]), ([rb".\n."], [2]),
    ([rb"\s.\S"], [2]), ([b"a.?b"], [2]), ([b"a.{2,3}b"], [2]), ([b"(?:.|x)y"], [2]), ([b"^.$"], [2]),
    ([b".\\z"], [2]), ([rb"\b.\b"], [2]), ([b"A.b"], [3]), ([b"[a-c].D"], [3]), ([rb"\C+"], [3]),
    ([b"a.", b".b"], [2, 2]), ([b"a.", b".b"], [2, 0]), ([b"a.", b".b"], [0, 2]), ([b"x.y", rb"\C\C", b"."], [2, 3, 0]),
    ([b"."], [0]), ([rb"\C"], [0]), ([b"a.b"], [1]),
]
NEWLINE_SUBJECTS = [b"a\nb a.b AxB\n\nxy", b"\n", b"ab\ncd", b"", b"x\ny a\n\nb"]


def _capture_stdout(fn):
    libc = ctypes.CDLL(None)
    libc.fflush(None)
    import tempfile
    with tempfile.TemporaryFile() as tf:
        saved = os.dup(1)
        os.dup2(tf.fileno(), 1)
        try:
            fn()
            libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        tf.seek(0)
        return tf.read()


def newline_goldens(path):
    ref = RefLib()
    L = ref.L
    L.sre_regex_dump.argtypes = [ctypes.c_void_p]
    L.sre_program_dump.argtypes = [ctypes.c_void_p]
    with open(path, "w") as f:
        for pats, flags in NEWLINE_CASES:
            pool = L.sre_create_pool(4096)
            ncaps = ctypes.c_size_t(0)
            eo = ctypes.c_ssize_t(-1)
            if len(pats) == 1:
                re = L.sre_regex_parse(pool, pats[0], ctypes.byref(ncaps), flags[0], ctypes.byref(eo))
            else:
                arr = (ctypes.c_char_p * len(pats))(*pats)
                fl = (ctypes.c_int * len(pats))(*flags)
                ei = ctypes.c_ssize_t(-1)
                re = L.sre_regex_parse_multi(pool, arr, len(pats), ctypes.byref(ncaps), fl, ctypes.byref(eo), ctypes.byref(ei))
            assert re, (pats, flags)
            ast = _capture_stdout(lambda: L.sre_regex_dump(re)).decode("latin-1")
            prog = L.sre_regex_compile(pool, re)
            assert prog
            pdump = _capture_stdout(lambda: L.sre_program_dump(prog)).decode("latin-1")
            runs = []
            for subj in NEWLINE_SUBJECTS:
                rc, ov = ref.pike_first(prog, ncaps.value, subj)
                runs.append({"s": subj.hex(), "thompson": ref.thompson(prog, subj), "pike_rc": rc,
                             "pike_ov": ov if rc >= 0 else None})
            f.write(json.dumps({"re": [p.hex() for p in pats], "flags": flags, "ncaps": ncaps.value,
                                "ast": ast, "prog": pdump, "runs": runs}) + "\n")
            L.sre_destroy_pool(pool)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--newline-only":
        newline_goldens(os.path.join(HERE, "newline_flag.jsonl"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--gen-data-only":
        gen_data_goldens(os.path.join(HERE, "gen_data.jsonl"))
        return
    blocks_path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/blocks.jsonl"
    blocks = [json.loads(l) for l in open(blocks_path)]
    with ThreadPoolExecutor(8) as ex:
        out = list(ex.map(do_block, blocks))
    with gzip.GzipFile(os.path.join(HERE, "t_blocks.jsonl.gz"), "wb", mtime=0) as f:
        for b in out:
            f.write((json.dumps(b, sort_keys=True) + "\n").encode())
    gen_data_goldens(os.path.join(HERE, "gen_data.jsonl"))
    findall_goldens(os.path.join(HERE, "findall.jsonl"))
    n_ok = sum(1 for b in out if b["ref"]["rc"] == 0)
    print("blocks", len(out), "rc==0", n_ok, "errors", len(out) - n_ok)


if __name__ == "__main__":
    main()
