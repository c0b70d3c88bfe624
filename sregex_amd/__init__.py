"""sregex_amd — Python mirror of the sregex C API over the MI355X-native library.

This is a thin ctypes binding of ``sregex_amd/lib/libsregex.so`` (built by
``sregex_amd/csrc/Makefile`` / ``__graft_entry__.build()``).  Names, argument
meaning and error behaviour follow the reference's public header
(reference src/sregex/sregex.h:82-171):

    pool  = Pool()
    re    = parse(pool, [b"a|ab"])            # sre_regex_parse / sre_regex_parse_multi
    prog  = compile(pool, re)                 # sre_regex_compile
    ctx   = PikeCtx(pool, prog, ncaps)        # sre_vm_pike_create_ctx
    rc    = ctx.exec(b"blab", eof=True)       # sre_vm_pike_exec -> rc, ctx.ovector, ctx.pending

plus the additive device-resident batched API of include/sregex_hip.h
(``Scanner``).  The matcher itself runs on the GPU only: with no HIP device the
exec calls return SRE_ERROR and the scanner constructors raise.  Nothing here
imports or calls the test oracle.
"""
import ctypes
import os
import tempfile

SRE_OK, SRE_ERROR, SRE_AGAIN, SRE_BUSY, SRE_DONE, SRE_DECLINED = 0, -1, -2, -3, -4, -5
SRE_REGEX_CASELESS, SRE_REGEX_NEWLINE = 1, 2
HIP_THOMPSON, HIP_PIKE_FIRST, HIP_PIKE_COUNT = 0, 1, 2
ENGINE_AUTO, ENGINE_VM, ENGINE_SCAN, ENGINE_NFA = 0, 1, 2, 3

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SREGEX_AMD_LIB") or os.path.join(_HERE, "lib", "libsregex.so")

_vp, _sz, _ssz = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_ssize_t
_pssz = ctypes.POINTER(ctypes.c_ssize_t)

# every exported symbol with its signature: (restype, argtypes)
API = {
    # include/sregex/sregex.h
    "sre_create_pool": (_vp, [_sz]),
    "sre_reset_pool": (None, [_vp]),
    "sre_destroy_pool": (None, [_vp]),
    "sre_regex_parse": (_vp, [_vp, ctypes.c_char_p, ctypes.POINTER(_sz), ctypes.c_int, _pssz]),
    "sre_regex_dump": (None, [_vp]),
    "sre_regex_parse_multi": (_vp, [_vp, ctypes.POINTER(ctypes.c_char_p), _ssz, ctypes.POINTER(_sz),
                                    ctypes.POINTER(ctypes.c_int), _pssz, _pssz]),
    "sre_program_dump": (None, [_vp]),
    "sre_regex_compile": (_vp, [_vp, _vp]),
    "sre_vm_pike_create_ctx": (_vp, [_vp, _vp, _pssz, _sz]),
    "sre_vm_pike_exec": (_ssz, [_vp, _vp, _sz, ctypes.c_uint, ctypes.POINTER(_pssz)]),
    "sre_vm_thompson_create_ctx": (_vp, [_vp, _vp]),
    "sre_vm_thompson_exec": (_ssz, [_vp, _vp, _sz, ctypes.c_uint]),
    "sre_vm_thompson_jit_compile": (_ssz, [_vp, _vp, ctypes.POINTER(_vp)]),
    "sre_vm_thompson_jit_create_ctx": (_vp, [_vp, _vp]),
    "sre_vm_thompson_jit_get_handler": (_vp, [_vp]),
    "sre_vm_thompson_jit_free": (_ssz, [_vp]),
    # include/sregex_hip.h
    "sre_hip_device_count": (ctypes.c_int, []),
    "sre_hip_set_device": (ctypes.c_int, [ctypes.c_int]),
    "sre_hip_scanner_create": (_vp, [_vp, _vp, ctypes.c_int, ctypes.c_int]),
    "sre_hip_scanner_engine": (ctypes.c_int, [_vp]),
    "sre_hip_scanner_result_slots": (_sz, [_vp]),
    "sre_hip_scanner_set_segment_bytes": (ctypes.c_int, [_vp, _sz]),
    "sre_hip_scanner_last_fixups": (ctypes.c_int, [_vp]),
    "sre_hip_scanner_last_lineage_passes": (ctypes.c_int, [_vp]),
    "sre_hip_scanner_last_exact_passes": (ctypes.c_int, [_vp]),
    "sre_hip_scanner_last_count_rounds": (ctypes.c_int, [_vp]),
    "sre_hip_compat_route_counts": (None, [ctypes.POINTER(ctypes.c_ulonglong)]),
    "sre_hip_compat_trim": (ctypes.c_int, []),
    "sre_hip_scanner_set_tail_stream": (ctypes.c_int, [_vp, _vp]),
    "sre_hip_scanner_class_bits": (ctypes.c_int, [_vp]),
    "sre_hip_scanner_kernel_name": (ctypes.c_char_p, [_vp]),
    "sre_hip_scanner_last_kernel_ms": (ctypes.c_double, [_vp]),
    "sre_hip_scanner_last_segment_bytes": (_sz, [_vp]),
    "sre_hip_scanner_order_after_scan": (ctypes.c_int, [_vp, _vp]),
    "sre_hip_scan_enqueue": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_sz), _sz, _vp]),
    "sre_hip_scan_results": (ctypes.c_int, [_vp, _pssz]),
    "sre_hip_scan_batch": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_sz), _sz, _pssz, _vp]),
    "sre_hip_alloc": (_vp, [_sz]),
    "sre_hip_free": (None, [_vp]),
    "sre_hip_upload": (ctypes.c_int, [_vp, _vp, _sz]),
    "sre_hip_download": (ctypes.c_int, [_vp, _vp, _sz]),
    "sre_hip_synchronize": (ctypes.c_int, [_vp]),
    "sre_hip_gen_data": (ctypes.c_int, [_vp, _sz, ctypes.c_char_p, _sz, _vp]),
    "sre_hip_read_ceiling": (ctypes.c_int, [_vp, _sz, _vp]),
    "sre_hip_read_pattern": (ctypes.c_int, [_vp, _sz, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, _vp]),
}

_lib = None
_libc = None


def _missing_symbol(name, path):
    def fail(*_a, **_k):
        raise RuntimeError("%s is not exported by %s (SREGEX_AMD_LIB points at an older build)" % (name, path))
    return fail


def load_library(path=None):
    """Load libsregex.so and declare every entry point.  Fails loudly if the
    library has not been built (see __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "%s is missing: build it with `make -C sregex_amd/csrc` "
            "(or __graft_entry__.build()); there is no pure-Python matcher" % p)
    lib = ctypes.CDLL(p)
    for name, (res, args) in API.items():
        if os.environ.get("SREGEX_AMD_LIB") and not hasattr(lib, name):
            # an older build given for an A/B run (tools/exp_knobs.sh) may lack the newest entry points:
            # such a name fails on first USE with a clear message, never with ctypes' default int restype
            setattr(lib, name, _missing_symbol(name, p))
            continue
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _capture_stdout(fn):
    """Run fn() and return what C stdio wrote to fd 1 (the dumps use printf)."""
    global _libc
    if _libc is None:
        _libc = ctypes.CDLL(None)
    _libc.fflush(None)
    with tempfile.TemporaryFile() as tf:
        saved = os.dup(1)
        os.dup2(tf.fileno(), 1)
        try:
            fn()
            _libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        tf.seek(0)
        return tf.read()


class SyntaxError_(Exception):
    """Regex syntax error; str() is the reference CLI's message
    (reference src/sre_cli.c:121, 156)."""

    def __init__(self, offset, regex_id=None):
        self.offset, self.regex_id = offset, regex_id
        if regex_id is None:
            msg = "[error] syntax error at pos %d" % offset
        else:
            msg = "[error] regex %d: syntax error at pos %d" % (regex_id, offset)
        super().__init__(msg)


class Pool:
    def __init__(self, size=1024):
        self.lib = load_library()
        self.p = self.lib.sre_create_pool(size)
        if not self.p:
            raise MemoryError("sre_create_pool")

    def reset(self):
        self.lib.sre_reset_pool(self.p)

    def destroy(self):
        if self.p:
            self.lib.sre_destroy_pool(self.p)
            self.p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.destroy()


class Regex:
    def __init__(self, pool, handle, ncaps, nregexes):
        self.pool, self.h, self.ncaps, self.nregexes = pool, handle, ncaps, nregexes

    def dump(self):
        lib = self.pool.lib
        return _capture_stdout(lambda: lib.sre_regex_dump(self.h)).decode("latin-1")


def parse(pool, regexes, flags=None, multi=None):
    """sre_regex_parse for one regex, sre_regex_parse_multi for several (or when
    multi=True).  `flags` is a list of per-regex flag ints.  Raises SyntaxError_."""
    lib = pool.lib
    regexes = [bytes(r).split(b"\0")[0] for r in regexes]   # C strings
    if multi is None:
        multi = len(regexes) != 1
    ncaps = _sz(0)
    eo = _ssz(-1)
    if not multi:
        h = lib.sre_regex_parse(pool.p, regexes[0], ctypes.byref(ncaps),
                                (flags or [0])[0], ctypes.byref(eo))
        if not h:
            raise SyntaxError_(eo.value)
    else:
        arr = (ctypes.c_char_p * len(regexes))(*regexes)
        fl = (ctypes.c_int * len(regexes))(*flags) if flags else None
        ei = _ssz(-1)
        h = lib.sre_regex_parse_multi(pool.p, arr, len(regexes), ctypes.byref(ncaps), fl,
                                      ctypes.byref(eo), ctypes.byref(ei))
        if not h:
            raise SyntaxError_(eo.value, ei.value)
    return Regex(pool, h, ncaps.value, len(regexes))


class Program:
    def __init__(self, pool, handle, ncaps, nregexes):
        self.pool, self.h, self.ncaps, self.nregexes = pool, handle, ncaps, nregexes

    def dump(self):
        lib = self.pool.lib
        return _capture_stdout(lambda: lib.sre_program_dump(self.h)).decode("latin-1")


def compile(pool, regex):
    h = pool.lib.sre_regex_compile(pool.p, regex.h)
    if not h:
        raise RuntimeError("sre_regex_compile failed")
    return Program(pool, h, regex.ncaps, regex.nregexes)


def _as_buffer(data):
    """bytes -> (pointer or None, length); NULL for empty input like the CLI's
    splitted modes (reference src/sre_cli.c:373-376)."""
    if data is None or len(data) == 0:
        return None, 0, None
    buf = ctypes.create_string_buffer(bytes(data), len(data))
    return ctypes.cast(buf, _vp), len(data), buf


class PikeCtx:
    """sre_vm_pike_create_ctx / sre_vm_pike_exec.  `ovector` is caller-owned and
    sized 2 * (ncaps + 1), as the reference clients do (src/sre_cli.c:204-205)."""

    def __init__(self, pool, prog, ncaps=None):
        self.lib = pool.lib
        n = 2 * ((prog.ncaps if ncaps is None else ncaps) + 1)
        self.nov = n
        self.ovector = (ctypes.c_ssize_t * n)(*([0] * n))
        self.h = self.lib.sre_vm_pike_create_ctx(pool.p, prog.h, self.ovector, n * 8)
        if not self.h:
            raise MemoryError("sre_vm_pike_create_ctx")
        self.pending = None

    def exec(self, data, eof, want_pending=True, base=None, offset=0, length=None):
        """Feed one chunk.  With `base` (a ctypes buffer) the chunk is
        base[offset:offset+length] — lets a caller re-feed from a match end
        without copying (find-all iteration)."""
        if base is not None:
            ptr = ctypes.cast(ctypes.addressof(base) + offset, _vp)
            n = length
            keep = base
        else:
            ptr, n, keep = _as_buffer(data)
        pend = _pssz()
        rc = self.lib.sre_vm_pike_exec(self.h, ptr, n, 1 if eof else 0,
                                       ctypes.byref(pend) if want_pending else None)
        self.pending = (pend[0], pend[1]) if (want_pending and rc == SRE_AGAIN and pend) else None
        del keep
        return rc


class ThompsonCtx:
    """sre_vm_thompson_create_ctx / sre_vm_thompson_exec."""

    def __init__(self, pool, prog):
        self.lib = pool.lib
        self.h = self.lib.sre_vm_thompson_create_ctx(pool.p, prog.h)
        if not self.h:
            raise MemoryError("sre_vm_thompson_create_ctx")

    def exec(self, data, eof):
        ptr, n, keep = _as_buffer(data)
        rc = self.lib.sre_vm_thompson_exec(self.h, ptr, n, 1 if eof else 0)
        del keep
        return rc


class Scanner:
    """sre_hip_scanner_create + scan_* (include/sregex_hip.h): many device-resident
    streams, one compiled program."""

    def __init__(self, pool, prog, mode, engine=ENGINE_AUTO):
        self.lib = pool.lib
        self.h = self.lib.sre_hip_scanner_create(pool.p, prog.h, mode, engine)
        if not self.h:
            raise RuntimeError("sre_hip_scanner_create failed (no HIP device, or the requested "
                               "engine does not take this program)")
        self.slots = self.lib.sre_hip_scanner_result_slots(self.h)
        self.engine = self.lib.sre_hip_scanner_engine(self.h)
        self._n = 0

    def set_segment_bytes(self, nbytes):
        if self.lib.sre_hip_scanner_set_segment_bytes(self.h, nbytes) != 0:
            raise ValueError("segment size must be a multiple of 64")

    @property
    def engine_name(self):
        return {ENGINE_VM: "vm", ENGINE_SCAN: "scan", ENGINE_NFA: "nfa"}.get(self.engine, "?")

    @property
    def kernel_name(self):
        """the dominant kernel of a scan, as rocprofv3 names it"""
        return self.lib.sre_hip_scanner_kernel_name(self.h).decode()

    @property
    def last_exact_passes(self):
        return self.lib.sre_hip_scanner_last_exact_passes(self.h)

    @property
    def last_count_rounds(self):
        return self.lib.sre_hip_scanner_last_count_rounds(self.h)

    @property
    def last_lineage_passes(self):
        return self.lib.sre_hip_scanner_last_lineage_passes(self.h)

    @property
    def class_bits(self):
        return self.lib.sre_hip_scanner_class_bits(self.h)

    @property
    def last_kernel_ms(self):
        return self.lib.sre_hip_scanner_last_kernel_ms(self.h)

    @property
    def last_segment_bytes(self):
        return self.lib.sre_hip_scanner_last_segment_bytes(self.h)

    @property
    def last_fixups(self):
        return self.lib.sre_hip_scanner_last_fixups(self.h)

    def set_tail_stream(self, hip_stream):
        """queue what follows the scan kernel of every later call on hip_stream (see sregex_hip.h)"""
        if self.lib.sre_hip_scanner_set_tail_stream(self.h, hip_stream) != 0:
            raise RuntimeError("sre_hip_scanner_set_tail_stream failed")

    def order_after_scan(self, hip_stream):
        """make hip_stream wait for this scanner's last scan kernel (see sregex_hip.h)"""
        if self.lib.sre_hip_scanner_order_after_scan(self.h, hip_stream) != 0:
            raise RuntimeError("sre_hip_scanner_order_after_scan failed")

    def enqueue(self, d_ptrs, lens, hip_stream=None):
        n = len(d_ptrs)
        a = (_vp * n)(*d_ptrs)
        b = (_sz * n)(*lens)
        self._n = n
        if self.lib.sre_hip_scan_enqueue(self.h, a, b, n, hip_stream) != 0:
            raise RuntimeError("sre_hip_scan_enqueue failed")

    def results(self):
        out = (ctypes.c_ssize_t * (self._n * self.slots))()
        if self.lib.sre_hip_scan_results(self.h, out) != 0:
            raise RuntimeError("sre_hip_scan_results failed")
        s = self.slots
        return [list(out[i * s:(i + 1) * s]) for i in range(self._n)]

    def scan(self, d_ptrs, lens, hip_stream=None):
        self.enqueue(d_ptrs, lens, hip_stream)
        return self.results()


class DeviceBuffer:
    """A device allocation made through the library (no torch needed)."""

    def __init__(self, nbytes, lib=None):
        self.lib = lib or load_library()
        self.nbytes = nbytes
        self.ptr = self.lib.sre_hip_alloc(nbytes)
        if not self.ptr:
            raise RuntimeError("sre_hip_alloc(%d) failed" % nbytes)

    @classmethod
    def from_bytes(cls, data, lib=None):
        b = cls(max(len(data), 1), lib)
        if len(data) and b.lib.sre_hip_upload(b.ptr, bytes(data), len(data)) != 0:
            raise RuntimeError("sre_hip_upload failed")
        b.nbytes = len(data)
        return b

    def to_bytes(self, n=None):
        n = self.nbytes if n is None else n
        out = ctypes.create_string_buffer(n)
        if n and self.lib.sre_hip_download(out, self.ptr, n) != 0:
            raise RuntimeError("sre_hip_download failed")
        return out.raw

    def free(self):
        if self.ptr:
            self.lib.sre_hip_free(self.ptr)
            self.ptr = None


def compat_route_counts():
    """(whole buffer on a scanner, chunk on the table-driven scanner, exact VM) exec calls so far"""
    out = (ctypes.c_ulonglong * 3)()
    load_library().sre_hip_compat_route_counts(out)
    return tuple(out)


def gen_data_length(n, tail_len):
    """Length of stream(n, tail) = "abccc" x floor((n - tail_len) / 5) + tail
    (reference bench/gen-data.pl:9)."""
    return ((n - tail_len) // 5) * 5 + tail_len


def gen_data_host(n, tail):
    """The same stream on the host (tests, CPU baseline)."""
    return b"abccc" * ((n - len(tail)) // 5) + bytes(tail)
