#!/usr/bin/env python3
"""bench.py — GB/s of input scanned per GPU on gen-data.pl-style streams.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json):
  N == 1  configs[1]: ONE 4 GiB stream  "abccc" x k . "aaabbccb"  (bench/gen-data.pl:9
          scaled), pattern /[a-z]+@[a-z]+\\.[a-z]+/, Pike semantics (first match +
          captures), input resident in HBM (generated on device).
  N  > 1  configs[4] shape: 128*N independent 64 MiB streams of the same pattern
          (N = 8: the 1024 streams of configs[4]), stream i on rank i mod N
          (128 streams = 8 GiB per GPU, weak scaling),
          tails alternate matching (" a@abc.cc ") / non-matching ("aaabbccb");
          the only collective is one RCCL all-reduce of the per-rank match counts.
A step = one complete scan of the rank's resident input through the public
batched C ABI (sre_hip_scan_enqueue + sre_hip_scan_results), results included.
roofline.frac is the WHOLE step against the HBM peak (all kernels, copies and gaps
of a step: bytes / wall time per step); roofline.kernel_frac is the dominant scan
kernel alone (hipEvents around its launch).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PATTERN = rb"[a-z]+@[a-z]+\.[a-z]+"
GIB = 1 << 30
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _host_cores():
    """threads the CPU legs may use: this process's affinity, at most 16 (a one-GPU box's share of its host)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline_all_cores(per_stream=64 << 20):
    """the CPU path on ALL host cores of this box, one 64 MiB stream of configs[4] per core
    (the reference is single-threaded per stream; streams are independent): the reference's own
    bench/sregex.c --thompson, one process per core, and the oracle/ Pike restatement, one thread per core."""
    import sregex_amd as S
    from concurrent.futures import ThreadPoolExecutor
    cores = _host_cores()
    data = S.gen_data_host(per_stream, b"aaabbccb")
    out = {"unit": "GB/s", "cores": cores, "cpu_model": cpu_model(),
           "sample": "%d streams x %d MiB (one per core), configs[4] pattern" % (cores, per_stream >> 20)}
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "sregex-bench")
    if os.path.exists(ref_bin):
        path = "/tmp/sre_bench_stream.txt"
        with open(path, "wb") as f:
            f.write(data)
        try:
            t0 = time.perf_counter()
            procs = [subprocess.Popen([ref_bin, "--thompson", PATTERN.decode(), path], stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL) for _ in range(cores)]
            for p in procs:
                p.communicate(timeout=600)
            out["reference_thompson_value"] = cores * len(data) / (time.perf_counter() - t0) / 1e9
            out["kind"] = "reference"
        except Exception as e:          # noqa: BLE001
            out["reference_error"] = repr(e)
        finally:
            os.unlink(path)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import harness
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, [PATTERN]))
            ora = harness.OracleEngine()
            bufs = [ctypes.create_string_buffer(data, len(data)) for _ in range(cores)]

            def one(buf):
                p = ora.pike(prog, 0)
                rc = p.exec(None, True, want_pending=False, base=buf, offset=0, length=len(data))
                p.close()
                return rc

            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:       # ctypes releases the GIL for the call
                list(ex.map(one, bufs))
            out["port_value"] = cores * len(data) / (time.perf_counter() - t0) / 1e9
            out.setdefault("kind", "port")
    except Exception as e:              # noqa: BLE001
        out["port_error"] = repr(e)
    out["value"] = out.get("reference_thompson_value", out.get("port_value"))
    return out


def cpu_baseline(sample_bytes=32 << 20, big_sample=256 << 20):
    """The reference's own Pike path on the host cores of this box, one core
    (the reference is single-threaded; bench/sregex.c times one exec with
    CLOCK_PROCESS_CPUTIME_ID), on a bounded prefix of the same workload: 32 MiB for the reference's
    Pike VM (its capture vectors leak ~58 bytes per input byte, SURVEY.md note L: `value` is its
    page-fault path — compare with thompson_value / port_value), 256 MiB for the
    reference's Thompson VM and the leak-free port."""
    import sregex_amd as S
    data = S.gen_data_host(sample_bytes, b"aaabbccb")
    big = S.gen_data_host(big_sample, b"aaabbccb")
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "sregex-bench")
    out = {"unit": "GB/s", "cores": 1, "cpu_model": cpu_model()}
    if os.path.exists(ref_bin):
        path = "/tmp/sre_bench_sample.txt"
        with open(path, "wb") as f:
            f.write(data)
        try:
            txt = subprocess.run([ref_bin, "--pike", PATTERN.decode(), path], capture_output=True,
                                 text=True, timeout=600).stdout
            ms = float(txt.split(":")[-1].split("ms")[0].strip().split()[-1])
            out.update(kind="reference", value=len(data) / ms / 1e6,
                       sample="%d MiB prefix of the workload, reference bench/sregex.c --pike "
                              "(oracle/_ref/sregex-bench), 1 core" % (sample_bytes >> 20))
            with open(path, "wb") as f:
                f.write(big)
            txt = subprocess.run([ref_bin, "--thompson", PATTERN.decode(), path], capture_output=True,
                                 text=True, timeout=600).stdout
            ms = float(txt.split(":")[-1].split("ms")[0].strip().split()[-1])
            out["thompson_value"] = len(big) / ms / 1e6
            out["thompson_sample"] = "%d MiB prefix, reference bench/sregex.c --thompson, 1 core" % (big_sample >> 20)
        except Exception as e:          # noqa: BLE001 - report, do not hide
            out["reference_error"] = repr(e)
        finally:
            os.unlink(path)
    # the leak-free CPU restatement (oracle/), same sample, for comparison
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import harness
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, [PATTERN]))
            p = harness.OracleEngine().pike(prog, 0)
            buf = ctypes.create_string_buffer(big, len(big))
            t0 = time.process_time()
            rc = p.exec(None, True, want_pending=False, base=buf, offset=0, length=len(big))
            dt = time.process_time() - t0
            p.close()
        port = len(big) / dt / 1e9
        out["port_sample"] = "%d MiB prefix, oracle/ Pike restatement, 1 core" % (big_sample >> 20)
        if "value" not in out:
            out.update(kind="port", value=port,
                       sample="%d MiB prefix of the workload, oracle/ Pike restatement, 1 core"
                              % (sample_bytes >> 20))
        else:
            out["port_value"] = port
        out["port_rc"] = rc        # what the port returned on the sample (-5 = SRE_DECLINED: no '@' in it)
    except Exception as e:              # noqa: BLE001
        out["port_error"] = repr(e)
    return out


CFG3 = [b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"]
URI = rb"([a-z]+)://([^/ ]+)(/[^ ?]*)?(\?[^ ]*)?"
# a program the <= 55-state step automaton declines (256+ ordered lists) with 19 list-able
# threads: runs on the 64-bit-mask NFA tier
NFA_PAT = rb"(?:a|b)*a(?:a|b){7}@"
# ... 60 list-able threads (2^28 ordered lists); the two arms of every (?:a|b) are one thread of the
# shift-and form, which then fits 32 bits
NFA60_PAT, NFA60_TAIL = rb"(?:a|b)*a(?:a|b){27}@", b" " + b"ab" * 15 + b"@ "
# ... 57 list-able threads, 54 after merging: 64-bit masks
NFA57_PAT, NFA57_TAIL = rb"(?:a|b)*a[ab]{20}c[^x]{30}@", b" " + b"a" * 22 + b"c" + b"b" * 30 + b"@ "
# configs[2]'s 37-thread program FORCED onto the NFA tier (the table-driven scanner takes it by itself),
# first match: the stream must not match before its end, so its body is "BLegx" (partial matches of
# BLAH, e(f), gh at every position, never a whole one) and the tail "BLAH"
NFA37_BODY, NFA37_TAIL = b"BLegx", b"BLAH"


def workload_spec(name, S, nbytes, rank=0, world=1):
    """name -> dict(pats, mode, lens, tails, text, check(lens, tails, recs))"""
    from sregex_amd import shard
    pats, mode, engine, body = [PATTERN], S.HIP_PIKE_FIRST, S.ENGINE_AUTO, None
    if name == "many":
        per = 64 << 20
        nstreams = max(1, nbytes // per)
        # stream g lives on rank g mod world; even g carries a matching tail
        mine = shard.shard_streams(nstreams * world, rank, world)
        tails = [(b" a@abc.cc " if g % 2 == 0 else b"aaabbccb") for g in mine]
        lens = [S.gen_data_length(per, len(t)) for t in tails]
        text = ("configs[4] shape: %d streams x 64 MiB per GPU (round-robin over %d GPUs), "
                "/[a-z]+@[a-z]+\\.[a-z]+/ Pike, %s all-reduce of match counts"
                % (nstreams, world, "RCCL" if os.environ.get("SRE_BENCH_BACKEND", "nccl") == "nccl" else os.environ["SRE_BENCH_BACKEND"]))
    else:
        tail = b"aaabbccb"
        text = "configs[1]: /[a-z]+@[a-z]+\\.[a-z]+/ Pike first-match + captures"
        if name == "cfg2m":
            tail, text = b"@abc.cc ", text + ", matching tail '@abc.cc '"
        elif name == "cfg3":
            pats, mode = CFG3, S.HIP_PIKE_COUNT
            text = "configs[2]: 12 regexes of t/04-multi.t combined, find-all count"
        elif name == "cfg4":
            pats, tail = [URI], b" abc://abc.cc/ab/c?a=b "
            text = "configs[3]: URI pattern, 4 capture groups, matching tail"
        elif name == "cfg1":
            pats, mode, text = [b"a?a?a?aaa"], S.HIP_THOMPSON, "configs[0] pattern a?a?a?aaa, Thompson"
        elif name == "nfa":
            pats, tail = [NFA_PAT], b" abaabaabab@ "
            text = "declined by the step automaton: /(?:a|b)*a(?:a|b){7}@/ Pike first-match, NFA tier"
        elif name == "floorla":
            # round 3's first floor: a look-ahead match every 4 bytes (find-all count of \bfoo\b over "foo foo foo ..");
            # such matches are folded into the COUNT table now (sre_scan_host.cpp), the stream stays on the fast path
            pats, mode, body, tail = [rb"\bfoo\b"], S.HIP_PIKE_COUNT, b"foo ", b"foo "
            text = "/\\bfoo\\b/ find-all count over 'foo ' repeated (a look-ahead match every 4 bytes)"
        elif name == "floor":
            # worst case left for the table-driven scanner (tools/floor_probe.py): a match whose optional tail starts
            # and fails — the step out of the match's state neither records a match nor kills the list, the pending
            # match is left behind, and both that step and the list's death take the kernel's exact path: every
            # 3 bytes (find-all count of a(?:bc)? over "ab ab ab ..")
            pats, mode, body, tail = [rb"a(?:bc)?"], S.HIP_PIKE_COUNT, b"ab ", b"ab "
            text = "scanner floor: /a(?:bc)?/ find-all count over 'ab ' repeated (a pending match left behind every 3 bytes)"
        elif name == "words":
            # the everyday find-all: a greedy class over words — every byte of a word extends the pending match
            # (the round-3 floor before growing matches were folded into the COUNT table: 0.014)
            pats, mode, body, tail = [rb"[a-z]+"], S.HIP_PIKE_COUNT, b"foo bar ", b"foo bar "
            text = "/[a-z]+/ find-all count over 'foo bar ' repeated (every byte of a word extends the pending match)"
        elif name == "nfala":
            pats, tail = [NFA_PAT + b"$"], b" abaabaabab@"
            text = "declined by the step automaton, with a look-ahead assertion: /(?:a|b)*a(?:a|b){7}@$/ Pike first-match, NFA tier"
        elif name == "nfa60":
            pats, tail = [NFA60_PAT], NFA60_TAIL
            text = "declined by the step automaton, 60 list-able threads: /(?:a|b)*a(?:a|b){27}@/ Pike first-match, NFA tier"
        elif name == "nfa57w":
            pats, tail = [NFA57_PAT], NFA57_TAIL
            text = "declined by the step automaton, 57 list-able threads (64-bit masks): /(?:a|b)*a[ab]{20}c[^x]{30}@/ Pike first-match, NFA tier"
        elif name == "nfa37":
            pats, tail, engine, body = CFG3, NFA37_TAIL, S.ENGINE_NFA, NFA37_BODY
            text = "configs[2]'s 12 regexes (37 list-able threads) forced onto the NFA tier, Pike first-match, body 'BLegx'"
        elif name == "count_nfa":
            # find-all count of a program the step automaton declines: rounds of first-match searches on
            # the NFA tier (one round trip per match), a match every 64 MiB
            block = S.gen_data_length(64 << 20, 13)
            nblk = max(1, nbytes // block)
            return dict(name=name, pats=[NFA_PAT], mode=S.HIP_PIKE_COUNT, lens=[nblk * block], tails=[b" abaabaabab@ "],
                        text="declined by the step automaton: /(?:a|b)*a(?:a|b){7}@/ find-all count on the NFA tier, "
                             "a match every 64 MiB (%d x %d-byte gen-data blocks ending in ' abaabaabab@ ')" % (nblk, block),
                        block=block, check=(lambda recs, n=nblk * block, k=nblk: _assert_eq(recs[0], [0, k, n - 12, n - 1])))
        elif name in ("dense", "densef", "densela"):
            # a match every MiB: the stream is one 1 MiB gen-data block with a matching tail, repeated
            block = S.gen_data_length(1 << 20, 10)
            nblk = max(1, nbytes // block)
            mode = S.HIP_PIKE_FIRST if name == "densef" else S.HIP_PIKE_COUNT
            if name == "densela":
                pats = [rb"\b[a-z]+@[a-z]+\.[a-z]+\b"]       # look-ahead assertions in a find-all count
            text = ("%s, a match every MiB (%d x %d-byte gen-data blocks ending in ' a@abc.cc '), %s"
                    % ("configs[1] pattern" if name != "densela" else "/\\b[a-z]+@[a-z]+\\.[a-z]+\\b/",
                       nblk, block, "first match" if name == "densef" else "find-all count"))
            return dict(name=name, pats=pats, mode=mode, lens=[nblk * block], tails=[b" a@abc.cc "], text=text,
                        block=block, check=(
                            (lambda recs, n=nblk * block, k=nblk: _assert_eq(recs[0], [0, k, n - 9, n - 1]))
                            if name != "densef" else
                            (lambda recs, b=block: _assert_eq(recs[0], [0, 1, b - 9, b - 1]))))
        n = S.gen_data_length(nbytes, len(tail))
        if body and len(body) != 5:
            n = (nbytes - len(tail)) // len(body) * len(body) + len(tail)
        lens, tails = [n], [tail]
        text = "%s; 1 stream x %.2f GiB gen-data (%s.. + %r)" % (text, n / GIB, (body or b"abccc").decode(), tail.decode())

    def check(recs):
        # correctness of what is being timed (size-independent closed forms, each
        # validated against the oracle at small size in tests/test_gpu_parity.py)
        for n, t, r in zip(lens, tails, recs):
            if name == "cfg3":
                assert r == [7, n, n - 1, n, -1, -1], (r, n)     # every byte is a match of a / b / c
            elif name == "cfg4":
                assert r == [0, 1, n - 22, n - 1, n - 22, n - 19, n - 16, n - 10, n - 10, n - 5, n - 5, n - 1], r
            elif name == "cfg1":
                assert r[:2] == [0, 1], r
            elif name == "nfa":
                assert r[:4] == [0, 1, n - 12, n - 1], (r, n)
            elif name == "nfala":
                assert r[:4] == [0, 1, n - 11, n], (r, n)                    # the match ends at the end of input ($)
            elif name in ("nfa60", "nfa57w"):
                assert r[:4] == [0, 1, n - (len(t) - 1), n - 1], (r, n)      # from behind the tail's first blank to its '@'
            elif name == "nfa37":
                assert r == [8, 1, n - 4, n, -1, -1], (r, n)                 # regex 8 = BLAH
            elif name == "floorla":
                assert r == [0, n // 4, n - 4, n - 1], (r, n)                # one match per "foo ", the last one
            elif name == "floor":
                assert r == [0, n // 3, n - 3, n - 2], (r, n)                # one match "a" per "ab ", the last one
            elif name == "words":
                assert r == [0, n // 4, n - 4, n - 1], (r, n)                # one match per word, the last one "bar"
            elif t == b"@abc.cc ":
                assert r[:4] == [0, 1, 0, n - 1], (r, n)         # the match spans the whole stream
            elif b"@" in t:
                assert r[:4] == [0, 1, n - 9, n - 1], (r, n)      # "a@abc.cc" in front of the last space
            else:
                assert r[0] == S.SRE_DECLINED and r[1] == 0, r
    return dict(name=name, pats=pats, mode=mode, lens=lens, tails=tails, text=text, check=check, engine=engine, body=body)


def _assert_eq(got, want):
    assert got == want, (got, want)


class Resident:
    """device buffers of the rank, re-used by every workload of a run"""

    def __init__(self, torch, lib, hstream):
        self.torch, self.lib, self.hstream, self.bufs = torch, lib, hstream, []
        self._side = None

    def side_stream(self, k=0):
        if self._side is None:
            self._side = {}
        if k not in self._side:
            self._side[k] = self.torch.cuda.Stream()
        return self._side[k]

    def fill(self, lens, tails, block=0, body=None):
        need = [max(n, 16) for n in lens]
        if [b.numel() for b in self.bufs] != need:
            self.bufs = []
            self.torch.cuda.empty_cache()
            self.bufs = [self.torch.empty(n, dtype=self.torch.uint8, device="cuda") for n in need]
        for b, n, t in zip(self.bufs, lens, tails):
            if body:
                # another period than gen-data's "abccc": body x k + tail
                pat = self.torch.tensor(list(body), dtype=self.torch.uint8, device="cuda")
                k = (n - len(t)) // len(body)
                assert k * len(body) + len(t) == n, (n, len(body), len(t))
                b[:k * len(body)].view(-1, len(body)).copy_(pat.expand(k, len(body)))
                b[k * len(body):n].copy_(self.torch.tensor(list(t), dtype=self.torch.uint8, device="cuda"))
            elif block:
                # one generated block, repeated (a device-to-device broadcast copy)
                assert self.lib.sre_hip_gen_data(b.data_ptr(), block, t, len(t), self.hstream) == 0
                self.torch.cuda.synchronize()
                b[block:n].view(-1, block).copy_(b[:block].expand(n // block - 1, block))
            else:
                assert self.lib.sre_hip_gen_data(b.data_ptr(), n, t, len(t), self.hstream) == 0
        return [b.data_ptr() for b in self.bufs]


def measure(spec, S, torch, res, hstream, stream, steps, warmup, barrier):
    """W untimed + K timed passes of one workload over this rank's resident input,
    through the public batched C ABI, results included."""
    ptrs = res.fill(spec["lens"], spec["tails"], spec.get("block", 0), spec.get("body"))
    lens = spec["lens"]
    pool = S.Pool()
    prog = S.compile(pool, S.parse(pool, spec["pats"]))
    # two scanners take turns, each on its own HIP stream: step i is queued before the
    # results of step i-1 are collected (they travel to pinned memory as part of the
    # queued work), and the small kernels behind a scan (chain check, captures, the copy
    # of the records) overlap with the next scan instead of sitting between two of them
    depth = max(2, int(os.environ.get("SRE_BENCH_DEPTH", "2")))       # scanners (steps) in flight
    scs = [S.Scanner(pool, prog, spec["mode"], spec.get("engine", S.ENGINE_AUTO)) for _ in range(depth)]
    sc = scs[0]
    if sc.engine == S.ENGINE_VM and sum(lens) > (64 << 20):
        pool.destroy()
        raise RuntimeError("no throughput engine admits this program: the exact VM runs at MB/s, not benchmarked at this size")
    side = res.side_stream()
    # two scanners take turns, each on its own HIP stream, the scan kernels chained by an event
    # (sre_hip_scanner_order_after_scan): what a step queues behind its scan (chain check,
    # capture walk, the copy of the records) overlaps the next step's scan.
    # SRE_BENCH_STREAMS (experiment knob): "tail" = every scan on ONE stream, the tails on a
    # second (sre_hip_scanner_set_tail_stream) — no gap between two scans, but the next scan
    # then takes the chip before the tail kernels get a slot and the step's results wait for
    # it (same box: configs[1] 0.860 vs 0.846 ms, 128 streams 2.25 vs 1.64); "one" = no overlap.
    # Batches of many streams stay on ONE stream: their capture kernel (one single-lane walk
    # per stream) takes 1.2 ms when it shares the chip with the next scan (50 us alone), which
    # the steady state hides (1.61 vs 1.68 ms per step) but the first steps of every run do
    # not (+3.6 ms per run: profiles/r02_experiments.txt) — a short run would measure worse.
    scheme = os.environ.get("SRE_BENCH_STREAMS", "two" if len(lens) == 1 else "one")
    hs = [hstream] * depth
    if scheme == "tail":
        for x in scs:
            x.set_tail_stream(ctypes.c_void_p(side.cuda_stream))
    elif scheme != "one":
        hs = [hstream] + [ctypes.c_void_p(res.side_stream(k).cuda_stream) for k in range(depth - 1)]

    def run(nsteps):
        # step i: scanner and stream i mod depth; its scan kernel follows step i - 1's (event),
        # the results of step i - depth + 1 are collected once it is queued
        recs, kms, inflight = None, [], []
        trace = [] if os.environ.get("SRE_BENCH_TRACE") else None      # per-step wall times to stderr
        t_run = time.perf_counter()
        for i in range(nsteps):
            cur = scs[i % depth]
            if inflight and hs[0] is not hs[1]:
                inflight[-1].order_after_scan(hs[i % depth])    # scan kernels one after the other, tails overlapped
            cur.enqueue(ptrs, lens, hs[i % depth])
            inflight.append(cur)
            if len(inflight) == depth:
                done = inflight.pop(0)
                recs = done.results()
                kms.append(done.last_kernel_ms)
                if trace is not None:
                    trace.append((time.perf_counter(), done.last_kernel_ms, done.last_fixups))
        for done in inflight:
            recs = done.results()
            kms.append(done.last_kernel_ms)
        if trace:
            print("[bench trace] first results after %.2f ms, all after %.2f ms;" % ((trace[0][0] - t_run) * 1e3, (time.perf_counter() - t_run) * 1e3),
                  file=sys.stderr)
            print("[bench trace] " + " ".join("%.2f/%.2f/%d" % ((b[0] - a[0]) * 1e3, b[1], b[2]) for a, b in zip(trace, trace[1:])),
                  file=sys.stderr)
        return recs, kms

    torch.cuda.synchronize()            # the input was generated on the main stream
    recs, _ = run(max(warmup, depth))   # every scanner allocates its buffers outside the timed region
    spec["check"](recs)
    barrier()
    t0 = time.perf_counter()
    recs, kernel_ms = run(steps)
    barrier()
    dt = time.perf_counter() - t0
    spec["check"](recs)
    total = sum(lens)
    kms = sum(kernel_ms) / len(kernel_ms)
    # whole step: every kernel and copy of a step and whatever idles between them — the
    # wall clock over the K steps (two streams: no single stream sees all of it)
    step_gpu_ms = dt / steps * 1e3
    out = dict(dt=dt, total=total, recs=recs, kernel_ms=kms, step_gpu_ms=step_gpu_ms,
               matches=(sum(r[1] for r in recs if r[0] >= 0) if spec["mode"] == S.HIP_PIKE_COUNT
                        else sum(1 for r in recs if r[0] >= 0)),
               segment_bytes=sc.last_segment_bytes, fixup_rounds=sc.last_fixups,
               lineage_passes=sc.last_lineage_passes, engine=sc.engine_name, kernel=sc.kernel_name)
    pool.destroy()
    return out


def roofline(m):
    """whole-step fraction first (all kernels of the step), the dominant kernel alone beside it"""
    whole = m["total"] / (m["step_gpu_ms"] * 1e-3) / 1e9
    scan = m["total"] / (m["kernel_ms"] * 1e-3) / 1e9 if m["kernel_ms"] > 0 else None
    return {"bound": "hbm", "achieved": whole, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": whole / HBM_PEAK_GBS, "traffic": None,
            "step_ms": m["step_gpu_ms"], "kernel": m["kernel"], "kernel_ms": m["kernel_ms"],
            "kernel_achieved": scan, "kernel_frac": scan / HBM_PEAK_GBS if scan else None,
            "algorithmic_bytes_per_launch": m["total"]}


def git_commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                              text=True, timeout=10).stdout.strip() or None
    except Exception:       # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bytes", type=int, default=0,
                    help="bytes per GPU (default: 4 GiB single stream at N=1, 8 GiB = 128 x 64 MiB streams at N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="N=1: do not measure the other configurations beside the headline (config.variants)")
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg2m", "cfg3", "cfg4", "nfa", "nfala", "nfa37", "nfa60", "nfa57w", "count_nfa", "dense", "densef", "densela", "floor", "floorla", "words"],
                    help="N=1 headline workload: cfg2 = BASELINE configs[1] (default); cfg2m = same with a "
                         "matching tail (captures span the whole stream); cfg3 = configs[2] multi-regex "
                         "find-all count; cfg4 = configs[3] URI, 4 groups; cfg1 = configs[0]'s pattern, "
                         "Thompson; nfa = a program the step automaton declines (NFA tier); dense / densef = a match "
                         "every MiB, find-all count / first match")
    ap.add_argument("--many-streams", action="store_true",
                    help="use the N>1 workload shape (64 MiB streams) even on one GPU")
    args = ap.parse_args()

    import torch
    import sregex_amd as S
    from sregex_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # rehearsal knobs (one-GPU box): SRE_BENCH_DEVICE pins every rank to one
    # device, SRE_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU)
    device = int(os.environ.get("SRE_BENCH_DEVICE", local_rank))
    backend = os.environ.get("SRE_BENCH_BACKEND", "nccl")       # "nccl" == RCCL
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend)
    lib = S.load_library()
    assert lib.sre_hip_set_device(device) == 0
    stream = torch.cuda.current_stream()
    hstream = ctypes.c_void_p(stream.cuda_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    many = world > 1 or args.many_streams
    if args.bytes <= 0:
        args.bytes = 8 * GIB if many else 4 * GIB
    res = Resident(torch, lib, hstream)
    head = workload_spec("many" if many else args.config, S, args.bytes, rank, world)
    m = measure(head, S, torch, res, hstream, stream, args.steps, args.warmup, barrier)

    dt = shard.allreduce_max(m["dt"], "cuda")                        # slowest rank
    matches, total_all = shard.allreduce_counts([m["matches"], m["total"]], "cuda")   # the path's only exchange step
    fixups = shard.allreduce_max(float(m["fixup_rounds"]), "cuda")
    if many:
        # every second stream ends in a matching tail: half of the rank's streams, on every rank, and the
        # speculative entry states of this workload settle without a fix-up round
        want = (len(head["lens"]) * world + 1) // 2          # the even global stream indices
        assert matches == want, ("all-reduced match count", matches, want)
        assert fixups == 0, ("fix-up rounds on some rank", fixups)
    devices = shard.allgather_ints(device, "cuda")                   # which ordinal each rank drives

    if rank == 0:
        line = {
            "metric": "GB/s input scanned (whole job), gen-data stream resident in HBM",
            "value": total_all * args.steps / dt / 1e9, "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 2), "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": head["text"], "bytes_per_gpu": m["total"], "streams_per_gpu": len(head["lens"]),
                       "segment_bytes": m["segment_bytes"], "fixup_rounds": m["fixup_rounds"],
                       "matches": matches, "engine": m["engine"], "lineage_passes": m["lineage_passes"],
                       "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                       "world_size": world, "devices": devices, "commit": git_commit()},
            "roofline": roofline(m),
        }
        if world > 1 and len(set(devices)) != world:
            # ranks sharing a GPU: a plumbing rehearsal, not a scaling record
            line["config"]["rehearsal"] = True
            line["value_comparable"] = False
        prof = os.path.join(ROOT, "profiles", "r03_pmc_hbm.json")
        if world == 1 and head["name"] == "cfg2" and m["total"] == 4 * GIB - 3 and os.path.exists(prof):
            # HBM bytes per launch of the dominant kernel from separate rocprofv3 --pmc passes
            # of this same command (profiles/README.md): 2 x FETCH_SIZE (gfx950 correction,
            # MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KB -> bytes.  A stored
            # measurement (PMC counters cannot be read from inside the run): source named.
            pj = json.load(open(prof))
            pm = {(r["counter"], "sre_k_scan<1, 2" in r["kernel"]): r["mean_value_KB"] for r in pj["counters"]}
            if ("FETCH_SIZE", True) in pm:
                line["roofline"]["traffic"] = (2 * pm[("FETCH_SIZE", True)] + pm[("WRITE_SIZE", True)]) * 1024
                line["roofline"]["traffic_source"] = "profiles/r03_pmc_hbm.json (commit %s, %s)" % (
                    pj.get("commit"), pj.get("date"))
        if world == 1 and not many:
            # measured streaming-read ceiling of this box, same buffer
            ptr0, len0 = res.bufs[0].data_ptr(), head["lens"][0]
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lib.sre_hip_read_ceiling(ptr0, len0, hstream)
            ev0.record(stream)
            for _ in range(5):
                lib.sre_hip_read_ceiling(ptr0, len0, hstream)
            ev1.record(stream)
            torch.cuda.synchronize()
            line["roofline"]["measured_read_ceiling"] = len0 * 5 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
            # ... and of the scanner's own staging pattern with no automaton work
            # (one row per lane, whole 128-byte lines, one stage ahead)
            seg = m["segment_bytes"]
            if seg and seg % 128 == 0 and len0 >= seg:
                lib.sre_hip_read_pattern(ptr0, len0, seg, 128, 16384, hstream)
                ev0.record(stream)
                for _ in range(5):
                    lib.sre_hip_read_pattern(ptr0, len0, seg, 128, 16384, hstream)
                ev1.record(stream)
                torch.cuda.synchronize()
                line["roofline"]["measured_staging_ceiling"] = \
                    (len0 // seg * seg) * 5 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
        if world == 1 and not args.no_variants and head["name"] == "cfg2" and not args.many_streams:
            # the other configurations of BASELINE.json on this GPU, same harness: whole-step
            # time and whole-step fraction of the HBM peak (every kernel of the step), the
            # dominant kernel alone beside it
            variants = {}
            sweep = {"size_1.0GiB": GIB, "size_2.5GiB": 5 * GIB // 2, "size_3.3GiB": 33 * GIB // 10, "size_6.0GiB": 6 * GIB}
            for name in ("cfg2m", "cfg3", "cfg4", "nfa", "nfala", "nfa37", "nfa60", "nfa57w", "count_nfa", "dense", "densef", "densela", "floor", "floorla", "words",
                         "many") + tuple(sweep):
                try:
                    # size_*: the headline workload at other stream lengths (the segment geometry follows the total)
                    vs = workload_spec("cfg2" if name in sweep else name, S,
                                       sweep[name] if name in sweep else 8 * GIB if name == "many" else args.bytes)
                    vm = measure(vs, S, torch, res, hstream, stream, min(args.steps, 6), 2, barrier)
                    r = roofline(vm)
                    variants[name] = {"workload": vs["text"], "ms_per_step": vm["dt"] / min(args.steps, 6) * 1e3,
                                      "GBps": r["achieved"], "frac": r["frac"],
                                      "kernel": vm["kernel"], "kernel_ms": vm["kernel_ms"],
                                      "kernel_frac": r["kernel_frac"], "engine": vm["engine"],
                                      "matches": vm["matches"], "fixup_rounds": vm["fixup_rounds"],
                                      "lineage_passes": vm["lineage_passes"], "segment_bytes": vm["segment_bytes"]}
                    if name == "many" and not args.no_cpu_baseline:
                        variants[name]["cpu_baseline_all_cores"] = cpu_baseline_all_cores()
                except Exception as e:          # noqa: BLE001 - a variant must not take the headline down
                    variants[name] = {"error": repr(e)}
            line["config"]["variants"] = variants
        if world == 1 and not many and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
