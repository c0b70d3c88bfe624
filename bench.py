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


def cpu_baseline(sample_bytes=32 << 20):
    """The reference's own Pike path on the host cores of this box, one core
    (the reference is single-threaded; bench/sregex.c times one exec with
    CLOCK_PROCESS_CPUTIME_ID), on a bounded prefix of the same workload."""
    import sregex_amd as S
    data = S.gen_data_host(sample_bytes, b"aaabbccb")
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "sregex-bench")
    out = {"unit": "GB/s", "cores": 1}
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
            txt = subprocess.run([ref_bin, "--thompson", PATTERN.decode(), path], capture_output=True,
                                 text=True, timeout=600).stdout
            ms = float(txt.split(":")[-1].split("ms")[0].strip().split()[-1])
            out["thompson_value"] = len(data) / ms / 1e6
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
            buf = ctypes.create_string_buffer(data, len(data))
            t0 = time.process_time()
            rc = p.exec(None, True, want_pending=False, base=buf, offset=0, length=len(data))
            dt = time.process_time() - t0
            p.close()
        port = len(data) / dt / 1e9
        if "value" not in out:
            out.update(kind="port", value=port,
                       sample="%d MiB prefix of the workload, oracle/ Pike restatement, 1 core"
                              % (sample_bytes >> 20))
        else:
            out["port_value"] = port
        out["sample_rc"] = rc
    except Exception as e:              # noqa: BLE001
        out["port_error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bytes", type=int, default=0,
                    help="bytes per GPU (default: 4 GiB single stream at N=1, 8 GiB = 128 x 64 MiB streams at N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg2m", "cfg3", "cfg4"],
                    help="N=1 workload: cfg2 = BASELINE configs[1] (default, the headline); cfg2m = "
                         "same with a matching tail (captures span the whole stream); cfg3 = "
                         "configs[2] multi-regex find-all count; cfg4 = configs[3] URI, 4 groups; "
                         "cfg1 = configs[0]'s pattern, Thompson")
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
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("SRE_BENCH_BACKEND", "nccl"))    # "nccl" == RCCL
    lib = S.load_library()
    assert lib.sre_hip_set_device(device) == 0
    stream = torch.cuda.current_stream()
    hstream = ctypes.c_void_p(stream.cuda_stream)

    # ---- resident input -------------------------------------------------
    pats, mode = [PATTERN], S.HIP_PIKE_FIRST
    if args.bytes <= 0:
        args.bytes = 4 * GIB if (world == 1 and not args.many_streams) else 8 * GIB
    if world == 1 and not args.many_streams:
        tail = b"aaabbccb"
        name = "configs[1]: /[a-z]+@[a-z]+\\.[a-z]+/ Pike first-match + captures"
        if args.config == "cfg2m":
            tail, name = b"@abc.cc ", name + ", matching tail '@abc.cc '"
        elif args.config == "cfg3":
            pats = [b"a", b"ab", b"c", b"a(bc)", b"e(f)", b"gh", b"A", b"b", b"BLAH", rb"\s+", b"abcd", b"bc"]
            mode, name = S.HIP_PIKE_COUNT, "configs[2]: 12 regexes of t/04-multi.t combined, find-all count"
        elif args.config == "cfg4":
            pats = [rb"([a-z]+)://([^/ ]+)(/[^ ?]*)?(\?[^ ]*)?"]
            tail, name = b" abc://abc.cc/ab/c?a=b ", "configs[3]: URI pattern, 4 capture groups, matching tail"
        elif args.config == "cfg1":
            pats, mode, name = [b"a?a?a?aaa"], S.HIP_THOMPSON, "configs[0] pattern a?a?a?aaa, Thompson"
        n = S.gen_data_length(args.bytes, len(tail))
        lens = [n]
        tails = [tail]
        workload = "%s; 1 stream x %.2f GiB gen-data (abccc.. + %r)" % (name, n / GIB, tail.decode())
    else:
        per = 64 << 20
        nstreams = max(1, args.bytes // per)
        # stream g lives on rank g mod world; even g carries a matching tail
        mine = shard.shard_streams(nstreams * world, rank, world)
        tails = [(b" a@abc.cc " if g % 2 == 0 else b"aaabbccb") for g in mine]
        lens = [S.gen_data_length(per, len(t)) for t in tails]
        workload = ("configs[4] shape: %d streams x 64 MiB per GPU (round-robin over %d GPUs), "
                    "/[a-z]+@[a-z]+\\.[a-z]+/ Pike, RCCL all-reduce of match counts" % (nstreams, world))
    bufs = [torch.empty(max(n, 16), dtype=torch.uint8, device="cuda") for n in lens]
    for b, n, t in zip(bufs, lens, tails):
        assert lib.sre_hip_gen_data(b.data_ptr(), n, t, len(t), hstream) == 0
    ptrs = [b.data_ptr() for b in bufs]
    total = sum(lens)

    pool = S.Pool()
    prog = S.compile(pool, S.parse(pool, pats))
    # two scanners take turns: step i is queued before the results of step i-1
    # are collected (they travel to pinned memory as part of the queued work),
    # so the GPU goes from one pass straight into the next
    scs = [S.Scanner(pool, prog, mode, S.ENGINE_SCAN) for _ in range(2)]
    sc = scs[0]

    def run(nsteps):
        """nsteps passes over the resident batch; returns (last records, kernel ms of every pass)"""
        recs, kms, inflight = None, [], None
        for i in range(nsteps):
            cur = scs[i % 2]
            cur.enqueue(ptrs, lens, hstream)
            if inflight is not None:
                recs = inflight.results()
                kms.append(inflight.last_kernel_ms)
            inflight = cur
        if inflight is not None:
            recs = inflight.results()
            kms.append(inflight.last_kernel_ms)
        return recs, kms

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    args.warmup = max(args.warmup, 2)       # both scanners allocate their buffers outside the timed region
    recs, _ = run(args.warmup)
    # correctness of what is being timed (size-independent property): a stream
    # matches iff its tail holds the '@' form, and then spans the whole stream
    for n, t, r in zip(lens, tails, recs):
        if args.config == "cfg3":
            assert r == [7, n, n - 1, n, -1, -1], (r, n)     # every byte is a match of a / b / c
        elif args.config == "cfg4":
            assert r == [0, 1, n - 22, n - 1, n - 22, n - 19, n - 16, n - 10, n - 10, n - 5, n - 5, n - 1], r
        elif args.config == "cfg1":
            assert r[:2] == [0, 1], r
        elif t == b"@abc.cc ":
            assert r[:4] == [0, 1, 0, n - 1], (r, n)         # the match spans the whole stream
        elif b"@" in t:
            assert r[:4] == [0, 1, n - 9, n - 1], (r, n)      # "a@abc.cc" in front of the last space
        else:
            assert r[0] == S.SRE_DECLINED and r[1] == 0, r

    barrier()
    t0 = time.perf_counter()
    recs, kernel_ms = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0

    matches = sum(1 for r in recs if r[0] >= 0)
    dt = shard.allreduce_max(dt, "cuda")                        # slowest rank
    matches, total_all = shard.allreduce_counts([matches, total], "cuda")   # the path's only exchange step

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_all * args.steps / dt / 1e9
        kms = sum(kernel_ms) / len(kernel_ms)
        achieved = total / (kms * 1e-3) / 1e9           # this rank's kernel: 1 B per input byte
        line = {
            "metric": "GB/s input scanned (whole job), gen-data stream resident in HBM",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "bytes_per_gpu": total, "streams_per_gpu": len(lens),
                       "segment_bytes": sc.last_segment_bytes, "fixup_rounds": sc.last_fixups,
                       "matches": matches, "engine": "scan", "lineage_passes": sc.last_lineage_passes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "sre_k_scan<%d, %d>" % (2 if mode == S.HIP_PIKE_COUNT else 1, sc.class_bits),
                         "kernel_ms": kms,
                         "algorithmic_bytes_per_launch": total},
        }
        prof = os.path.join(ROOT, "profiles", "r01_pmc_hbm.json")
        if world == 1 and args.config == "cfg2" and not args.many_streams and total == 4 * GIB - 3 \
                and os.path.exists(prof):
            # HBM bytes per launch from the separate rocprofv3 --pmc passes of this
            # same command (profiles/README.md): 2 x FETCH_SIZE (gfx950 correction,
            # MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KB -> bytes
            pm = {(r["counter"], "sre_k_scan<1, 2>" in r["kernel"]): r["mean_value_KB"]
                  for r in json.load(open(prof))}
            if ("FETCH_SIZE", True) in pm:
                line["roofline"]["traffic"] = (2 * pm[("FETCH_SIZE", True)] + pm[("WRITE_SIZE", True)]) * 1024
                line["roofline"]["traffic_source"] = "profiles/r01_pmc_hbm.json"
        if world == 1 and not args.many_streams:
            # measured streaming-read ceiling of this box, same buffer
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lib.sre_hip_read_ceiling(ptrs[0], lens[0], hstream)
            ev0.record(stream)
            for _ in range(5):
                lib.sre_hip_read_ceiling(ptrs[0], lens[0], hstream)
            ev1.record(stream)
            torch.cuda.synchronize()
            line["roofline"]["measured_read_ceiling"] = lens[0] * 5 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
            # ... and of the scanner's own staging pattern with no automaton work
            # (one row per lane, whole 128-byte lines, one stage ahead)
            seg = sc.last_segment_bytes
            if seg % 128 == 0 and lens[0] >= seg:
                lib.sre_hip_read_pattern(ptrs[0], lens[0], seg, 128, 16384, hstream)
                ev0.record(stream)
                for _ in range(5):
                    lib.sre_hip_read_pattern(ptrs[0], lens[0], seg, 128, 16384, hstream)
                ev1.record(stream)
                torch.cuda.synchronize()
                line["roofline"]["measured_staging_ceiling"] = \
                    (lens[0] // seg * seg) * 5 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    pool.destroy()


if __name__ == "__main__":
    main()
