"""The N > 1 path on CPU: two processes (gloo), streams sharded round-robin,
per-rank match counts combined by ONE all-reduce — the same code bench.py runs
over RCCL.  The per-stream verdicts here come from the oracle (this is a test of
the partitioning and the exchange step, not of the kernels)."""
import os
import socket
import sys

import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nstreams, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import sregex_amd as S
    from sregex_amd import shard
    import harness
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_streams(nstreams, rank, world)
    assert len(mine) == shard.local_stream_count(nstreams, rank, world)
    ora = harness.OracleEngine()
    matches = nbytes = 0
    records, per_regex = [], [0, 0]
    with S.Pool() as pool:
        # two regexes: the counter vector has one entry per regex id
        prog = S.compile(pool, S.parse(pool, [rb"[a-z]+@[a-z]+\.[a-z]+", rb"aaabb"]))
        for g in mine:
            tail = b"@abc.cc " if g % 2 == 0 else b"aaabbccb"
            data = S.gen_data_host(2048 + 5 * g, tail)
            p = ora.pike(prog, 0)
            rc = p.exec(data, True, want_pending=False)
            records.append([rc, 1 if rc >= 0 else 0] + list(p.ovector))
            p.close()
            matches += rc >= 0
            if rc >= 0:
                per_regex[rc] += 1
            nbytes += len(data)
    total = shard.allreduce_counts([matches, nbytes, len(mine)])
    counters = shard.allreduce_counter_vector([matches] + per_regex)
    everything = shard.gather_stream_records(records, nstreams, rank, world)
    tmax = shard.allreduce_max(float(rank + 1))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, total, tmax, counters, everything, records))


def test_round_robin_shards_and_one_allreduce():
    world, nstreams = 2, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nstreams, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort()
    owned = sorted(i for g in got for i in g[1])
    assert owned == list(range(nstreams))                    # a partition, nothing twice
    assert got[0][1] == [0, 2, 4, 6, 8, 10] and got[1][1] == [1, 3, 5, 7, 9]
    # every even stream matches, every odd one does not; all ranks agree on the sums
    exp_bytes = sum((2048 + 5 * g - 8) // 5 * 5 + 8 for g in range(nstreams))
    for g in got:
        total, tmax, counters, everything = g[2], g[3], g[4], g[5]
        assert total == [11, exp_bytes, nstreams]
        assert tmax == 2.0
        # uint64[1 + nregexes]: even streams match regex 0 (the '@' form, which starts at
        # offset 0), odd ones regex 1 ("aaabb" in the tail)
        assert counters == [11, 6, 5]
        # the all-gather hands every rank every stream's {rc, count, ovector}, in global order
        assert len(everything) == nstreams
        for i, rec in enumerate(everything):
            L = (2048 + 5 * i - 8) // 5 * 5 + 8
            assert rec == ([0, 1, 0, L - 1] if i % 2 == 0 else [1, 1, L - 8, L - 3]), (i, rec)
    # ... and they are the owners' own records
    for g in got:
        for i, gi in enumerate(g[1]):
            assert got[0][5][gi] == g[6][i]
