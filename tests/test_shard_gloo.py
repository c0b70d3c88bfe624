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
    with S.Pool() as pool:
        prog = S.compile(pool, S.parse(pool, [rb"[a-z]+@[a-z]+\.[a-z]+"]))
        for g in mine:
            tail = b"@abc.cc " if g % 2 == 0 else b"aaabbccb"
            data = S.gen_data_host(2048 + 5 * g, tail)
            p = ora.pike(prog, 0)
            rc = p.exec(data, True, want_pending=False)
            p.close()
            matches += rc >= 0
            nbytes += len(data)
    total = shard.allreduce_counts([matches, nbytes, len(mine)])
    tmax = shard.allreduce_max(float(rank + 1))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, total, tmax))


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
    owned = sorted(i for _, mine, _, _ in got for i in mine)
    assert owned == list(range(nstreams))                    # a partition, nothing twice
    assert got[0][1] == [0, 2, 4, 6, 8, 10] and got[1][1] == [1, 3, 5, 7, 9]
    # every even stream matches, every odd one does not; all ranks agree on the sums
    exp_bytes = sum((2048 + 5 * g - 8) // 5 * 5 + 8 for g in range(nstreams))
    for _, _, total, tmax in got:
        assert total == [6, exp_bytes, nstreams]
        assert tmax == 2.0
