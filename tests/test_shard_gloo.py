"""N>1 path on CPU: world_size-2 gloo run of the sharding + timed-region protocol bench.py uses (no data collective)."""
import os
import sys
import time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "neural-color-transfer_amd", "python"))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nct.shard import shard_pairs, timed_region
    mine = shard_pairs(7, rank, world)
    done = []

    def step(i):
        time.sleep(0.05 * (rank + 1))          # rank 1 is twice as slow: the MAX over ranks must be reported by both
        done.append(i)

    elapsed = timed_region(step, steps=3, warmup=1, dist=dist)
    q.put((rank, mine, len(done), elapsed))
    dist.destroy_process_group()


def test_two_rank_sharding_and_max_time():
    world, port = 2, 29511 + os.getpid() % 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, n0, e0), (r1, m1, n1, e1) = res
    assert m0 == [0, 2, 4, 6] and m1 == [1, 3, 5]                 # static i mod N, every pair exactly once
    assert sorted(m0 + m1) == list(range(7))
    assert n0 == n1 == 4                                           # warmup + steps calls on every rank
    assert abs(e0 - e1) < 1e-9 and e0 >= 3 * 0.1 - 0.02            # both ranks report the slowest rank's time


def test_single_process_path():
    from nct.shard import shard_pairs, timed_region
    assert shard_pairs(5, 0, 1) == [0, 1, 2, 3, 4]
    calls = []
    e = timed_region(lambda i: calls.append(i), steps=2, warmup=1)
    assert calls == [0, 1, 2] and e >= 0
