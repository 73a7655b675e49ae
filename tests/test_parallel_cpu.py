"""N > 1 path on CPU: world_size-2 gloo processes exercising ovo_amd.parallel (the collectives bench.py uses)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ovo_amd import parallel
    r, lr, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    # per-step exchange: packed sum-reduce of the instance delta tables
    g = torch.Generator().manual_seed(rank)
    delta = torch.randn(64, 16, generator=g)
    cnt = torch.full((64,), float(rank + 1))
    mine = delta.clone()
    parallel.allreduce_sum_([delta, cnt])
    # dense merge: bucketed in-place reduce (tiny bucket -> several collectives), i32 counts
    acc = torch.full((1000, 8), float(rank + 1))
    c = torch.full((1000,), rank + 1, dtype=torch.int32)
    calls = parallel.allreduce_dense_(acc, c, bucket_bytes=4096)
    t = parallel.max_over_ranks(10.0 + rank, "cpu")
    # per-step exchange of the touched descriptor rows: fixed-size all-gather, rank-major
    rows = torch.full((4, 3), float(rank))
    rows[:, 0] = torch.tensor([rank, -1.0, rank + 10, -1.0])
    g2 = parallel.allgather(torch.arange(6, dtype=torch.int32).reshape(2, 3) + 100 * rank)      # the round's exchange (any dtype)
    assert g2.shape == (world, 2, 3) and all(torch.equal(g2[r2], torch.arange(6, dtype=torch.int32).reshape(2, 3) + 100 * r2) for r2 in range(world))
    nan_row = torch.tensor([float("nan"), 1.5 + rank])
    g3 = parallel.allgather(nan_row)                               # descriptors of empty masks are NaN: they must survive the exchange
    assert torch.isnan(g3[:, 0]).all() and torch.equal(g3[:, 1], torch.tensor([1.5 + r2 for r2 in range(world)]))
    gathered = parallel.allgather_rows(rows)
    assert gathered.shape == (world * 4, 3)
    for r2 in range(world):
        assert torch.equal(gathered[4 * r2:4 * r2 + 4, 0], torch.tensor([r2, -1.0, r2 + 10, -1.0])) and (gathered[4 * r2:4 * r2 + 4, 1:] == r2).all()
    parallel.barrier()
    out[rank] = (mine, delta, cnt, acc, c, calls, t)
    dist.destroy_process_group()


def test_gloo_world2_reduce():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    total = out[0][0] + out[1][0]
    for r in range(world):
        mine, delta, cnt, acc, c, calls, t = out[r]
        torch.testing.assert_close(delta, total)
        assert torch.equal(cnt, torch.full((64,), 3.0))
        assert torch.equal(acc, torch.full((1000, 8), 3.0)) and torch.equal(c, torch.full((1000,), 3, dtype=torch.int32))
        assert calls == 8 + 1 and t == 11.0


def test_single_process_is_a_noop():
    from ovo_amd import parallel
    x = torch.ones(4)
    parallel.allreduce_sum_([x])
    assert parallel.world_size() == 1 and torch.equal(x, torch.ones(4)) and parallel.max_over_ranks(2.5, "cpu") == 2.5
    assert parallel.allreduce_dense_(torch.ones(4, 2), torch.ones(4, dtype=torch.int32)) == 0


def test_dense_shard_bookkeeping():
    """Block-cyclic point shards of the dense accumulators (pipeline.FramePipeline.local_rows / gather_dense): every point has exactly one
    owner and one local row, local rows of a rank are dense in [0, local_rows(n)), and the merge order restores point order."""
    from ovo_amd.pipeline import FramePipeline
    B = 8
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 9, 16, 17, 63, 64, 65, 200):
            owners = [[] for _ in range(world)]
            for p in range(n):
                blk = p // B
                owners[blk % world].append(((blk // world) * B + p % B, p))
            for r in range(world):
                fake = type("P", (), {"world": world, "rank": r, "SHARD_BLOCK": B})()
                nl = FramePipeline.local_rows(fake, n)
                rows = sorted(lr for lr, _ in owners[r])
                assert rows == list(range(nl)), (world, n, r, nl, rows[:5])
            # merge: [world, per * B] -> block b = (b // world, b % world)
            nb = -(-n // B)
            per = -(-nb // world) if nb else 0
            if per:
                local = torch.full((world, per * B), -1, dtype=torch.int64)
                for r in range(world):
                    for lr, p in owners[r]:
                        local[r, lr] = p
                merged = local.reshape(world, per, B).transpose(0, 1).reshape(per * world * B)[:n]
                assert torch.equal(merged, torch.arange(n))
