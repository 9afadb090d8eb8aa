"""N>1 path on CPU: two gloo ranks shard environments, step nothing, and gather state in world-major order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from newton_amd.sharding import gather_body_state, max_over_ranks, shard_range


def test_shard_range_partitions_exactly():
    for total in (1, 7, 4096, 32768, 10):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total_envs, bodies_per_env, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = shard_range(total_envs, rank, world)
    # the unsharded "truth": body i of env w carries the value w*100 + i
    envs = np.arange(b, e)
    bq = np.zeros(((e - b) * bodies_per_env, 7), dtype=np.float32)
    bq[:, 0] = np.repeat(envs, bodies_per_env) * 100 + np.tile(np.arange(bodies_per_env), e - b)
    bqd = -bq[:, :6].copy()
    gq, gqd = gather_body_state(torch.from_numpy(bq), torch.from_numpy(bqd))
    tmax = max_over_ranks(1.0 + rank)
    if rank == 0:
        q_out.put((gq.numpy(), gqd.numpy(), tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_is_world_major():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total, nb = 5, 3  # uneven split: 3 + 2 envs
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, nb, q)) for r in range(2)]
    for p in procs:
        p.start()
    gq, gqd, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.repeat(np.arange(total), nb) * 100 + np.tile(np.arange(nb), total)
    assert gq.shape == (total * nb, 7) and np.array_equal(gq[:, 0], want.astype(np.float32))
    assert np.array_equal(gqd[:, 0], -want.astype(np.float32))
    assert tmax == 2.0
