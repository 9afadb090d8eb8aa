"""Environment sharding across the GPUs of one node.

Environments never interact (broad_phase_common.py:264-265: different non-negative worlds never pair), so the
path shards embarrassingly: each rank owns a contiguous range of environments, holds its own Model / State /
Contacts slices and steps them independently -- zero communication per step.  The only collective is the optional
end-of-rollout gather of body_q (28 B) + body_qd (24 B) per body, a single all_gather over RCCL/xGMI
(`backend="nccl"` is RCCL on ROCm; `gloo` on CPU for tests).
"""
from __future__ import annotations

import os


def dist_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total_envs: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [begin, end) env range of `rank`; remainders go to the lowest ranks."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, rem = divmod(total_envs, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_model(model, rank: int, world: int, device=None):
    """Rank `rank`'s shard of a global model: worlds shard_range(world_count, rank, world) plus the shared (world -1) shapes,
    as a self-contained Model (newton_amd.worlds.slice_worlds).  Stepping the shards independently and concatenating their
    states in rank order is bitwise the unsharded result (tests/test_shard_equivalence.py)."""
    from .worlds import slice_worlds  # noqa: PLC0415

    b, e = shard_range(model.world_count, rank, world)
    if e <= b:
        raise ValueError(f"rank {rank} of {world} owns no environment of a {model.world_count}-world model")
    return slice_worlds(model, b, e, device=device)


def max_over_ranks(value: float, device=None, force_collective: bool = False) -> float:
    """MAX all-reduce of a host scalar (the bench's timing contract). No-op without a process group (or with one rank, unless
    `force_collective`: the single-rank RCCL smoke test)."""
    import torch  # noqa: PLC0415
    import torch.distributed as dist  # noqa: PLC0415

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_body_state(body_q, body_qd, force_collective: bool = False):
    """all_gather the per-rank AoS body_q [B_r, 7] / body_qd [B_r, 6] tensors into rank-ordered global arrays
    (= the unsharded model's world-major order, because shards are contiguous env ranges)."""
    import torch  # noqa: PLC0415
    import torch.distributed as dist  # noqa: PLC0415

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return body_q, body_qd
    world = dist.get_world_size()
    counts = torch.zeros(world, dtype=torch.int64, device=body_q.device)
    counts[dist.get_rank()] = body_q.shape[0]
    dist.all_reduce(counts)
    nmax = int(counts.max().item())
    packed = torch.zeros((nmax, 13), dtype=body_q.dtype, device=body_q.device)
    packed[: body_q.shape[0], :7] = body_q
    packed[: body_q.shape[0], 7:] = body_qd
    out = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(out, packed)
    rows = torch.cat([out[r][: int(counts[r].item())] for r in range(world)], dim=0)
    return rows[:, :7].contiguous(), rows[:, 7:].contiguous()
