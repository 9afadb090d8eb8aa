"""Shard equivalence: one N-env model sliced with shard_range into per-rank sub-models, each stepped
independently, concatenated in rank order == the unsharded rollout, BITWISE.

CPU: two gloo ranks, each running the emulated gfx950 kernels (tests/emu) on its shard and all_gather-ing the final state;
also a single-process check over 2 and 3 shards and of tile_worlds.  GPU (-m gpu): the same with sequential shards on cuda:0
through the product path, at 4096 envs."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))

import newton_amd as nt  # noqa: E402
from newton_amd.sharding import shard_model, shard_range  # noqa: E402
from newton_amd.worlds import slice_worlds, tile_worlds  # noqa: E402

DT = 1e-3


def _lower(model, dz):
    E = model.world_count
    model.joint_q.reshape(E, -1)[:, 2] -= dz
    bq, bqd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
    model.body_q, model.body_qd = bq, bqd


def _global_model(n):
    from scenes import quadruped_scene

    model = quadruped_scene(n, seed=1)
    _lower(model, 0.25)
    rng = np.random.default_rng(11)
    model.body_qd = (model.body_qd + rng.normal(0, 0.1, size=model.body_qd.shape)).astype(np.float32)
    return model


def _emu_rollout(model, substeps=5):
    import harness as H

    em = H.EmuModel(model)
    out = H.xpbd_rollout(em, H.EmuState(em), H.EmuState(em), H.EmuControl(em), H.EmuContacts(em), DT, substeps)
    return out.aos("body_q"), out.aos("body_qd")


def test_slices_are_valid_models_and_tile_round_trips():
    model = _global_model(7)
    parts = [slice_worlds(model, b, e) for b, e in ((0, 3), (3, 4), (4, 7))]
    assert sum(p.world_count for p in parts) == 7
    for k in ("body_q", "body_qd", "joint_q", "joint_qd", "joint_target_q", "body_mass"):
        assert np.array_equal(np.concatenate([getattr(p, k) for p in parts]), getattr(model, k)), k
    for p in parts:
        for k in ("pair_a", "pair_b", "body_joint_list", "body_pair_list", "joint_parent", "shape_body", "np", "cpp", "ns", "ng"):
            assert np.array_equal(getattr(p.env, k), getattr(model.env, k)), k
    tiled = tile_worlds(parts[0], 2)
    assert tiled.world_count == 6 and np.array_equal(tiled.body_q[:39], parts[0].body_q)
    assert np.array_equal(tiled.shape_contact_pairs[:, 0] % 13, np.tile(np.arange(13), 6) % 13)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_emulated_rollout_is_bitwise_the_unsharded_one(oracle_lib, world):
    import harness

    harness.lib()
    model = _global_model(11)  # uneven split
    q_ref, qd_ref = _emu_rollout(model)
    qs, qds = zip(*[_emu_rollout(shard_model(model, r, world)) for r in range(world)])
    assert np.array_equal(np.concatenate(qs), q_ref)
    assert np.array_equal(np.concatenate(qds), qd_ref)


def _worker(rank, world, port, total_envs, q_out):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from newton_amd.sharding import gather_body_state

    model = _global_model(total_envs)  # every rank builds the same global model (same seeds), then keeps its slice
    shard = shard_model(model, rank, world)
    q, qd = _emu_rollout(shard)
    gq, gqd = gather_body_state(torch.from_numpy(q), torch.from_numpy(qd))
    if rank == 0:
        q_out.put((gq.numpy(), gqd.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_step_their_shards_and_gather_the_unsharded_state(oracle_lib):
    import harness
    import torch.multiprocessing as mp

    harness.lib()  # build the emulated library once, before the ranks start
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total = 9
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    gq, gqd = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    q_ref, qd_ref = _emu_rollout(_global_model(total))
    assert np.array_equal(gq, q_ref) and np.array_equal(gqd, qd_ref)


@pytest.mark.gpu
def test_two_sequential_gpu_shards_equal_the_unsharded_rollout_bitwise():
    """4096 envs on cuda:0 as one model vs two / three shard models stepped one after the other (what two ranks would run)."""
    import torch

    def run(model):
        m = slice_worlds(model, 0, model.world_count, device="cuda:0")
        s0, s1 = m.state(), m.state()
        contacts = nt.CollisionPipeline(m).contacts()
        out = nt.solvers.SolverXPBD(m, iterations=2).rollout(s0, s1, None, contacts, DT, 10)
        torch.cuda.synchronize()
        return out.body_q.cpu().numpy(), out.body_qd.cpu().numpy(), contacts.rigid_contact_count_per_env.cpu().numpy()

    n = int(os.environ.get("NT_FULL_SIZE_C4_ENVS", "4096"))  # (shrunk by the emulated dry run)
    model = _global_model(n)
    q_ref, qd_ref, c_ref = run(model)
    for world in (2, 3):
        assert [shard_range(n, r, world) for r in range(world)][-1][1] == n
        qs, qds, cs = zip(*[run(shard_model(model, r, world)) for r in range(world)])
        assert np.array_equal(np.concatenate(qs), q_ref)
        assert np.array_equal(np.concatenate(qds), qd_ref)
        assert np.array_equal(np.concatenate(cs), c_ref)
