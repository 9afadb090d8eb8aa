"""The multi-GPU path's collectives on the real backend with one rank: torch.distributed 'nccl' (= RCCL on ROCm), WORLD_SIZE = 1,
in a child process -- process-group init on the device, the bench's MAX all-reduce and the optional end-of-rollout all_gather of
body_q / body_qd after a sharded rollout.  (World sizes > 1 run on gloo in tests/test_sharding_gloo.py; a
multi-GPU box is the driver's to schedule.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["NT_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NT_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda:0"))
import newton_amd as nt
from newton_amd.sharding import gather_body_state, max_over_ranks, shard_model
from scenes import quadruped_scene
g = quadruped_scene(64, seed=1)
m = shard_model(g, dist.get_rank(), dist.get_world_size(), device="cuda:0")
s0, s1 = m.state(), m.state()
pipe = nt.CollisionPipeline(m); contacts = pipe.contacts(); solver = nt.solvers.SolverXPBD(m)
out = solver.rollout(s0, s1, m.control(), contacts, 1e-3, 10)
q, qd = out.body_q, out.body_qd
gq, gqd = gather_body_state(q, qd, force_collective=True)
assert torch.equal(gq, q) and torch.equal(gqd, qd)
assert max_over_ranks(1.25, device="cuda:0", force_collective=True) == 1.25
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("NCCL_SINGLE_OK", tuple(gq.shape))
'''


def test_rccl_collectives_with_one_rank():
    import torch

    if getattr(torch.cuda, "_newton_emulated", False):
        pytest.skip("RCCL needs the device (not emulated)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", NT_ROOT=root,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_SINGLE_OK (832, 7)" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
