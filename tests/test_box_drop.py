"""newton/tests/test_rigid_contact.py:775-845 (test_box_drop, XPBD iterations=2): two boxes dropped on the ground, the upper
one tilted -- no body ever moves faster than free fall from the drop height allows, nothing tunnels through the ground, and
after one second both rest near the origin.  Box-plane goes through the analytic path, box-box through MPR/GJK + manifold.
Oracle on the CPU, HIP on the GPU."""
import numpy as np
import pytest

import newton_amd as nt
from newton_amd import _np_math as nm

BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("backend", BACKENDS)
def test_box_drop(oracle_lib, backend):
    h = 0.5
    b = nt.ModelBuilder()
    b.add_ground_plane()
    b1 = b.add_body(xform=[0.0, 0.0, h * 1.2, 0.0, 0.0, 0.0, 1.0])
    b.add_shape_box(b1, hx=h, hy=h, hz=h)
    b2 = b.add_body(xform=[0.0, 0.0, h * 4.2, *nm.quat_from_axis_angle([1.0, 0.0, 0.0], 0.5)])
    b.add_shape_box(b2, hx=h, hy=h, hz=h)
    model = b.finalize(device="cuda:0" if backend == "hip" else None)
    v_max = np.sqrt(2.0 * 9.81 * h * 3.2)
    substeps, dt, frames = 8, 1.0 / 60.0 / 8, 60
    max_vz = 0.0

    if backend == "oracle":
        from oracle_bridge import Oracle, OracleState

        o = Oracle(model)
        s0, s1, oc = OracleState(model), OracleState(model), o.contacts()
        for _ in range(frames):
            for _ in range(substeps):
                s0.body_f[:] = 0
                o.collide(s0.body_q, oc)
                o.xpbd_step(s0, s1, o.control(), oc, dt, iterations=2)
                s0, s1 = s1, s0
            max_vz = max(max_vz, float(np.abs(s0.body_qd[:, 2]).max()))
        q, qd = s0.body_q, s0.body_qd
    else:
        solver = nt.solvers.SolverXPBD(model, iterations=2)
        pipe = nt.CollisionPipeline(model)
        contacts = pipe.contacts()
        s0, s1 = model.state(), model.state()
        for _ in range(frames):
            for _ in range(substeps):
                s0.clear_forces()
                pipe.collide(s0, contacts)
                solver.step(s0, s1, None, contacts, dt)
                s0, s1 = s1, s0
            max_vz = max(max_vz, float(s0.body_qd[:, 2].abs().max().item()))
        q, qd = s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy()

    assert max_vz < v_max
    for i in range(model.body_count):
        assert abs(q[i, 0]) < 1.0 and abs(q[i, 1]) < 1.0
        assert q[i, 2] > 0.5 * h  # did not fall through the ground
        assert np.linalg.norm(qd[i, :3]) < 1.0  # approximately at rest
