"""The C++ checker (oracle/) against golden vectors produced by EXECUTING the reference's own SolverXPBD source
(tests/golden/make_xpbd_reference_vectors.py: /root/reference/newton/_src/solvers/xpbd/solver_xpbd.py + kernels.py +
solvers/solver.py run on a pure-Python stand-in for the Warp API, thread by thread in ascending tid order).  Teacher-forced:
every step starts from the reference's state.  This pins the checker's solver arithmetic and orchestration -- joint forces,
integration, contact and joint position solves, delta application, restitution -- to the reference code itself; the only
restated layer left is the fp32 operation order inside the Warp builtins (quat_rotate, transform products ...), which the
stand-in and oracle/wp_builtins.h share, and the libm behind sin / cos / acos (numpy float32 vs glibc).
The GPU twin (HIP path vs the same vectors) is tests/test_zx_round2_gpu.py::test_hip_path_against_reference_vectors."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = os.path.join(HERE, "golden", "xpbd_reference_vectors.npz")


def cases():
    """name -> (scene factory, steps, dt, solver kwargs, joint_f factory, root lowering): MUST mirror the generator."""
    from scenes import box_stack_scene, joint_zoo_scene, pendulum_scene, quadruped_scene

    return {
        "quadruped_standing": (lambda: quadruped_scene(2, seed=7), 6, 1e-3, dict(iterations=2),
                               lambda nd: 0.4 * np.sin(np.arange(nd)).astype(np.float32)),
        "quadruped_impact_restitution": (lambda: quadruped_scene(1, seed=3), 4, 1e-3, dict(iterations=2, enable_restitution=True), None),
        "pendulum": (lambda: pendulum_scene(2, seed=2), 8, 2e-3, dict(iterations=3), None),
        "joint_zoo": (lambda: joint_zoo_scene(1, seed=5), 5, 1e-3,
                      dict(iterations=3, joint_linear_compliance=1e-4, joint_angular_compliance=2e-4), None),
        "joint_zoo_free_root": (lambda: joint_zoo_scene(1, seed=6, free_root=True), 4, 2e-3, dict(iterations=2, angular_damping=0.1), None),
        "box_stack_no_weighting": (lambda: box_stack_scene(1, n_boxes=3, seed=1, jitter=2e-3), 4, 1.0 / 240.0,
                                   dict(iterations=4, rigid_contact_con_weighting=False, angular_damping=0.05), None),
    }


def _errors(q, qd, q_ref, qd_ref):
    pos = np.abs(q[:, :3] - q_ref[:, :3]).max()
    rot = np.minimum(np.abs(q[:, 3:] - q_ref[:, 3:]).max(axis=1), np.abs(q[:, 3:] + q_ref[:, 3:]).max(axis=1)).max()
    return pos, rot, np.abs(qd[:, :3] - qd_ref[:, :3]).max(), np.abs(qd[:, 3:] - qd_ref[:, 3:]).max()


@pytest.mark.parametrize("name", list(cases()))
def test_checker_reproduces_the_reference_solver_step_by_step(oracle_lib, name):
    import oracle_bridge as ob

    ref = np.load(VEC)
    make, steps, dt, kw, jf, = cases()[name]
    model = make()
    if jf is not None:
        model.joint_f = jf(len(model.joint_f))
    orc = ob.Oracle(model)
    worst = np.zeros(4)
    for k in range(steps):
        q, qd = ref[f"{name}/body_q{k}"], ref[f"{name}/body_qd{k}"]
        ct = orc.contacts()
        orc.collide(q, ct)
        assert int(ct.count[0]) == int(ref[f"{name}/contacts{k}"][0])
        s_in, s_out = ob.OracleState(model, q, qd), ob.OracleState(model, q, qd)
        orc.xpbd_step(s_in, s_out, orc.control(), ct if ct.count[0] else None, dt, **kw)
        e = _errors(s_out.body_q, s_out.body_qd, ref[f"{name}/body_q{k + 1}"], ref[f"{name}/body_qd{k + 1}"])
        worst = np.maximum(worst, e)
    print(name, "max abs error vs the reference run: pos %.3g rot %.3g lin vel %.3g ang vel %.3g" % tuple(worst))
    # one step from identical inputs.  Measured: bit-identical positions and linear velocities in every case; rotations within
    # 1.5e-8 and angular velocities within 3e-6 where asin / acos enter (numpy float32 vs glibc)
    assert worst[0] <= 1e-7 and worst[1] <= 1e-7 and worst[2] <= 1e-6 and worst[3] <= 1e-5, worst
