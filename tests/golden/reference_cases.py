"""Case table shared by tests/golden/make_xpbd_reference_vectors.py (which runs the REFERENCE solver source on these cases) and
tests/test_reference_vectors.py / the GPU twin (which hold the checker and the HIP path against the recorded vectors)."""
import numpy as np


def _scenes():
    from scenes import box_stack_scene, joint_zoo_scene, pendulum_scene, quadruped_scene

    return box_stack_scene, joint_zoo_scene, pendulum_scene, quadruped_scene


def _free_child_scene(*a, **k):
    from scenes import free_child_scene

    return free_child_scene(*a, **k)


def cases():
    box_stack_scene, joint_zoo_scene, pendulum_scene, quadruped_scene = _scenes()
    sin_f = lambda nd: 0.4 * np.sin(np.arange(nd)).astype(np.float32)  # noqa: E731
    return {
        # ---- SolverXPBD (solver_xpbd.py:329-862)
        "quadruped_standing": dict(scene=lambda: quadruped_scene(2, seed=7), steps=6, dt=1e-3, kw=dict(iterations=2), lower=0.24,
                                   joint_f=sin_f),
        "quadruped_impact_restitution": dict(scene=lambda: quadruped_scene(1, seed=3), steps=4, dt=1e-3,
                                             kw=dict(iterations=2, enable_restitution=True), lower=0.2405, drop_speed=0.8),
        "pendulum": dict(scene=lambda: pendulum_scene(2, seed=2), steps=8, dt=2e-3, kw=dict(iterations=3)),
        "joint_zoo": dict(scene=lambda: joint_zoo_scene(1, seed=5), steps=5, dt=1e-3,
                          kw=dict(iterations=3, joint_linear_compliance=1e-4, joint_angular_compliance=2e-4)),
        "joint_zoo_free_root": dict(scene=lambda: joint_zoo_scene(1, seed=6, free_root=True), steps=4, dt=2e-3,
                                    kw=dict(iterations=2, angular_damping=0.1)),
        "box_stack_no_weighting": dict(scene=lambda: box_stack_scene(1, n_boxes=3, seed=1, jitter=2e-3), steps=4, dt=1.0 / 240.0,
                                       kw=dict(iterations=4, rigid_contact_con_weighting=False, angular_damping=0.05)),
        "box_stack_sunk_restitution": dict(scene=lambda: box_stack_scene(1, n_boxes=3, seed=4, jitter=2e-3), steps=3, dt=1.0 / 240.0,
                                           kw=dict(iterations=2, enable_restitution=True), sink=0.002, drop_speed=0.3),
        # attributes set after construction (solver_xpbd.py:171): velocities from the position change of the step (:767-783)
        "quadruped_velocity_from_delta": dict(scene=lambda: quadruped_scene(2, seed=15), steps=4, dt=1e-3, kw=dict(iterations=2),
                                              attrs=dict(compute_body_velocity_from_position_delta=True), lower=0.2405, joint_f=sin_f),
        "box_stack_velocity_from_delta_restitution": dict(scene=lambda: box_stack_scene(1, n_boxes=3, seed=8, jitter=2e-3), steps=3,
                                                          dt=1.0 / 240.0, kw=dict(iterations=2, enable_restitution=True),
                                                          attrs=dict(compute_body_velocity_from_position_delta=True), sink=0.002,
                                                          drop_speed=0.3),
        # with reporting: Contacts.force (update_contacts, solver_xpbd.py:864-921) and State.body_parent_f (:732-754)
        "quadruped_report": dict(scene=lambda: quadruped_scene(1, seed=13, height_jitter=0.0), steps=3, dt=1e-3, kw=dict(iterations=2), lower=0.2425,
                                 joint_f=sin_f, report=True),
        "box_stack_report": dict(scene=lambda: box_stack_scene(1, n_boxes=3, seed=6, jitter=2e-3), steps=3, dt=1.0 / 240.0,
                                 kw=dict(iterations=3), sink=0.001, report=True),
        # ---- SolverSemiImplicit (solver_semi_implicit.py:123-217)
        "semi/pendulum": dict(scene=lambda: pendulum_scene(2, seed=4), steps=6, dt=5e-4, solver="semi_implicit", kw={}),
        "semi/joint_zoo": dict(scene=lambda: joint_zoo_scene(1, seed=8), steps=4, dt=2e-4, solver="semi_implicit",
                               kw=dict(angular_damping=0.1, joint_attach_ke=2.0e4, joint_attach_kd=50.0)),
        "semi/box_stack": dict(scene=lambda: box_stack_scene(1, n_boxes=3, seed=2, jitter=3e-3), steps=4, dt=5e-4,
                               solver="semi_implicit", kw=dict(friction_smoothing=0.5), sink=0.002, drop_speed=0.05),
        "semi/box_stack_contact_props": dict(scene=lambda: box_stack_scene(1, n_boxes=2, seed=3, jitter=3e-3), steps=3, dt=5e-4,
                                             solver="semi_implicit", kw={}, props=(3.0e4, 40.0, 0.5), sink=0.003, drop_speed=0.02),
        "semi/quadruped": dict(scene=lambda: quadruped_scene(1, seed=9), steps=3, dt=2e-4, solver="semi_implicit", kw={}, lower=0.241),
        # ---- SolverFeatherstone (solver_featherstone.py:462-1066), dense (non-tiled) mass-matrix path
        "fs/pendulum": dict(scene=lambda: pendulum_scene(2, seed=5), steps=5, dt=1e-3, solver="featherstone", kw={}),
        "fs/joint_zoo": dict(scene=lambda: joint_zoo_scene(1, seed=9), steps=4, dt=5e-4, solver="featherstone",
                             kw=dict(angular_damping=0.1)),
        "fs/joint_zoo_free_root": dict(scene=lambda: joint_zoo_scene(1, seed=10, free_root=True), steps=3, dt=5e-4,
                                       solver="featherstone", kw={}),
        "fs/quadruped_interval3": dict(scene=lambda: quadruped_scene(1, seed=12), steps=5, dt=5e-4, solver="featherstone",
                                       kw=dict(update_mass_matrix_interval=3), lower=0.241),
        "fs/joint_zoo_interval2": dict(scene=lambda: joint_zoo_scene(1, seed=14), steps=4, dt=5e-4, solver="featherstone",
                                       kw=dict(update_mass_matrix_interval=2)),
        # FREE / DISTANCE joints below the root (solver_featherstone.py:229-265,1006-1046)
        "fs/free_child": dict(scene=lambda: _free_child_scene(2, seed=16), steps=4, dt=5e-4, solver="featherstone", kw={},
                              joint_f=sin_f),
        "fs/free_child_free_root": dict(scene=lambda: _free_child_scene(1, seed=17, free_root=True), steps=4, dt=5e-4,
                                        solver="featherstone", kw=dict(angular_damping=0.05)),
        "fs/quadruped": dict(scene=lambda: quadruped_scene(1, seed=11), steps=3, dt=5e-4, solver="featherstone",
                             kw=dict(friction_smoothing=0.5), lower=0.241, joint_f=sin_f),
    }


def fk_cases():
    """Models whose (joint_q, joint_qd) go through newton.eval_fk."""
    box_stack_scene, joint_zoo_scene, pendulum_scene, quadruped_scene = _scenes()
    return {"joint_zoo": lambda: joint_zoo_scene(2, seed=21), "joint_zoo_free_root": lambda: joint_zoo_scene(2, seed=22, free_root=True),
            "quadruped": lambda: quadruped_scene(2, seed=23), "pendulum": lambda: pendulum_scene(3, seed=24)}


def prepare(case):
    """Build the case's model and apply its initial-state edits (root lowering, sinking, drop speed, joint forces)."""
    import newton_amd as nt

    model = case["scene"]()
    if case.get("joint_f") is not None:
        model.joint_f = case["joint_f"](len(model.joint_f))
    if case.get("lower"):
        jq = np.array(model.joint_q, copy=True).reshape(model.world_count, -1)
        jq[:, 2] -= case["lower"]
        model.joint_q = jq.reshape(-1)
        model.body_q, model.body_qd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
    if case.get("sink"):  # free bodies: body k sinks by (k + 1) * sink, so that every contact of a stack penetrates
        model.body_q = np.array(model.body_q, np.float32, copy=True).reshape(-1, 7)
        model.body_q[:, 2] -= case["sink"] * (np.arange(len(model.body_q)) + 1)
    if case.get("drop_speed"):
        model.body_qd = np.array(model.body_qd, np.float32, copy=True).reshape(-1, 6)
        model.body_qd[:, 2] = -case["drop_speed"]
    return model
