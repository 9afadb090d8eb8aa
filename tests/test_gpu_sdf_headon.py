"""Two box meshes with texture SDFs collide head-on in zero gravity: the reference's mesh-mesh SDF-vs-SDF pipeline test
(newton/tests/test_collision_pipeline.py:84-252 CollisionSetup, :474-509 test_mesh_mesh_sdf_modes / _sdf_vs_sdf): body A flies at
5 m/s into body B, collide() once per frame, ten SolverXPBD substeps on those contacts, 100 frames; afterwards neither body has
picked up lateral velocity (|v_y|, |v_z| < 0.1) and B has been pushed along +x.  The meshes are convex, so the shapes are added as
convex hulls with SDFs -- the pairs the SDF leg of CollisionPipeline.collide takes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("broad_phase", ["explicit", "nxn", "sap"])
def test_mesh_mesh_sdf_vs_sdf(broad_phase):
    import newton_amd as nt

    b = nt.ModelBuilder(gravity=(0.0, 0.0, 0.0))
    b.rigid_gap = 0.005
    bodies = []
    for x in (-1.0, 1.0):
        mesh = nt.Mesh.create_box(0.5, 0.5, 0.5)
        mesh.build_sdf(max_resolution=64)
        body = b.add_body(xform=[x, 0.0, 0.0, 0, 0, 0, 1])
        b.add_shape_convex_hull(body, mesh=mesh)
        bodies.append(body)
    init_velocity = 5.0
    b.body_qd[0][0] = init_velocity
    b.joint_qd[0] = init_velocity
    model = b.finalize(device="cuda:0")
    assert len(model.env.sdf_pair) == 1 and model.env.np == 0  # the one pair goes through the SDF leg
    pipe = nt.CollisionPipeline(model, broad_phase=broad_phase)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model)
    s0, s1, control = model.state(), model.state(), model.control()
    substeps, dt = 10, 1.0 / 60.0 / 10
    touched = False
    for _ in range(100):
        pipe.collide(s0, contacts)
        touched = touched or int(contacts.rigid_contact_count.item()) > 0
        for _ in range(substeps):
            s0.clear_forces()
            solver.step(s0, s1, control, contacts, dt)
            s0, s1 = s1, s0
    qd = s0.body_qd.cpu().numpy()
    assert touched
    assert np.isfinite(qd).all()
    for body in bodies:  # TestLevel.VELOCITY_YZ on A, VELOCITY_LINEAR's lateral part on B, tolerance 0.1
        assert abs(qd[body, 1]) < 0.1 and abs(qd[body, 2]) < 0.1, qd
    assert qd[1, 0] > 0.03 and qd[0, 0] <= init_velocity, qd  # B moves forward, A has not gained speed
