"""Three hydroelastic cubes stacked on the ground stay put for one second under SolverXPBD(iterations=10) with the contacts of
CollisionPipeline.collide's hydroelastic leg -- the reference's own end-to-end test (newton/tests/test_hydroelastic.py:442-637:
build_stacked_cubes_scene / run_stacked_cubes_hydroelastic_test, registered for XPBD at :2759-2773), primitive 1 m cubes, with the
reduction (anchor contacts, as the reference's scene configures it) and without it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NUM_CUBES, SIM_DT, SIM_TIME, MAX_ROTATION_DEG = 3, 1.0 / 60.0, 1.0, 10.0


@pytest.mark.parametrize("reduce_contacts,threshold_factor,substeps", [(True, 0.20, 10), (False, 0.50, 20)])
def test_stacked_primitive_cubes_hydroelastic_xpbd(reduce_contacts, threshold_factor, substeps):
    import torch

    if not torch.cuda.is_available() or getattr(torch.cuda, "_newton_emulated", False) or torch.version.hip is None:
        pytest.skip("1 200 collide + step calls on ~2 000 faces each: device only")
    import newton_amd as nt

    cube_half = 0.5
    narrow_band = contact_gap = cube_half * 0.2
    b = nt.ModelBuilder()
    cfg = b.default_shape_cfg
    cfg.mu, cfg.gap = 0.5, contact_gap
    cfg.configure_sdf(max_resolution=32, is_hydroelastic=True)
    cfg.sdf_narrow_band_range = (-narrow_band, narrow_band)
    b.add_ground_plane()
    z0 = []
    for i in range(NUM_CUBES):
        z0.append(cube_half + i * cube_half * 2.0)
        body = b.add_body(xform=[0.0, 0.0, z0[-1], 0, 0, 0, 1])
        b.add_shape_box(body, hx=cube_half, hy=cube_half, hz=cube_half)
    model = b.finalize(device="cuda:0")
    assert bool(model.env.sdf_pair_hydro.all())  # the cube-cube pairs take the SDF-SDF leg
    solver = nt.solvers.SolverXPBD(model, iterations=10)
    H = nt.geometry.HydroelasticSDF
    pipe = nt.CollisionPipeline(model, broad_phase="explicit", sdf_contacts_per_shape=4000, sdf_hydro_faces_per_shape=8000,
                                sdf_hydroelastic_config=H.Config(reduce_contacts=reduce_contacts, anchor_contact=True))
    contacts = pipe.contacts()
    s0, s1, control = model.state(), model.state(), model.control()
    pipe.collide(s0, contacts)
    assert pipe._sdf_leg.overflow(contacts._flat)["candidate_pairs"] == NUM_CUBES - 1  # sdf_sdf pairs after the broad phase
    for _ in range(int(SIM_TIME / SIM_DT)):
        for _ in range(substeps):
            s0.clear_forces()
            pipe.collide(s0, contacts)
            solver.step(s0, s1, control, contacts, SIM_DT / substeps)
            s0, s1 = s1, s0
    torch.cuda.synchronize()
    info = pipe._sdf_leg.overflow(contacts._flat)
    assert not info["overflow"] and info["rows"] > 0, info
    q = s0.body_q.cpu().numpy()
    for i in range(NUM_CUBES):
        displacement = np.linalg.norm(q[i, :3] - np.array([0.0, 0.0, z0[i]]))
        assert displacement < threshold_factor * cube_half, (i, displacement)
        angle = 2.0 * np.arccos(np.clip(abs(q[i, 6]), 0.0, 1.0))
        assert angle < np.radians(MAX_ROTATION_DEG), (i, np.degrees(angle))
