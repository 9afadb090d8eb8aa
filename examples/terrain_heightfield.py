"""Convex shapes dropped on terrain -- a `newton.Heightfield` or the same surface as a triangle mesh -- with SolverXPBD: the caller loop
of the reference's terrain examples (`CollisionPipeline.collide` + `SolverXPBD.step` per substep) on `newton_amd`.

Every (shape, terrain) pair goes through the triangle leg of the collision pipeline (newton/_src/geometry/narrow_phase.py:553-583,
633-638, 1455-1665): a grid-cell (heightfield) or block-bounds (mesh) midphase, GJK / MPR + manifold per
triangle, the reference's buffered contact reduction.

    python examples/terrain_heightfield.py --worlds 256 --frames 120 [--mesh]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import newton_amd as nt  # noqa: E402


def surface(x, y):
    return 0.05 * np.sin(3.0 * x) * np.cos(2.5 * y)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, default=256)
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--substeps", type=int, default=10)
    ap.add_argument("--mesh", action="store_true", help="the terrain as a triangle mesh instead of a Heightfield")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args()

    n, half = 49, 2.0
    xs = np.linspace(-half, half, n)
    heights = np.array([[surface(x, y) for x in xs] for y in xs], np.float32)  # [row = y][col = x]

    env = nt.ModelBuilder()
    env.default_shape_cfg.gap = 0.004
    env.default_shape_cfg.mu = 0.6
    spots = [(-1.0, -1.0), (0.0, -1.0), (1.0, -1.0), (-1.0, 0.5), (0.0, 0.5), (1.0, 0.5)]
    for k, (x, y) in enumerate(spots):
        b = env.add_body(xform=[x, y, surface(x, y) + 0.15, 0.0, 0.0, 0.0, 1.0])
        if k % 3 == 0:
            env.add_shape_box(b, hx=0.08, hy=0.06, hz=0.05)
        elif k % 3 == 1:
            env.add_shape_sphere(b, radius=0.06)
        else:
            env.add_shape_capsule(b, radius=0.04, half_height=0.08)
    scene = nt.ModelBuilder()
    scene.default_shape_cfg.gap = 0.004
    scene.default_shape_cfg.mu = 0.6
    scene.replicate(env, args.worlds)
    if args.mesh:
        pts = np.array([(x, y, heights[j, i]) for j, y in enumerate(xs) for i, x in enumerate(xs)], np.float32)
        tri = []
        for j in range(n - 1):
            for i in range(n - 1):
                a, b, c, d = j * n + i, j * n + i + 1, (j + 1) * n + i, (j + 1) * n + i + 1
                tri += [a, b, d, a, d, c]
        scene.add_shape_mesh(-1, mesh=nt.Mesh(pts, np.array(tri, np.int32)))
    else:
        scene.add_shape_heightfield(heightfield=nt.Heightfield(heights, n, n, hx=half, hy=half))
    model = scene.finalize(device=args.device)

    pipeline = nt.CollisionPipeline(model, broad_phase="sap")
    contacts = pipeline.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=4)
    state_0, state_1 = model.state(), model.state()
    dt = 1.0 / 60.0 / args.substeps
    t0 = time.perf_counter()
    for _ in range(args.frames):
        for _ in range(args.substeps):
            state_0.clear_forces()
            pipeline.collide(state_0, contacts)
            solver.step(state_0, state_1, None, contacts, dt)
            state_0, state_1 = state_1, state_0
    q = state_0.body_q.cpu().numpy().reshape(-1, 7)
    wall = time.perf_counter() - t0
    clearance = q[:, 2] - surface(q[:, 0], q[:, 1])
    print(f"{args.worlds} worlds x {len(spots)} shapes on a {'mesh' if args.mesh else 'heightfield'} terrain: {args.frames} frames in {wall:.2f} s "
          f"({args.worlds * args.frames * args.substeps / wall / 1e6:.3f} M env-steps/s incl. Python); clearance above the surface "
          f"min {clearance.min():.3f} m, max {clearance.max():.3f} m; contacts in the last substep {int(contacts.rigid_contact_count.item())}")
    assert np.all(np.isfinite(q)) and clearance.min() > 0.02, "a shape fell through the terrain"


if __name__ == "__main__":
    main()
