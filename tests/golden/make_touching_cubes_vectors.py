#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- tests/golden/touching_cubes_reference_vectors.npz: the reference's own collision kernels (executed on
tests/golden/refshim through make_collide_reference_vectors.reference_collide) on a state of the contact-force scene of
newton/tests/test_solver_xpbd.py:845-1135 in which its two bottom cubes -- placed EXACTLY face to face by that test -- get a
degenerate MPR / GJK result: one contact 0.65 m from the centres of 0.5 m cubes, normal tilted 18 degrees, i.e. a 0.3 m
"penetration" that XPBD answers by throwing the cubes apart at 60 m/s.  The state was reached by the HIP path (round 6, frame 110 of
the test; the round-5 arithmetic walked past it), recorded on the MI355X and is replayed here: reference, checker and device produce
the same contact bit for bit, so this is pinned as REFERENCE BEHAVIOUR -- and tests/test_contact_force.py keeps its pyramid cubes
1 mm apart instead of exactly touching.  Run from the repo root:  python tests/golden/make_touching_cubes_vectors.py <state.npz>
(<state.npz>: body_q of the six bodies, as recorded by tools/contact_force_blowup.py on the GPU box)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)


def scene(gap: float = 0.0, device=None):
    """The scene of test_contact_forces_sum_to_weight; gap: clearance between the two bottom cubes of the pyramid."""
    import newton_amd as nt

    I4, h = [0.0, 0.0, 0.0, 1.0], 0.5
    b = nt.ModelBuilder()
    b.add_ground_plane()
    b.default_shape_cfg.density = 1000.0
    s = b.add_body(xform=[0.0, 0.0, 0.25, *I4])
    b.add_shape_sphere(s, radius=0.25)
    b.default_shape_cfg.density = 2000.0
    s = b.add_body(xform=[10.0, 0.0, 0.5, *I4])
    b.add_shape_sphere(s, radius=0.5)
    b.default_shape_cfg.density = 1000.0
    for x, z in ((20.0, h), (30.0 - h - 0.5 * gap, h), (30.0 + h + 0.5 * gap, h), (30.0, 3.0 * h)):
        c = b.add_body(xform=[x, 0.0, z, *I4])
        b.add_shape_box(c, hx=h, hy=h, hz=h)
    return b.finalize(device=device)


def main():
    import make_collide_reference_vectors as gen
    import oracle_bridge as ob

    q = np.load(sys.argv[1])["q"][-1].astype(np.float32)
    model = scene()
    orc = ob.Oracle(model)
    ct = orc.contacts()
    pairs, _, _ = orc.collide(q, ct)
    res = gen.reference_collide(model, q, pairs, ct.max)
    blob = {"body_q": q, "pairs": np.asarray(pairs, np.int32)}
    blob.update({k: v for k, v in res.items()})
    np.savez_compressed(os.path.join(HERE, "touching_cubes_reference_vectors.npz"), **blob)
    n = int(res["count"][0])
    i = [k for k in range(n) if (res["shape0"][k], res["shape1"][k]) == (4, 5)]
    print("reference contacts", n, "; cubes 4-5:", [(res["point0"][k].tolist(), res["normal"][k].tolist()) for k in i])


if __name__ == "__main__":
    main()
