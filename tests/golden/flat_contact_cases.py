"""Case table shared by tests/golden/make_flat_contact_reference_vectors.py (which runs the REFERENCE's write_contact and
eval_body_contact) and the tests: a handful of bodies and shapes (some static, body -1), per-shape materials, and ContactData
rows the way the reduced mesh-SDF stage emits them (world point, normal a -> b, distance, margins), some of them beyond the gap."""
import numpy as np

CASES = {"dynamic_pairs": dict(seed=1, bodies=6, shapes=9, rows=60, static=0, props=False),
         "with_static_shapes": dict(seed=2, bodies=4, shapes=8, rows=80, static=3, props=False),
         "per_contact_properties": dict(seed=3, bodies=5, shapes=7, rows=50, static=1, props=True)}


def make(name):
    c = CASES[name]
    rng = np.random.default_rng(c["seed"])
    B, S, N = c["bodies"], c["shapes"], c["rows"]
    q = rng.normal(size=(B, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    body_q = np.concatenate([rng.uniform(-0.5, 0.5, size=(B, 3)), q], axis=1).astype(np.float32)
    body_qd = rng.uniform(-1.0, 1.0, size=(B, 6)).astype(np.float32)
    body_com = rng.uniform(-0.05, 0.05, size=(B, 3)).astype(np.float32)
    shape_body = rng.integers(0, B, size=S).astype(np.int32)
    shape_body[: c["static"]] = -1
    mat = dict(ke=rng.uniform(1e3, 1e4, S), kd=rng.uniform(10, 100, S), kf=rng.uniform(100, 1000, S),
               ka=rng.uniform(0.0, 0.002, S), mu=rng.uniform(0.2, 1.0, S))
    mat = {k: v.astype(np.float32) for k, v in mat.items()}
    shape_gap = rng.uniform(0.002, 0.01, S).astype(np.float32)
    shape_margin = rng.uniform(0.0, 0.003, S).astype(np.float32)
    a = rng.integers(0, S - 1, size=N)
    b = np.array([rng.integers(x + 1, S) for x in a])
    nrm = rng.normal(size=(N, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    center = rng.uniform(-0.6, 0.6, size=(N, 3))
    msum = shape_margin[a] + shape_margin[b]
    gsum = shape_gap[a] + shape_gap[b]
    dist = msum + rng.uniform(-0.01, 1.3, size=N) * gsum  # some rows beyond the gap: write_contact rejects them
    dist[::7] = (msum + gsum)[::7]                        # exactly on the threshold
    rows = dict(shape_a=a.astype(np.int32), shape_b=b.astype(np.int32), center=center.astype(np.float32),
                normal=nrm.astype(np.float32), distance=dist.astype(np.float32), margin_a=shape_margin[a], margin_b=shape_margin[b],
                key=np.arange(N, dtype=np.int32) * 4)
    props = None
    if c["props"]:
        props = dict(stiffness=np.where(rng.random(N) < 0.5, rng.uniform(1e3, 1e5, N), 0.0).astype(np.float32),
                     damping=np.where(rng.random(N) < 0.5, rng.uniform(1, 50, N), 0.0).astype(np.float32),
                     friction=np.where(rng.random(N) < 0.5, rng.uniform(0.5, 1.5, N), 0.0).astype(np.float32))
    return dict(body_q=body_q, body_qd=body_qd, body_com=body_com, shape_body=shape_body, mat=mat, shape_gap=shape_gap,
                rows=rows, props=props, friction_smoothing=np.float32(1.0))
