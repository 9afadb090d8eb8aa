"""Seeded RANDOM cases for the executed-reference fixtures (tests/golden/make_fuzz_reference_vectors.py runs the reference's solver
source on them; tests/test_fuzz_reference_vectors.py holds the checker -- and on the GPU the HIP path -- against the record): random
joint trees over every joint type, several / no shapes per body, every primitive and hull type, collision groups, filter pairs,
disabled joints, per-world mass / friction jitter, random states (tests/fuzz_scenes.py).  The hand-picked cases of
reference_cases.py pin named features; these pin whatever a seed happens to build."""
import numpy as np

N_CASES = 50
SEED0 = 4100


def _kw_xpbd(rng):
    kw = dict(iterations=int(rng.integers(1, 5)))
    if rng.random() < 0.3:
        kw["rigid_contact_con_weighting"] = False
    if rng.random() < 0.3:
        kw["enable_restitution"] = True
    if rng.random() < 0.4:
        kw["joint_linear_compliance"] = float(rng.choice([0.0, 1e-5, 1e-3]))
        kw["joint_angular_compliance"] = float(rng.choice([0.0, 1e-4, 1e-2]))
    if rng.random() < 0.4:
        kw["angular_damping"] = float(rng.uniform(0.0, 0.2))
    if rng.random() < 0.3:
        kw["rigid_contact_relaxation"] = float(rng.uniform(0.5, 1.0))
        kw["joint_linear_relaxation"] = float(rng.uniform(0.4, 1.0))
        kw["joint_angular_relaxation"] = float(rng.uniform(0.3, 1.0))
    return kw


def cases():
    from fuzz_scenes import random_scene

    out = {}
    for i in range(N_CASES):
        seed = SEED0 + i
        rng = np.random.default_rng(seed ^ 0x5EED)
        pick = i % 10
        if pick < 6:  # SolverXPBD, articulated or free bodies
            free = pick == 5
            out[f"fuzz/xpbd_{seed}"] = dict(
                scene=(lambda s=seed, f=free: random_scene(s, world_count=2, articulated=not f, allow_hull=True, extras=(s % 3 == 0))),
                steps=2, dt=float(rng.choice([1e-3, 2e-3, 1.0 / 240.0])), kw=_kw_xpbd(rng), joint_f=lambda nd, s=seed: np.random.default_rng(s).normal(0, 1.0, nd).astype(np.float32))
        elif pick < 8:
            out[f"fuzz/semi_{seed}"] = dict(
                scene=(lambda s=seed: random_scene(s, world_count=2, articulated=True, allow_hull=True)), steps=2, dt=float(rng.choice([1e-4, 2e-4])),
                solver="semi_implicit", kw=dict(angular_damping=float(rng.uniform(0.0, 0.1)), friction_smoothing=float(rng.uniform(0.3, 1.0))))
        else:
            out[f"fuzz/fs_{seed}"] = dict(
                scene=(lambda s=seed: random_scene(s, world_count=2, articulated=True, allow_hull=True, featherstone_compatible=True)),
                steps=2, dt=float(rng.choice([5e-4, 1e-3])), solver="featherstone", kw=dict(angular_damping=float(rng.uniform(0.0, 0.1))))
    return out
