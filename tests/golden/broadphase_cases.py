"""Case table shared by tests/golden/make_broadphase_reference_vectors.py (which runs the REFERENCE broad-phase classes) and
tests/test_reference_vectors.py (which holds the checker against the record)."""
import numpy as np

CASES = ["single_world", "multiple_worlds", "shape_flags", "per_shape_gap"]


def variants(name):
    from test_broad_phase_standalone import make_case

    lower, upper, gap, group, world, flags = make_case(name)
    n = lower.shape[0]
    rng = np.random.default_rng(len(name))
    base = dict(lower=lower, upper=upper, gap=gap, group=group, world=world, flags=flags, filter_pairs=None, shape_body=None,
                body_flags=None, include=True)
    # explicit list: a random half of all i < j pairs, shuffled (BroadPhaseExplicit only tests the AABBs of the pairs it is given)
    ii, jj = np.triu_indices(n, 1)
    pick = rng.permutation(len(ii))[: len(ii) // 2]
    base["explicit_pairs"] = np.stack([ii[pick], jj[pick]], axis=1).astype(np.int32)
    out = {"plain": base}
    # excluded pairs: sorted lexicographically, as the reference requires for its binary search (broad_phase_common.py:132-170)
    k = rng.permutation(len(ii))[: len(ii) // 5]
    fp = np.stack([ii[k], jj[k]], axis=1).astype(np.int32)
    fp = fp[np.lexsort((fp[:, 1], fp[:, 0]))]
    out["filtered"] = dict(base, filter_pairs=fp)
    # immovable-pair filter: static shapes (body -1), kinematic bodies (flag bit 0), dynamic bodies
    nb = 12
    shape_body = rng.integers(-1, nb, size=n).astype(np.int32)
    body_flags = np.where(rng.random(nb) < 0.4, 1, 0).astype(np.int32)
    out["immovable"] = dict(base, shape_body=shape_body, body_flags=body_flags, include=False)
    # swept AABBs (speculative-contact broad phase): a displacement per shape of the order of the box spacing; "swept_capped"
    # additionally caps the projected displacement of the sort-and-sweep intervals (sort_axis_displacement_limit), which drops
    # fast pairs from the SAP result but not from N x N; "swept_filtered" adds excluded pairs and the immovable filter
    rng = np.random.default_rng(1000 + len(name))
    disp = (rng.standard_normal((n, 3)) * 0.6).astype(np.float32)
    disp[rng.random(n) < 0.25] = 0.0          # resting shapes: the delta == 0 branch on every axis of some pairs
    disp[rng.random(n) < 0.2, 1] = 0.0        # axis-aligned motion: delta == 0 on a single axis
    out["swept"] = dict(base, displacement=disp, limit=None)
    out["swept_capped"] = dict(base, displacement=disp, limit=0.05)
    out["swept_capped_zero"] = dict(base, displacement=disp, limit=0.0)  # intervals not extended at all: the most pairs lost
    out["swept_filtered"] = dict(base, displacement=disp, limit=None, filter_pairs=fp, shape_body=shape_body, body_flags=body_flags,
                                 include=False)
    return out
