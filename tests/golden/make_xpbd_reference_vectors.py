#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/xpbd_reference_vectors.npz by EXECUTING the reference's own SolverXPBD
(/root/reference/newton/_src/solvers/xpbd/solver_xpbd.py + kernels.py + solvers/solver.py, unmodified) in this container.

warp-lang is not available here, so the reference source runs on tests/golden/refshim (a pure-Python stand-in for the Warp API:
fp32 scalars, builtins in the operation order of oracle/wp_builtins.h; kernels executed thread by thread in ascending tid
order, atomics applied in that order -- the order a serial Warp-CPU launch uses).  Inputs: small scenes built with the in-repo
ModelBuilder and the contacts of the in-repo C++ checker (collision has its own reference-held known answers); outputs: the
states after every reference step.  tests/test_reference_vectors.py holds the C++ checker -- and the HIP path on the GPU --
against these vectors.  Run from the repo root:  python tests/golden/make_xpbd_reference_vectors.py
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(HERE, "refshim"))
import lazy_ref  # noqa: E402

lazy_ref.install(dummies={"newton._src.sim": ("Contacts", "Control", "Model", "State", "ModelBuilder")})
import warp as wp  # noqa: E402  (the shim)

ref = importlib.import_module("newton._src.solvers.xpbd.solver_xpbd")
ref_semi = importlib.import_module("newton._src.solvers.semi_implicit.solver_semi_implicit")
ref_fs = importlib.import_module("newton._src.solvers.featherstone.solver_featherstone")


def arr(a, dtype):
    return wp.to_array(np.asarray(a), dtype)


def ref_model(m):
    """Duck-typed stand-in for newton.Model holding the attributes SolverXPBD reads, as shim arrays."""
    r = types.SimpleNamespace()
    r.device = wp.Device()
    r.requires_grad = False
    for k in ("particle_count", "spring_count", "edge_count", "tet_count", "tri_count"):
        setattr(r, k, 0)
    r.particle_grid = None
    r.body_count, r.joint_count, r.shape_count = len(m.body_mass), len(m.joint_type), len(m.shape_type)
    r.world_count = m.world_count
    r.rigid_contact_count = None
    r.body_com, r.body_mass, r.body_inv_mass = arr(m.body_com, wp.vec3), arr(m.body_mass, float), arr(m.body_inv_mass, float)
    r.body_inertia, r.body_inv_inertia = arr(m.body_inertia, wp.mat33), arr(m.body_inv_inertia, wp.mat33)
    r.body_flags, r.body_world = arr(m.body_flags, int), arr(m.body_world, int)
    r.gravity = arr(np.asarray(m.gravity).reshape(-1, 3), wp.vec3)
    for k in ("joint_type", "joint_parent", "joint_child", "joint_q_start", "joint_qd_start", "joint_target_q_start"):
        setattr(r, k, arr(getattr(m, k), int))
    r.joint_enabled = arr(np.asarray(m.joint_enabled).astype(bool), bool)
    dd = np.asarray(m.joint_dof_dim).reshape(-1, 2)
    r.joint_dof_dim = wp.Array2([[int(a), int(b)] for a, b in dd])
    r.joint_X_p, r.joint_X_c = arr(m.joint_X_p, wp.transform), arr(m.joint_X_c, wp.transform)
    r.joint_axis = arr(np.asarray(m.joint_axis).reshape(-1, 3), wp.vec3)
    for k in ("joint_limit_lower", "joint_limit_upper", "joint_target_ke", "joint_target_kd"):
        setattr(r, k, arr(getattr(m, k), float))
    r.shape_body = arr(m.shape_body, int)
    for k in ("shape_material_mu", "shape_material_mu_torsional", "shape_material_mu_rolling", "shape_material_restitution",
              "shape_material_ke", "shape_material_kd", "shape_material_kf", "shape_material_ka", "joint_limit_ke", "joint_limit_kd",
              "joint_armature", "joint_damping"):
        setattr(r, k, arr(getattr(m, k), float))
    r.joint_q, r.joint_qd = arr(m.joint_q, float), arr(m.joint_qd, float)
    r.joint_dof_count, r.joint_coord_count = m.joint_dof_count, m.joint_coord_count
    r.particle_max_radius = r.particle_cohesion = 0.0
    ctrl = types.SimpleNamespace(joint_f=arr(m.joint_f, float), joint_target_q=arr(m.joint_target_q, float),
                                 joint_target_qd=arr(m.joint_target_qd, float), tet_activations=None)
    r.control = lambda clone_variables=False: ctrl
    r.request_contact_attributes = lambda *a: None
    return r, ctrl


def add_articulation_tables(r, m):
    """What SolverFeatherstone reads beyond the maximal-coordinate solvers: terminal entries of the start arrays,
    articulation ranges, joint_ancestor (builder.py: the joint whose child is this joint's parent)."""
    r.joint_q_start = arr(np.concatenate([np.asarray(m.joint_q_start), [m.joint_coord_count]]), int)
    r.joint_qd_start = arr(np.concatenate([np.asarray(m.joint_qd_start), [m.joint_dof_count]]), int)
    r.articulation_count = int(m.articulation_count)
    r.articulation_start, r.articulation_end = arr(m.articulation_start, int), arr(m.articulation_end, int)
    parent, child = np.asarray(m.joint_parent), np.asarray(m.joint_child)
    joint_of_child = {int(c): j for j, c in enumerate(child)}
    r.joint_ancestor = arr(np.array([joint_of_child.get(int(p), -1) if p >= 0 else -1 for p in parent]), int)
    r.body_q, r.body_qd = arr(m.body_q, wp.transform), arr(m.body_qd, wp.spatial_vector)


def ref_state(body_q, body_qd, body_f=None):
    s = types.SimpleNamespace(requires_grad=False, particle_q=None, particle_qd=None, particle_f=None, body_parent_f=None,
                              particle_count=0, body_count=len(body_q))
    s.body_q, s.body_qd = arr(body_q, wp.transform), arr(body_qd, wp.spatial_vector)
    s.body_f = arr(np.zeros((len(body_q), 6), np.float32) if body_f is None else body_f, wp.spatial_vector)
    return s


def ref_contacts(oc, n, props=None):
    """Reference-shaped Contacts from the flat arrays of the in-repo checker's collide (first n rows are live)."""
    c = types.SimpleNamespace(force=None, rigid_contact_max=max(n, 1), soft_contact_max=0)
    c.rigid_contact_count = arr(np.array([n]), int)
    c.rigid_contact_shape0, c.rigid_contact_shape1 = arr(oc["shape0"][:n], int), arr(oc["shape1"][:n], int)
    for k in ("point0", "point1", "offset0", "offset1", "normal"):
        setattr(c, "rigid_contact_" + k, arr(oc[k][:n], wp.vec3))
    c.rigid_contact_margin0, c.rigid_contact_margin1 = arr(oc["margin0"][:n], float), arr(oc["margin1"][:n], float)
    c.rigid_contact_stiffness = c.rigid_contact_damping = c.rigid_contact_friction = None
    if props is not None:  # per-contact overrides (contacts.py:227-277), the same value on every contact
        c.rigid_contact_stiffness, c.rigid_contact_damping = arr(np.full(n, props[0]), float), arr(np.full(n, props[1]), float)
        c.rigid_contact_friction = arr(np.full(n, props[2]), float)
    c.__bool__ = lambda: True
    return c


def to_np(a, n):
    return np.array([[float(c) for c in x] for x in a], dtype=np.float32).reshape(len(a), n)


def run_case(name, case):
    """Teacher-forced: every step starts from the REFERENCE state of the previous step; contacts from the in-repo checker's
    collide on that state (exactly the arrays a Newton CollisionPipeline would hand to the solver)."""
    import oracle_bridge as ob
    import reference_cases as rc

    model = rc.prepare(case)
    rm, ctrl = ref_model(model)
    kind = case.get("solver", "xpbd")
    if kind == "xpbd":
        solver = ref.SolverXPBD(rm, **case["kw"])
        for k_, v_ in case.get("attrs", {}).items():
            assert hasattr(solver, k_), k_
            setattr(solver, k_, v_)
    elif kind == "semi_implicit":
        solver = ref_semi.SolverSemiImplicit(rm, **case["kw"])
    else:
        add_articulation_tables(rm, model)
        solver = ref_fs.SolverFeatherstone(rm, **case["kw"])
    jq, jqd = np.array(model.joint_q, np.float32), np.array(model.joint_qd, np.float32)
    q, qd = np.array(model.body_q, np.float32).reshape(-1, 7), np.array(model.body_qd, np.float32).reshape(-1, 6)
    out = {"body_q0": q.copy(), "body_qd0": qd.copy(), "joint_q0": jq.copy(), "joint_qd0": jqd.copy()}
    orc = ob.Oracle(model)
    for k in range(case["steps"]):
        ct = orc.contacts()
        orc.collide(q, ct)
        n = int(ct.count[0])
        oc = {f: getattr(ct, f) for f in ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")}
        s_in, s_out = ref_state(q, qd), ref_state(q, qd)
        for s_ in (s_in, s_out):
            s_.joint_q, s_.joint_qd = arr(jq, float), arr(jqd, float)
        rcont = ref_contacts(oc, n, case.get("props")) if n else None
        if case.get("report"):
            s_out.body_parent_f = wp.zeros(len(q), dtype=wp.spatial_vector)
            if rcont is not None:
                rcont.force = wp.zeros(rcont.rigid_contact_max, dtype=wp.spatial_vector)
        solver.step(s_in, s_out, None, rcont, case["dt"])
        if case.get("report"):
            out[f"body_parent_f{k + 1}"] = to_np(s_out.body_parent_f, 6)
            if rcont is not None:
                solver.update_contacts(rcont)
                out[f"contact_force{k + 1}"] = to_np(rcont.force, 6)
        q, qd = to_np(s_out.body_q, 7), to_np(s_out.body_qd, 6)
        if kind == "featherstone":
            jq, jqd = np.array(s_out.joint_q, np.float32), np.array(s_out.joint_qd, np.float32)
        out[f"body_q{k + 1}"], out[f"body_qd{k + 1}"], out[f"contacts{k}"] = q.copy(), qd.copy(), np.array([n])
        out[f"joint_q{k + 1}"], out[f"joint_qd{k + 1}"] = jq.copy(), jqd.copy()
        print(name, "step", k, "contacts", n, "max |qd|", float(np.abs(qd).max()), flush=True)
    return out


def run_fk(name, make):
    """newton.eval_fk (newton/_src/sim/articulation.py:500-573) of the reference on a model's joint state."""
    art = importlib.import_module("newton._src.sim.articulation")
    model = make()
    rm, _ = ref_model(model)
    add_articulation_tables(rm, model)
    nj = len(model.joint_type)
    ja = np.zeros(nj, np.int32)
    for a, (s0, s1) in enumerate(zip(np.asarray(model.articulation_start), np.asarray(model.articulation_end))):
        ja[s0:s1] = a
    rm.joint_articulation = arr(ja, int)
    target = types.SimpleNamespace(body_q=wp.zeros(rm.body_count, dtype=wp.transform), body_qd=wp.zeros(rm.body_count, dtype=wp.spatial_vector))
    art.eval_fk(rm, arr(model.joint_q, float), arr(model.joint_qd, float), target)
    print(name, "fk bodies", rm.body_count, flush=True)
    return {"joint_q": np.array(model.joint_q, np.float32), "joint_qd": np.array(model.joint_qd, np.float32),
            "body_q": to_np(target.body_q, 7), "body_qd": to_np(target.body_qd, 6)}


def main():
    sys.path.insert(0, HERE)
    import reference_cases as rc

    blob = {}
    for name, make in rc.fk_cases().items():
        for k, v in run_fk(name, make).items():
            blob[f"fk/{name}/{k}"] = v
    for name, case in rc.cases().items():
        for k, v in run_case(name, case).items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "xpbd_reference_vectors.npz"), **blob)
    print("wrote", len(blob), "arrays")


if __name__ == "__main__":
    main()
