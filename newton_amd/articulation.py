"""eval_fk: joint coordinates -> maximal coordinates (newton/_src/sim/articulation.py:236-573).

Host implementation (numpy, vectorised over environments) used to seed ``body_q`` / ``body_qd`` before stepping;
it is model-preparation code, not part of the per-substep hot path.
"""
from __future__ import annotations

import numpy as np

from .enums import JointType


def _qmul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + bw * ax + ay * bz - by * az, aw * by + bw * ay + az * bx - bz * ax,
                     aw * bz + bw * az + ax * by - bx * ay, aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def _qrot(q, v):
    qv, w = q[..., :3], q[..., 3:4]
    return v * (2.0 * w * w - 1.0) + np.cross(qv, v) * w * 2.0 + qv * np.sum(qv * v, axis=-1, keepdims=True) * 2.0


def _qinv(q):
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def _quat_axis_angle(axis, angle):
    h = 0.5 * np.asarray(angle)[..., None]
    return np.concatenate([axis * np.sin(h), np.cos(h)], axis=-1)


def _quat_from_cols(c0, c1, c2):
    """wp.quat_from_matrix(wp.matrix_from_cols(c0, c1, c2)) for a batch of (near-)orthonormal column triples."""
    m = np.stack([c0, c1, c2], axis=-1)  # [..., row, col]
    out = np.zeros(m.shape[:-2] + (4,))
    for idx in np.ndindex(m.shape[:-2]):
        a = m[idx]
        tr = a[0, 0] + a[1, 1] + a[2, 2]
        if tr >= 0.0:
            h = np.sqrt(tr + 1.0)
            w = 0.5 * h
            h = 0.5 / h
            x, y, z = (a[2, 1] - a[1, 2]) * h, (a[0, 2] - a[2, 0]) * h, (a[1, 0] - a[0, 1]) * h
        else:
            k = int(np.argmax(np.diag(a)))
            i, j = (k + 1) % 3, (k + 2) % 3
            h = np.sqrt((a[k, k] - (a[i, i] + a[j, j])) + 1.0)
            v = [0.0, 0.0, 0.0]
            v[k] = 0.5 * h
            h = 0.5 / h
            v[i] = (a[k, i] + a[i, k]) * h
            v[j] = (a[j, k] + a[k, j]) * h
            w = (a[j, i] - a[i, j]) * h
            x, y, z = v
        q = np.array([x, y, z, w])
        out[idx] = q / np.linalg.norm(q)
    return out


def _xmul(a, b):
    return np.concatenate([_qrot(a[..., 3:], b[..., :3]) + a[..., :3], _qmul(a[..., 3:], b[..., 3:])], axis=-1)


def _xinv(t):
    qi = _qinv(t[..., 3:])
    return np.concatenate([-_qrot(qi, t[..., :3]), qi], axis=-1)


def eval_fk_numpy(model, joint_q, joint_qd):
    """Returns (body_q [B,7], body_qd [B,6]) as float32 arrays."""
    if getattr(model, "is_heterogeneous", False):  # worlds differ: FK per world group, world-major concatenation (hetero.py)
        parts = model.world_groups.parts
        jq = np.split(np.asarray(joint_q, dtype=np.float32), np.cumsum([p.joint_coord_count for p in parts])[:-1])
        jqd = np.split(np.asarray(joint_qd, dtype=np.float32), np.cumsum([p.joint_dof_count for p in parts])[:-1])
        res = [eval_fk_numpy(p, a, b) for p, a, b in zip(parts, jq, jqd)]
        return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])
    t = model.env
    E, nb, nj = t.env_count, t.nb, t.nj
    jq = np.asarray(joint_q, dtype=np.float64).reshape(E, t.nc)
    jqd = np.asarray(joint_qd, dtype=np.float64).reshape(E, t.nd)
    body_q = np.asarray(model.body_q, dtype=np.float64).reshape(E, nb, 7).copy()
    body_qd = np.asarray(model.body_qd, dtype=np.float64).reshape(E, nb, 6).copy()
    com = np.asarray(model.body_com, dtype=np.float64).reshape(E, nb, 3)
    X_p = np.asarray(model.joint_X_p, dtype=np.float64).reshape(E, nj, 7)
    X_c = np.asarray(model.joint_X_c, dtype=np.float64).reshape(E, nj, 7)
    axis_all = np.asarray(model.joint_axis, dtype=np.float64).reshape(E, t.nd, 3)
    art = np.asarray(model.joint_articulation).reshape(E, nj)[0] if nj else []
    ident = np.tile(np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]), (E, 1))
    for j in range(nj):
        if art[j] == -1:
            continue
        jt = int(t.joint_type[j])
        parent, child = int(t.joint_parent[j]), int(t.joint_child[j])
        qs, qds = int(t.joint_q_start[j]), int(t.joint_qd_start[j])
        lin, ang = int(t.joint_lin_count[j]), int(t.joint_ang_count[j])
        X_j = ident.copy()
        v_lin = np.zeros((E, 3))
        v_ang = np.zeros((E, 3))
        if jt == JointType.PRISMATIC:
            ax = axis_all[:, qds]
            X_j[:, :3] = ax * jq[:, qs:qs + 1]
            v_lin = ax * jqd[:, qds:qds + 1]
        elif jt == JointType.REVOLUTE:
            ax = axis_all[:, qds]
            h = 0.5 * jq[:, qs:qs + 1]
            X_j[:, 3:6] = ax * np.sin(h)
            X_j[:, 6:7] = np.cos(h)
            v_ang = ax * jqd[:, qds:qds + 1]
        elif jt == JointType.BALL:
            X_j[:, 3:7] = jq[:, qs:qs + 4]
            v_ang = jqd[:, qds:qds + 3]
        elif jt in (JointType.FREE, JointType.DISTANCE):
            X_j = jq[:, qs:qs + 7].copy()
            v_lin = jqd[:, qds:qds + 3]
            v_ang = jqd[:, qds + 3:qds + 6]
        elif jt == JointType.D6:
            pos = np.zeros((E, 3))
            for k in range(lin):
                pos += axis_all[:, qds + k] * jq[:, qs + k:qs + k + 1]
                v_lin = v_lin + axis_all[:, qds + k] * jqd[:, qds + k:qds + k + 1]
            X_j[:, :3] = pos
            if ang == 1:
                ax = axis_all[:, qds + lin]
                h = 0.5 * jq[:, qs + lin:qs + lin + 1]
                X_j[:, 3:6] = ax * np.sin(h)
                X_j[:, 6:7] = np.cos(h)
                v_ang = ax * jqd[:, qds + lin:qds + lin + 1]
            elif ang == 2:  # compute_2d_rotational_dofs (articulation.py:36-83)
                ax0, ax1 = axis_all[:, qds + lin], axis_all[:, qds + lin + 1]
                q_off = _quat_from_cols(ax0, ax1, np.cross(ax0, ax1))
                a0 = _qrot(q_off, np.broadcast_to([1.0, 0.0, 0.0], (E, 3)))
                local_1 = _qrot(q_off, np.broadcast_to([0.0, 1.0, 0.0], (E, 3)))
                q_0 = _quat_axis_angle(a0, jq[:, qs + lin])
                a1 = _qrot(q_0, local_1)
                q_1 = _quat_axis_angle(a1, jq[:, qs + lin + 1])
                X_j[:, 3:7] = _qmul(q_1, q_0)
                v_ang = a0 * jqd[:, qds + lin:qds + lin + 1] + a1 * jqd[:, qds + lin + 1:qds + lin + 2]
            elif ang == 3:  # compute_3d_rotational_dofs (articulation.py:127-178)
                ax0, ax1, ax2 = (axis_all[:, qds + lin + k] for k in range(3))
                q_0 = _quat_axis_angle(ax0, jq[:, qs + lin])
                a1 = _qrot(q_0, ax1)
                q_1 = _quat_axis_angle(a1, jq[:, qs + lin + 1])
                q_10 = _qmul(q_1, q_0)
                a2 = _qrot(q_10, ax2)
                q_2 = _quat_axis_angle(a2, jq[:, qs + lin + 2])
                X_j[:, 3:7] = _qmul(q_2, q_10)
                v_ang = (ax0 * jqd[:, qds + lin:qds + lin + 1] + a1 * jqd[:, qds + lin + 1:qds + lin + 2]
                         + a2 * jqd[:, qds + lin + 2:qds + lin + 3])
        elif jt == JointType.FIXED:
            pass
        else:
            continue
        X_wpj = X_p[:, j]
        if parent >= 0:
            X_wp = body_q[:, parent]
            X_wpj = _xmul(X_wp, X_wpj)
        X_wcj = _xmul(X_wpj, X_j)
        X_wc = _xmul(X_wcj, _xinv(X_c[:, j]))
        x_child = X_wc[:, :3]
        v_parent = np.zeros((E, 3))
        w_parent = np.zeros((E, 3))
        if parent >= 0:
            qd_p = body_qd[:, parent]
            w_parent = qd_p[:, 3:]
            r = x_child - (X_wp[:, :3] + _qrot(X_wp[:, 3:], com[:, parent]))
            v_parent = np.cross(w_parent, r) + qd_p[:, :3]
        lin_w = _qrot(X_wpj[:, 3:], v_lin)
        ang_w = _qrot(X_wpj[:, 3:], v_ang)
        com_w = _qrot(X_wc[:, 3:], com[:, child])
        if jt in (JointType.FREE, JointType.DISTANCE):
            lin_origin = lin_w - np.cross(ang_w, com_w)
        else:
            lin_origin = lin_w + np.cross(ang_w, x_child - X_wcj[:, :3])
        v_o = v_parent + lin_origin
        w_o = w_parent + ang_w
        body_q[:, child] = X_wc
        body_qd[:, child, :3] = np.cross(w_o, com_w) + v_o
        body_qd[:, child, 3:] = w_o
    return body_q.reshape(-1, 7).astype(np.float32), body_qd.reshape(-1, 6).astype(np.float32)


def _host_array(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def _fk_body_selection(model, mask, indices, body_flag_filter):
    """bool [body_count]: bodies whose inbound joint belongs to a selected articulation and whose flags pass the filter."""
    A = int(model.articulation_count)
    if mask is not None and indices is not None:
        raise ValueError("eval_fk: 'mask' and 'indices' cannot be used together")
    art_sel = np.ones(A, dtype=bool)
    if mask is not None:
        art_sel = np.asarray(mask.detach().cpu().numpy() if hasattr(mask, "detach") else mask).astype(bool).reshape(-1)
        if art_sel.shape[0] != A:
            raise ValueError(f"eval_fk: mask has {art_sel.shape[0]} entries, the model has {A} articulations")
    if indices is not None:
        idx = np.asarray(indices.detach().cpu().numpy() if hasattr(indices, "detach") else indices, dtype=np.int64).reshape(-1)
        art_sel = np.zeros(A, dtype=bool)
        art_sel[idx] = True
    joint_art = np.asarray(model.joint_articulation)
    sel = np.zeros(model.body_count, dtype=bool)
    ok = joint_art >= 0
    sel[np.asarray(model.joint_child)[ok]] = art_sel[joint_art[ok]]
    if body_flag_filter is not None:
        sel &= (np.asarray(model.body_flags) & int(body_flag_filter)) != 0
    return sel


def eval_fk(model, joint_q, joint_qd, state, mask=None, indices=None, body_flag_filter=None):
    """newton.eval_fk(model, joint_q, joint_qd, state, mask=None, indices=None, body_flag_filter=BodyFlags.ALL)
    (newton/_src/sim/articulation.py:500-573): writes state.body_q / state.body_qd.  ``mask`` (bool per articulation) or
    ``indices`` (articulation ids) restrict the update to some articulations, ``body_flag_filter`` to bodies whose flags
    match (e.g. BodyFlags.KINEMATIC to re-pose only the prescribed bodies).  ``state`` may be a State or the Model itself (as
    in example_basic_urdf.py:88)."""
    if mask is not None or indices is not None or body_flag_filter is not None:
        sel = _fk_body_selection(model, mask, indices, body_flag_filter)
        if sel.all():
            return eval_fk(model, joint_q, joint_qd, state)
        from .state import State as _State  # noqa: PLC0415

        scratch = _State(model) if isinstance(state, _State) else None
        if scratch is not None and getattr(model, "is_gpu", False):
            # full FK into a scratch state (one kernel launch), then copy the selected bodies
            import torch  # noqa: PLC0415

            eval_fk(model, joint_q, joint_qd, scratch)
            rows = torch.as_tensor(np.flatnonzero(sel), device=scratch.body_q.device)
            for name in ("body_q", "body_qd"):
                cur, new = getattr(state, name), getattr(scratch, name)
                cur[rows] = new[rows]
                setattr(state, name, cur)
            return
        bq, bqd = eval_fk_numpy(model, _host_array(joint_q), _host_array(joint_qd))
        cur_q = np.array(_host_array(state.body_q), dtype=np.float32).reshape(-1, 7)
        cur_qd = np.array(_host_array(state.body_qd), dtype=np.float32).reshape(-1, 6)
        cur_q[sel], cur_qd[sel] = bq[sel], bqd[sel]
        state.body_q, state.body_qd = cur_q, cur_qd
        return

    def host(x):
        return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)

    from .state import State, pack_soa  # noqa: PLC0415

    t = model.env
    on_device = getattr(model, "is_gpu", False) and isinstance(state, State) and t.nj > 0
    if on_device and not np.any(np.asarray(model.joint_articulation) == -1):
        # device path: one launch of eval_fk_kernel through the C ABI (nt_eval_fk)
        import ctypes as C  # noqa: PLC0415

        from . import _lib  # noqa: PLC0415

        dm = model.device_model()
        jq = pack_soa(model, joint_q, 1, t.nc)
        jqd = pack_soa(model, joint_qd, 1, t.nd)
        d = state._desc()
        _lib.check(dm.lib.nt_eval_fk(C.byref(dm.desc), jq.data_ptr(), jqd.data_ptr(), C.byref(d), dm.stream()), "nt_eval_fk")
        return

    bq, bqd = eval_fk_numpy(model, host(joint_q), host(joint_qd))
    state.body_q = bq
    state.body_qd = bqd
