"""Uniform-parameter tile of the fused XPBD rollout (nt_model.params_uniform; one block-shared parameter copy per workgroup,
SURVEY.md section 8 row (a)1-3 / DESIGN.md "uniform tile"): emulated kernels, bitwise against the per-environment tile, plus the
host-side detection.  The GPU twin lives in tests/test_zx_round2_gpu.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


@pytest.fixture(scope="module")
def H():
    import harness

    harness.lib()
    return harness


def _rollout(H, model, cfg, substeps=6, uniform=None):
    em = H.EmuModel(model)
    if uniform is not None:
        em.desc.params_uniform = uniform
    a, b, ct, ctrl = H.EmuState(em), H.EmuState(em), H.EmuContacts(em), H.EmuControl(em)
    ctrl.joint_f[:] = 0.3  # exercise the control rows that stay per environment
    old = os.environ.get("NT_XPBD_CFG")
    try:
        if cfg:
            os.environ["NT_XPBD_CFG"] = cfg
        else:
            os.environ.pop("NT_XPBD_CFG", None)
        H.xpbd_rollout(em, a, b, ctrl, ct, 1e-3, substeps, iterations=2)
    finally:
        if old is None:
            os.environ.pop("NT_XPBD_CFG", None)
        else:
            os.environ["NT_XPBD_CFG"] = old
    out = a if substeps % 2 == 0 else b
    return em, out.body_q.copy(), out.body_qd.copy(), ct.data.copy(), ct.shape0.copy()


@pytest.mark.parametrize("cfg", ["16,512,1,1", "16,256,2,1", "32,512,1,1", "8,128,4,1"])
def test_uniform_tile_is_bitwise_the_per_environment_tile(H, cfg):
    from scenes import quadruped_scene

    import newton_amd as nt

    model = quadruped_scene(40, seed=5)  # 40 envs: ragged last tile for every shape; seed: per-env STATE jitter, same parameters
    model.joint_q.reshape(40, -1)[:, 2] -= 0.24  # feet in the ground: live contacts, so that the contact records are compared too
    model.body_q, model.body_qd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
    em, q0, qd0, cd0, s0 = _rollout(H, model, "16,512,1,0")
    assert em.desc.params_uniform == 1
    _, q1, qd1, cd1, s1 = _rollout(H, model, cfg)
    assert np.array_equal(q0.view(np.int32), q1.view(np.int32)) and np.array_equal(qd0.view(np.int32), qd1.view(np.int32))
    # the Contacts of the launch: ids everywhere, records of the live slots (a dead slot keeps whatever an earlier substep left there on
    # the tiles that write HBM every substep, and nothing on the LDS-record tiles, which write the last substep's records only)
    live = np.broadcast_to((s0 >= 0)[None, ...], cd0.shape) if cd0.ndim == s0.ndim + 1 else None
    assert np.array_equal(s0, s1)
    if live is not None:
        assert np.array_equal(cd0.view(np.int32)[live], cd1.view(np.int32)[live]) and live.any()
    else:
        assert np.array_equal(cd0.view(np.int32), cd1.view(np.int32))
    assert np.abs(qd0).max() > 0.0


def test_randomised_worlds_are_detected_and_refused_by_the_uniform_tile(H):
    from scenes import quadruped_scene

    from newton_amd import _lib

    model = quadruped_scene(8, seed=5)
    model.body_mass = np.array(model.body_mass, copy=True)
    model.body_mass[13 * 3 + 2] *= 1.25  # one link of world 3
    model.body_inv_mass = np.where(model.body_mass > 0, 1.0 / np.maximum(model.body_mass, 1e-30), 0.0).astype(np.float32)
    em = H.EmuModel(model)
    assert em.desc.params_uniform == 0
    with pytest.raises(Exception):
        _rollout(H, model, "16,256,2,1")  # NT_ERR_UNSUPPORTED: the shape is only valid for uniform models
    _rollout(H, model, None)  # the default dispatch takes the per-environment tile


def test_candidate_compaction_of_staged_tiles_is_bitwise_the_plain_pair_loop(H):
    """8-box stacks: 36 candidate pairs per environment.  At 8 environments per workgroup (32 slot lanes) the pair interval
    compacts each environment's AABB hits into an LDS list first (nt_collide.hpp: phase_pair_broad_staged); at 1 environment per
    workgroup (256 lanes) every pair has its own lane and nothing is compacted.  Same contacts, same states, bit for bit."""
    from scenes import box_stack_scene

    model = box_stack_scene(11, n_boxes=8, seed=3, jitter=2e-3)
    outs = []
    for epb in (8, 1):
        em = H.EmuModel(model)
        assert em.t.np == 36
        a, b, ct, ctrl = H.EmuState(em), H.EmuState(em), H.EmuContacts(em), H.EmuControl(em)
        H.xpbd_rollout(em, a, b, ctrl, ct, 1.0 / 240.0, 4, epb=epb, iterations=4)
        outs.append((a.body_q.copy(), a.body_qd.copy(), ct.data.copy(), ct.shape0.copy(), ct.env_count.copy()))
    for x, y in zip(*outs):
        assert np.array_equal(x.view(np.int32) if x.dtype == np.float32 else x, y.view(np.int32) if y.dtype == np.float32 else y)
    assert outs[0][4][:11].min() > 0


@pytest.mark.parametrize("cfg", ["8,256,2,1,1", "16,512,1,1,1", "16,256,2,1,1"])
def test_convex_uniform_tile_is_bitwise_the_per_environment_tile(H, cfg):
    """Box stacks (MPR / GJK pairs): the uniform-parameter tiles of the convex rollout against the default per-environment tile."""
    from scenes import box_stack_scene

    model = box_stack_scene(19, n_boxes=5, seed=7, jitter=2e-3)
    outs = []
    for c in (None, cfg):
        em = H.EmuModel(model)
        assert em.desc.params_uniform == 1 and em.t.np_analytic < em.t.np
        a, b, ct, ctrl = H.EmuState(em), H.EmuState(em), H.EmuContacts(em), H.EmuControl(em)
        old = os.environ.pop("NT_XPBD_CFG", None)
        try:
            if c:
                os.environ["NT_XPBD_CFG"] = c
            H.xpbd_rollout(em, a, b, ctrl, ct, 1.0 / 240.0, 4, epb=8 if c is None else 0, iterations=4)
        finally:
            os.environ.pop("NT_XPBD_CFG", None)
            if old is not None:
                os.environ["NT_XPBD_CFG"] = old
        outs.append((a.body_q.copy(), a.body_qd.copy(), ct.data.copy(), ct.shape0.copy()))
    for x, y in zip(*outs):
        assert np.array_equal(x.view(np.int32) if x.dtype == np.float32 else x, y.view(np.int32) if y.dtype == np.float32 else y)
