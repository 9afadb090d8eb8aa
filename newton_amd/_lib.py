"""ctypes binding of libnewton_hip.so (the C ABI declared in include/newton_hip.h, include/newton_hip_broadphase.h and include/newton_hip_mesh.h).

The product path has NO CPU fallback: if the shared library is missing or cannot be loaded, every
solver / collision entry point raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.join(_HERE, "libnewton_hip.so")
# The product loads the in-tree library and nothing else: no override of any kind read from outside the package.  Measurement tools and the no-GPU dry run of the
# GPU test files point the loader elsewhere from OUTSIDE the package, before the first load() (tools/with_lib.py,
# tests/emu/emu_plugin.py assign LIB_PATH); load() says so on stderr whenever that happened.
LIB_PATH = _DEFAULT_LIB

NT_CONTACT_FLOATS = 17
NT_BODY_PARAM_FLOATS = 23
NT_JOINT_PARAM_FLOATS = 14
NT_DOF_PARAM_FLOATS = 11
NT_SHAPE_PARAM_FLOATS = 20

_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)


class nt_model(C.Structure):
    _fields_ = [
        ("env_count", C.c_int32), ("env_stride", C.c_int32), ("nb", C.c_int32), ("nj", C.c_int32), ("nd", C.c_int32),
        ("nc", C.c_int32), ("ntq", C.c_int32), ("ns", C.c_int32), ("ng", C.c_int32), ("np", C.c_int32),
        ("cpp", C.c_int32),
        ("np_analytic", C.c_int32), ("na", C.c_int32), ("max_art_dofs", C.c_int32), ("shape_local0", C.c_int32),
        ("contact_scratch_in_hbm", C.c_int32),
        ("params_uniform", C.c_int32),
        ("body_flags", C.c_void_p), ("joint_type", C.c_void_p), ("joint_enabled", C.c_void_p),
        ("joint_parent", C.c_void_p), ("joint_child", C.c_void_p), ("joint_q_start", C.c_void_p),
        ("joint_qd_start", C.c_void_p), ("joint_tq_start", C.c_void_p), ("joint_lin_count", C.c_void_p),
        ("joint_ang_count", C.c_void_p), ("shape_body", C.c_void_p), ("shape_type", C.c_void_p),
        ("shape_flags", C.c_void_p), ("shape_group", C.c_void_p), ("pair_a", C.c_void_p), ("pair_b", C.c_void_p),
        ("body_joint_start", C.c_void_p), ("body_joint_list", C.c_void_p), ("body_pair_start", C.c_void_p),
        ("body_pair_list", C.c_void_p), ("art_start", C.c_void_p),
        ("shape_mesh_start", C.c_void_p), ("shape_mesh_count", C.c_void_p), ("gshape_id", C.c_void_p),
        ("mesh_points", C.c_void_p),
        ("shape_mesh_bounds", C.c_void_p),
        ("body_param", C.c_void_p), ("gravity", C.c_void_p), ("joint_param", C.c_void_p), ("dof_param", C.c_void_p),
        ("shape_param", C.c_void_p), ("gshape_param", C.c_void_p),
    ]


class nt_state(C.Structure):
    _fields_ = [("body_q", C.c_void_p), ("body_qd", C.c_void_p), ("body_f", C.c_void_p), ("joint_q", C.c_void_p),
                ("joint_qd", C.c_void_p), ("body_parent_f", C.c_void_p)]


class nt_control(C.Structure):
    _fields_ = [("joint_f", C.c_void_p), ("joint_target_q", C.c_void_p), ("joint_target_qd", C.c_void_p)]


class nt_flat_rows(C.Structure):
    _fields_ = [("row_start", C.c_void_p), ("shape0", C.c_void_p), ("shape1", C.c_void_p), ("point0", C.c_void_p),
                ("point1", C.c_void_p), ("offset0", C.c_void_p), ("offset1", C.c_void_p), ("normal", C.c_void_p),
                ("margin0", C.c_void_p), ("margin1", C.c_void_p), ("stiffness", C.c_void_p), ("damping", C.c_void_p),
                ("friction_scale", C.c_void_p), ("body_blk_start", C.c_void_p), ("body_blk_list", C.c_void_p), ("cw", C.c_void_p),
                ("impulse", C.c_void_p), ("restitution", C.c_void_p)]


class nt_contacts(C.Structure):
    _fields_ = [("shape0", C.c_void_p), ("shape1", C.c_void_p), ("data", C.c_void_p), ("env_count", C.c_void_p),
                ("pair_hit", C.c_void_p), ("cw", C.c_void_p), ("cr", C.c_void_p), ("prop", C.c_void_p), ("world_xform", C.c_void_p),
                ("world_aabb_lower", C.c_void_p), ("world_aabb_upper", C.c_void_p), ("flat", nt_flat_rows)]


class nt_sdf_scene(C.Structure):
    _fields_ = [("env_count", C.c_int32), ("env_stride", C.c_int32), ("nb", C.c_int32), ("ns", C.c_int32),
                ("shape_local0", C.c_int32), ("template_pairs", C.c_int32), ("template_pair", C.c_void_p),
                ("gshape_id", C.c_void_p), ("shape_body", C.c_void_p), ("shape_gap", C.c_void_p), ("pairs_per_world", C.c_int32),
                ("template_kind", C.c_void_p), ("world_pair_kind", C.c_void_p)]


class nt_sdf_rows_io(C.Structure):
    _fields_ = [("pair_count", C.c_void_p), ("world_pairs", C.c_void_p), ("blk", C.c_void_p), ("pair_row", C.c_void_p),
                ("row_start", C.c_void_p), ("raw_count", C.c_void_p), ("raw_pair", C.c_void_p), ("raw_key", C.c_void_p),
                ("raw_data", C.c_void_p), ("raw_capacity", C.c_int32), ("row_capacity", C.c_int32), ("shape0", C.c_void_p),
                ("shape1", C.c_void_p), ("point0", C.c_void_p), ("point1", C.c_void_p), ("offset0", C.c_void_p),
                ("offset1", C.c_void_p), ("normal", C.c_void_p), ("margin0", C.c_void_p), ("margin1", C.c_void_p),
                ("key", C.c_void_p), ("raw_rank", C.c_void_p), ("raw_stiffness", C.c_void_p), ("raw_friction", C.c_void_p), ("stiffness", C.c_void_p),
                ("damping", C.c_void_p), ("friction_scale", C.c_void_p), ("raw_base", C.c_int32), ("raw_radius", C.c_void_p)]


class nt_flat_history(C.Structure):
    _fields_ = [("prev_row_start", C.c_void_p), ("prev_pair_count", C.c_void_p), ("prev_world_pairs", C.c_void_p),
                ("prev_pair_row", C.c_void_p), ("prev_pair_rows", C.c_void_p), ("prev_live", C.c_void_p),
                ("prev_pos_world", C.c_void_p), ("prev_normal", C.c_void_p), ("prev_body_frame", C.c_void_p),
                ("prev_claim", C.c_void_p)]


class nt_flat_force_params(C.Structure):
    _fields_ = [("body_q", C.c_void_p), ("body_qd", C.c_void_p), ("body_com", C.c_void_p), ("shape_material", C.c_void_p),
                ("friction_smoothing", C.c_float), ("body_f", C.c_void_p)]


class nt_xpbd_params(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("joint_linear_relaxation", C.c_float),
                ("joint_angular_relaxation", C.c_float), ("joint_linear_compliance", C.c_float),
                ("joint_angular_compliance", C.c_float), ("rigid_contact_relaxation", C.c_float),
                ("rigid_contact_con_weighting", C.c_int32), ("angular_damping", C.c_float),
                ("enable_restitution", C.c_int32), ("compute_body_velocity_from_position_delta", C.c_int32)]


class nt_xpbd_report(C.Structure):
    _fields_ = [("contact_impulse", C.c_void_p), ("joint_impulse", C.c_void_p)]


class nt_broadphase_in(C.Structure):
    _fields_ = [("lower", C.c_void_p), ("upper", C.c_void_p), ("gap", C.c_void_p), ("group", C.c_void_p), ("world", C.c_void_p),
                ("filter_pairs", C.c_void_p), ("num_filter_pairs", C.c_int32), ("include_static_kinematic_pairs", C.c_int32),
                ("shape_body", C.c_void_p), ("body_flags", C.c_void_p)]


class nt_broadphase_motion(C.Structure):
    """include/newton_hip_broadphase.h: per-shape displacement of the swept broad phases."""
    _fields_ = [("displacement", C.c_void_p), ("sort_axis_displacement_limit", C.c_float)]


class nt_mesh_plane_args(C.Structure):
    """include/newton_hip_mesh.h: MESH vs infinite plane (vertex leg)."""
    _fields_ = [("pairs", C.c_void_p), ("pair_count", C.c_int32), ("pair_world_prefix", C.c_void_p), ("worlds", C.c_int32),
                ("pairs_per_world", C.c_int32), ("pair_kind", C.c_void_p), ("shape_type", C.c_void_p), ("shape_transform", C.c_void_p),
                ("shape_data", C.c_void_p), ("shape_gap", C.c_void_p), ("shape_vertex_range", C.c_void_p), ("vertices", C.c_void_p),
                ("shape_aabb_lower", C.c_void_p), ("shape_aabb_upper", C.c_void_p), ("shape_voxel_res", C.c_void_p),
                ("reduce", C.c_int32), ("out_count", C.c_void_p), ("out_pair", C.c_void_p), ("out_key", C.c_void_p),
                ("out_data", C.c_void_p), ("capacity", C.c_int32), ("out_blk", C.c_void_p)]


class nt_heightfield(C.Structure):
    """include/newton_hip_mesh.h: HeightfieldData (utils/heightfield.py:141-156)."""
    _fields_ = [("data_offset", C.c_int32), ("nrow", C.c_int32), ("ncol", C.c_int32), ("hx", C.c_float), ("hy", C.c_float),
                ("min_z", C.c_float), ("max_z", C.c_float)]


class nt_mesh_triangle_args(C.Structure):
    """include/newton_hip_mesh.h: MESH vs convex primitive (triangle leg)."""
    _fields_ = [("pairs", C.c_void_p), ("pair_count", C.c_int32), ("pair_world_prefix", C.c_void_p), ("worlds", C.c_int32),
                ("pairs_per_world", C.c_int32), ("pair_kind", C.c_void_p), ("shape_type", C.c_void_p), ("shape_transform", C.c_void_p),
                ("shape_data", C.c_void_p), ("shape_gap", C.c_void_p), ("shape_vertex_range", C.c_void_p),
                ("shape_triangle_range", C.c_void_p), ("vertices", C.c_void_p), ("indices", C.c_void_p),
                ("shape_aabb_lower", C.c_void_p), ("shape_aabb_upper", C.c_void_p), ("shape_voxel_res", C.c_void_p),
                ("reduce", C.c_int32), ("out_count", C.c_void_p), ("out_pair", C.c_void_p), ("out_key", C.c_void_p),
                ("out_data", C.c_void_p), ("out_radius", C.c_void_p), ("capacity", C.c_int32), ("out_blk", C.c_void_p),
                ("block_bounds", C.c_void_p), ("shape_block_start", C.c_void_p), ("hull_points", C.c_void_p),
                ("shape_hull_range", C.c_void_p), ("shape_heightfield_index", C.c_void_p), ("heightfields", C.c_void_p),
                ("elevations", C.c_void_p)]


class nt_sdf(C.Structure):
    _fields_ = [("coarse", C.c_void_p), ("subgrid", C.c_void_p), ("slots", C.c_void_p), ("cx", C.c_int32), ("cy", C.c_int32),
                ("cz", C.c_int32), ("tex_size", C.c_int32), ("subgrid_size", C.c_int32), ("quantization", C.c_int32),
                ("scale_baked", C.c_int32), ("box_lower", C.c_float * 3), ("box_upper", C.c_float * 3),
                ("inv_dx", C.c_float * 3), ("voxel_size", C.c_float * 3), ("voxel_radius", C.c_float),
                ("min_value", C.c_float), ("value_range", C.c_float)]


class nt_mesh_sdf_args(C.Structure):
    _fields_ = [("pairs", C.c_void_p), ("pair_count", C.c_int32), ("shape_transform", C.c_void_p), ("shape_data", C.c_void_p),
                ("shape_gap", C.c_void_p), ("shape_sdf_index", C.c_void_p), ("sdf_table", C.c_void_p), ("sdf_count", C.c_int32),
                ("shape_edge_range", C.c_void_p), ("edge_centers", C.c_void_p), ("edge_halves", C.c_void_p),
                ("out_count", C.c_void_p), ("out_pair", C.c_void_p), ("out_key", C.c_void_p), ("out_data", C.c_void_p),
                ("capacity", C.c_int32), ("pair_count_device", C.c_void_p), ("pair_world_prefix", C.c_void_p),
                ("worlds", C.c_int32), ("pairs_per_world", C.c_int32), ("out_blk", C.c_void_p), ("pair_kind", C.c_void_p),
                ("hit_count", C.c_void_p), ("hit_stripes", C.c_void_p), ("hit_stripe_count", C.c_int32), ("hit_capacity", C.c_int32),
                ("hit_pair", C.c_void_p), ("hit_fp", C.c_void_p), ("hit_rec", C.c_void_p), ("hit_blk", C.c_void_p),
                ("unit_ctx", C.c_void_p)]


class nt_contact_reduce_shapes(C.Structure):
    _fields_ = [("shape_aabb_lower", C.c_void_p), ("shape_aabb_upper", C.c_void_p), ("shape_voxel_res", C.c_void_p),
                ("threads", C.c_int32), ("shape_edge_radius_max", C.c_void_p), ("keep_all", C.c_int32)]


class nt_contact_reduce_list(C.Structure):
    _fields_ = [("segment_start", C.c_void_p), ("segments", C.c_int32), ("pos", C.c_void_p), ("normal", C.c_void_p),
                ("depth", C.c_void_p), ("fp", C.c_void_p), ("centered", C.c_void_p), ("inner", C.c_void_p), ("outer", C.c_void_p),
                ("local", C.c_void_p), ("aabb_lo", C.c_void_p), ("aabb_hi", C.c_void_p), ("res", C.c_void_p),
                ("out_count", C.c_void_p), ("out_index", C.c_void_p), ("out_normal", C.c_void_p), ("capacity", C.c_int32)]


class nt_contact_rows(C.Structure):
    _fields_ = [("row_count", C.c_int32), ("row_count_device", C.c_void_p), ("row_pair", C.c_void_p), ("pairs", C.c_void_p),
                ("row_data", C.c_void_p), ("body_q", C.c_void_p), ("shape_body", C.c_void_p), ("shape_gap", C.c_void_p),
                ("out_shape0", C.c_void_p), ("out_shape1", C.c_void_p), ("out_point0", C.c_void_p), ("out_point1", C.c_void_p),
                ("out_offset0", C.c_void_p), ("out_offset1", C.c_void_p), ("out_normal", C.c_void_p), ("out_margin0", C.c_void_p),
                ("out_margin1", C.c_void_p)]


class nt_flat_contact_forces(C.Structure):
    _fields_ = [("body_q", C.c_void_p), ("body_qd", C.c_void_p), ("body_com", C.c_void_p), ("shape_ke", C.c_void_p),
                ("shape_kd", C.c_void_p), ("shape_kf", C.c_void_p), ("shape_ka", C.c_void_p), ("shape_mu", C.c_void_p),
                ("shape_body", C.c_void_p), ("contact_count", C.c_void_p), ("contact_max", C.c_int32), ("point0", C.c_void_p),
                ("point1", C.c_void_p), ("normal", C.c_void_p), ("shape0", C.c_void_p), ("shape1", C.c_void_p),
                ("margin0", C.c_void_p), ("margin1", C.c_void_p), ("contact_stiffness", C.c_void_p), ("contact_damping", C.c_void_p),
                ("contact_friction_scale", C.c_void_p), ("friction_smoothing", C.c_float), ("body_f", C.c_void_p)]


class nt_contact_history(C.Structure):
    _fields_ = [("prev_pos_world", C.c_void_p), ("prev_normal", C.c_void_p), ("prev_live", C.c_void_p),
                ("prev_body_frame", C.c_void_p)]


class nt_hydro_args(C.Structure):
    _fields_ = [("pairs", C.c_void_p), ("pair_count", C.c_int32), ("shape_transform", C.c_void_p), ("shape_data", C.c_void_p),
                ("shape_gap", C.c_void_p), ("shape_kh", C.c_void_p), ("shape_sdf_index", C.c_void_p), ("sdf_table", C.c_void_p),
                ("sdf_count", C.c_int32), ("tri_range", C.c_void_p), ("flat_edge_verts", C.c_void_p),
                ("margin_contact_area", C.c_float), ("edge_clamp_min", C.c_float), ("out_count", C.c_void_p),
                ("out_pair", C.c_void_p), ("out_key", C.c_void_p), ("out_shapes", C.c_void_p), ("out_data", C.c_void_p),
                ("capacity", C.c_int32), ("pair_world_prefix", C.c_void_p), ("worlds", C.c_int32), ("pairs_per_world", C.c_int32),
                ("pair_kind", C.c_void_p), ("out_pairs_normalized", C.c_void_p), ("out_blk", C.c_void_p), ("out_rank", C.c_void_p),
                ("out_stiffness", C.c_void_p),
                ("reduce", C.c_int32), ("shape_aabb_lower", C.c_void_p), ("shape_aabb_upper", C.c_void_p), ("shape_voxel_res", C.c_void_p),
                ("face_count", C.c_void_p), ("face_rec", C.c_void_p), ("face_capacity", C.c_int32), ("out_friction", C.c_void_p),
                ("stage_count", C.c_void_p), ("stage_queue", C.c_void_p), ("stage_queue_capacity", C.c_int32),
                ("stage_chunk_capacity", C.c_int32), ("stage_pair", C.c_void_p), ("stage_item", C.c_void_p), ("stage_chunk", C.c_void_p), ("stage_active", C.c_void_p)]


class nt_semi_implicit_params(C.Structure):
    _fields_ = [("angular_damping", C.c_float), ("friction_smoothing", C.c_float), ("joint_attach_ke", C.c_float),
                ("joint_attach_kd", C.c_float)]


class nt_featherstone_params(C.Structure):
    _fields_ = [("angular_damping", C.c_float), ("friction_smoothing", C.c_float), ("update_mass_matrix_interval", C.c_int32),
                ("step_index", C.c_int32), ("force_update", C.c_int32), ("mass_matrix_cache", C.c_void_p),
                ("dense_mass_matrix", C.c_int32)]


class nt_collide_params(C.Structure):
    _fields_ = [("broad_phase", C.c_int32), ("envs_per_block", C.c_int32)]


# every symbol include/*.h declares: name -> (restype, argtypes)
_P = C.c_void_p
class nt_newton_model(C.Structure):
    """Flat newton.Model arrays handed to nt_model_create (host pointers), include/newton_hip.h."""
    _fields_ = [
        ("world_count", C.c_int32),
        ("body_count", C.c_int32),
        ("joint_count", C.c_int32),
        ("shape_count", C.c_int32),
        ("joint_dof_count", C.c_int32),
        ("joint_coord_count", C.c_int32),
        ("joint_target_q_count", C.c_int32),
        ("articulation_count", C.c_int32),
        ("shape_contact_pair_count", C.c_int32),
        ("mesh_point_count", C.c_int32),
        ("gravity_count", C.c_int32),
        ("body_world", _P),
        ("body_flags", _P),
        ("body_com", _P),
        ("body_mass", _P),
        ("body_inv_mass", _P),
        ("body_inertia", _P),
        ("body_inv_inertia", _P),
        ("joint_world", _P),
        ("joint_type", _P),
        ("joint_enabled", _P),
        ("joint_parent", _P),
        ("joint_child", _P),
        ("joint_q_start", _P),
        ("joint_qd_start", _P),
        ("joint_target_q_start", _P),
        ("joint_dof_dim", _P),
        ("joint_X_p", _P),
        ("joint_X_c", _P),
        ("joint_axis", _P),
        ("joint_limit_lower", _P),
        ("joint_limit_upper", _P),
        ("joint_target_ke", _P),
        ("joint_target_kd", _P),
        ("joint_limit_ke", _P),
        ("joint_limit_kd", _P),
        ("joint_armature", _P),
        ("joint_damping", _P),
        ("articulation_start", _P),
        ("articulation_end", _P),
        ("shape_world", _P),
        ("shape_body", _P),
        ("shape_type", _P),
        ("shape_flags", _P),
        ("shape_collision_group", _P),
        ("shape_transform", _P),
        ("shape_scale", _P),
        ("shape_margin", _P),
        ("shape_gap", _P),
        ("shape_material_mu", _P),
        ("shape_material_mu_torsional", _P),
        ("shape_material_mu_rolling", _P),
        ("shape_material_ke", _P),
        ("shape_material_kd", _P),
        ("shape_material_kf", _P),
        ("shape_material_ka", _P),
        ("shape_material_restitution", _P),
        ("shape_contact_pairs", _P),
        ("shape_mesh_start", _P),
        ("shape_mesh_count", _P),
        ("mesh_points", _P),
        ("gravity", _P),
        ("shape_sdf_index", _P),
        ("shape_edge_range", _P),
    ]


SYMBOLS = {
    "nt_model_create": (C.c_int32, [C.POINTER(nt_newton_model), C.c_int32, C.POINTER(_P)]),
    "nt_model_get": (C.POINTER(nt_model), [_P]),
    "nt_model_pair_order": (C.c_int32, [_P, C.POINTER(C.c_int64)]),
    "nt_model_sdf_pairs": (C.c_int32, [_P, C.POINTER(C.c_int32), _P, _P, _P]),
    "nt_model_refresh_params": (C.c_int32, [_P, C.POINTER(nt_newton_model)]),
    "nt_model_destroy": (None, [_P]),
    "nt_model_last_error": (C.c_char_p, []),
    "nt_clear_forces": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_state), _P]),
    "nt_collide": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_state), C.POINTER(nt_contacts),
                                C.POINTER(nt_collide_params), _P]),
    "nt_xpbd_step": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_xpbd_params), C.POINTER(nt_state), C.POINTER(nt_state),
                                  C.POINTER(nt_control), C.POINTER(nt_contacts), C.c_float, C.c_int32,
                                  C.POINTER(nt_xpbd_report), _P]),
    "nt_semi_implicit_step": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_semi_implicit_params), C.POINTER(nt_state),
                                           C.POINTER(nt_state), C.POINTER(nt_control), C.POINTER(nt_contacts), C.c_float,
                                           C.c_int32, _P]),
    "nt_featherstone_step": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_featherstone_params), C.POINTER(nt_state),
                                          C.POINTER(nt_state), C.POINTER(nt_control), C.POINTER(nt_contacts), C.c_float,
                                          C.c_int32, _P]),
    "nt_featherstone_rollout": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_featherstone_params), C.POINTER(nt_collide_params),
                                             C.POINTER(nt_state), C.POINTER(nt_state), C.POINTER(nt_control),
                                             C.POINTER(nt_contacts), C.c_float, C.c_int32, _P]),
    "nt_featherstone_lds_bytes_per_env": (C.c_int32, [C.POINTER(nt_model)]),
    "nt_xpbd_rollout_shape": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_xpbd_params), C.POINTER(nt_collide_params),
                              C.POINTER(C.c_int32)]),
    "nt_xpbd_rollout": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_xpbd_params), C.POINTER(nt_collide_params),
                                     C.POINTER(nt_state), C.POINTER(nt_state), C.POINTER(nt_control),
                                     C.POINTER(nt_contacts), C.c_float, C.c_int32, _P]),
    "nt_state_reset": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_state), C.POINTER(nt_state), _P, _P]),
    "nt_eval_fk": (C.c_int32, [C.POINTER(nt_model), _P, _P, C.POINTER(nt_state), _P]),
    "nt_pack_aos": (C.c_int32, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "nt_unpack_aos": (C.c_int32, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "nt_contacts_export": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_contacts), C.c_int32, _P, _P, _P, _P, _P, _P, _P,
                                        _P, _P, _P, _P, _P]),
    "nt_contacts_export_force": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_contacts), _P, C.c_float, C.c_int32, _P, _P, _P]),
    "nt_broadphase_nxn": (C.c_int32, [C.POINTER(nt_broadphase_in), _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P]),
    "nt_broadphase_sap": (C.c_int32, [C.POINTER(nt_broadphase_in), _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P]),
    "nt_broadphase_sap_device": (C.c_int32, [C.POINTER(nt_broadphase_in), _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P,
                                               _P, _P, C.c_int32, _P]),
    "nt_broadphase_explicit": (C.c_int32, [C.POINTER(nt_broadphase_in), _P, C.c_int32, _P, _P, C.c_int32, _P]),
    # include/newton_hip_mesh.h
    "nt_mesh_plane_pairs": (C.c_int32, [C.POINTER(nt_mesh_plane_args), _P]),
    "nt_mesh_triangle_pairs": (C.c_int32, [C.POINTER(nt_mesh_triangle_args), _P]),
    # include/newton_hip_broadphase.h
    "nt_broadphase_nxn_swept": (C.c_int32, [C.POINTER(nt_broadphase_in), C.POINTER(nt_broadphase_motion), _P, _P, C.c_int32, C.c_int32,
                                            C.c_int32, _P, _P, C.c_int32, _P]),
    "nt_broadphase_sap_device_swept": (C.c_int32, [C.POINTER(nt_broadphase_in), C.POINTER(nt_broadphase_motion), _P, _P, C.c_int32,
                                                   C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32, _P]),
    "nt_broadphase_explicit_swept": (C.c_int32, [C.POINTER(nt_broadphase_in), C.POINTER(nt_broadphase_motion), _P, C.c_int32, _P, _P,
                                                 C.c_int32, _P]),
    "nt_error_string": (C.c_char_p, [C.c_int32]),
    "nt_build_info": (C.c_char_p, []),
    "nt_lds_bytes_per_env": (C.c_int32, [C.POINTER(nt_model)]),
    "nt_pick_envs_per_block": (C.c_int32, [C.POINTER(nt_model), C.c_int32]),
    "nt_calibration_copy": (C.c_int32, [_P, _P, C.c_int64, _P]),
    "nt_bandwidth_probe": (C.c_int32, [_P, _P, C.c_int64, _P]),
    "nt_graph_capture_begin": (C.c_int32, [_P]),
    "nt_graph_capture_end": (C.c_int32, [_P, C.POINTER(_P)]),
    "nt_graph_launch": (C.c_int32, [_P, _P]),
    "nt_graph_destroy": (None, [_P]),
    "nt_sdf_sample": (C.c_int32, [C.POINTER(nt_sdf), _P, C.c_int32, _P, _P, _P]),
    "nt_sdf_sample_hw": (C.c_int32, [C.POINTER(nt_sdf), _P, C.c_int32, _P, _P]),
    "nt_sdf_sample_voxels": (C.c_int32, [C.POINTER(nt_sdf), _P, C.c_int32, _P, _P]),
    "nt_mesh_sdf_collide": (C.c_int32, [C.POINTER(nt_mesh_sdf_args), _P]),
    "nt_mesh_sdf_collide_reduced": (C.c_int32, [C.POINTER(nt_mesh_sdf_args), C.POINTER(nt_contact_reduce_shapes), _P]),
    "nt_contacts_reduce_list": (C.c_int32, [C.POINTER(nt_contact_reduce_list), _P]),
    "nt_contact_rows_write": (C.c_int32, [C.POINTER(nt_contact_rows), _P]),
    "nt_eval_body_contact_flat": (C.c_int32, [C.POINTER(nt_flat_contact_forces), _P]),
    "nt_contacts_match": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_state), C.POINTER(nt_contacts), C.POINTER(nt_contact_history),
                                       C.c_float, C.c_float, _P, _P, _P]),
    "nt_contacts_replay_matched": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_state), C.POINTER(nt_contacts),
                                   C.POINTER(nt_contact_history), C.c_void_p, _P]),
    "nt_contacts_save_history": (C.c_int32, [C.POINTER(nt_model), C.POINTER(nt_state), C.POINTER(nt_contacts),
                                              C.POINTER(nt_contact_history), _P]),
    "nt_hydro_collide": (C.c_int32, [C.POINTER(nt_hydro_args), _P]),
    "nt_hydro_pairs": (C.c_int32, [C.POINTER(nt_hydro_args), _P]),
    "nt_sdf_candidate_pairs": (C.c_int32, [C.POINTER(nt_sdf_scene), _P, _P, _P, _P, _P, _P]),
    "nt_sdf_rows_finalize": (C.c_int32, [C.POINTER(nt_sdf_scene), C.POINTER(nt_sdf_rows_io), _P, _P, _P, _P, _P]),
    "nt_flat_rows_forces": (C.c_int32, [C.POINTER(nt_sdf_scene), C.POINTER(nt_flat_rows), C.POINTER(nt_flat_force_params), _P]),
    "nt_flat_rows_match": (C.c_int32, [C.POINTER(nt_sdf_scene), C.POINTER(nt_sdf_rows_io), _P, C.POINTER(nt_flat_history), C.c_float,
                                       C.c_float, _P, _P]),
    "nt_flat_rows_replay_matched": (C.c_int32, [C.POINTER(nt_sdf_scene), C.POINTER(nt_sdf_rows_io), _P, C.POINTER(nt_flat_history), _P, _P]),
    "nt_flat_rows_save_history": (C.c_int32, [C.POINTER(nt_sdf_scene), C.POINTER(nt_sdf_rows_io), _P, C.POINTER(nt_flat_history), _P]),
}

_lib = None


class NewtonHipError(RuntimeError):
    pass


def load():
    """Load libnewton_hip.so (once). Raises NewtonHipError loudly if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NewtonHipError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(needs hipcc). There is no CPU fallback for the product path.")
    if os.path.abspath(LIB_PATH) != os.path.abspath(_DEFAULT_LIB):
        import sys

        print(f"[newton_amd] library path reassigned by a tool / test harness ({LIB_PATH}): measurement / test use only",
              file=sys.stderr)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise NewtonHipError(f"could not load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != 0:
        msg = load().nt_error_string(status).decode()
        raise NewtonHipError(f"{what} failed: {msg} (status {status})")
