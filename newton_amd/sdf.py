"""Sparse "texture" SDFs: host-side construction and the sampling arithmetic.

Restates newton/_src/geometry/sdf_texture.py:
  * primitive sources       `_query_primitive_sdf` :184-201 (geometry/kernels.py:181-626)
  * sparse construction     `_build_sparse_sdf` :1785-2142 -- coarse grid at the subgrid corners, occupancy test at the subgrid
                            centres (`check_subgrid_occupied_kernel` :344-368), linearity demotion (:371-451, :707-726),
                            packed (subgrid_size+1)^3 blocks with uint16 / uint8 / float32 storage (:487-690), indirection
                            slots with 10-bit block coordinates, SLOT_EMPTY / SLOT_LINEAR sentinels (:44-45)
  * sizing                  `_create_texture_sdf_from_source` :2337-2398, `create_texture_sdf_from_mesh/primitive` :2401-2551
  * sampling                `_locate_cell_coords` :786-828, `_texture_sample_sdf_variant` :1008-1126 (software trilinear over
                            point-sampled texels, |p - clamp(p)| extension outside the box), `texture_sample_sdf_grad`
                            :1560-1616, `texture_sample_sdf_at_voxel` :949-1004

The reference bakes on the GPU with Warp kernels and stores the grids in CUDA 3-D textures that it then point-samples at
texel centres; here the bake is vectorised numpy on the host (it runs once per asset) and the "textures" are the plain 3-D
arrays they always were semantically.  `TextureSDF.sample / sample_grad` restate the sampler in
float32 numpy; the device samplers (csrc/nt_sdf.hpp, C ABI nt_sdf_sample) and the test-side checker follow the same operation order.
Mesh sources use exact point-triangle distances with the generalized winding number for the sign (the reference's default
SIGN_MODE_WINDING, `get_distance_to_mesh`), brute force over the triangles -- a host job for assets of 10^2..10^4 faces.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .enums import GeoType

SLOT_EMPTY = np.uint32(0xFFFFFFFF)   # no subgrid data (far-field cell)
SLOT_LINEAR = np.uint32(0xFFFFFFFE)  # subgrid demoted to coarse interpolation


class QuantizationMode:
    FLOAT32 = 4
    UINT16 = 2
    UINT8 = 1


# ------------------------------------------------------------------------------------------------
# analytic primitive SDFs (geometry/kernels.py), vectorised over points [N,3]
# ------------------------------------------------------------------------------------------------
def _capped_cone_z(bottom_radius, top_radius, half_height, p):
    q0 = np.linalg.norm(p[:, :2], axis=1)
    q1 = p[:, 2]
    k1 = np.array([top_radius, half_height])
    k2 = np.array([top_radius - bottom_radius, 2.0 * half_height])
    ca0 = q0 - np.minimum(q0, np.where(q1 < 0.0, bottom_radius, top_radius))
    ca1 = np.abs(q1) - half_height
    denom = float(np.dot(k2, k2))
    t = np.zeros_like(q0)
    if denom > 0.0:
        t = np.clip(((k1[0] - q0) * k2[0] + (k1[1] - q1) * k2[1]) / denom, 0.0, 1.0)
    cb0 = q0 - k1[0] + k2[0] * t
    cb1 = q1 - k1[1] + k2[1] * t
    sign = np.where((cb0 < 0.0) & (ca1 < 0.0), -1.0, 1.0)
    return sign * np.sqrt(np.minimum(ca0 * ca0 + ca1 * ca1, cb0 * cb0 + cb1 * cb1))


def _barrel_cylinder_z(radius, half_height, barrel_radius, p):
    """`_sdf_barrel_cylinder_data_z` (geometry/kernels.py:347-383): exact signed distance to a z-up barrel cylinder -- the closer of the
    circular side arc (clamped to where it meets the end ring) and the end disk in the radial / axial half plane."""
    radial = np.sqrt(p[:, 0] ** 2 + p[:, 1] ** 2)
    z_abs = np.abs(p[:, 2])
    end_radial_offset = np.sqrt(max(barrel_radius * barrel_radius - half_height * half_height, 0.0))
    center = radius - end_radial_offset
    d0, d1 = radial - center, z_abs
    dl = np.sqrt(d0 * d0 + d1 * d1)
    far = dl > 1.0e-8
    safe = np.where(far, dl, 1.0)
    arc_radial = np.where(far, center + barrel_radius * d0 / safe, center + barrel_radius)
    arc_z = np.where(far, barrel_radius * d1 / safe, 0.0)
    on_cap = arc_radial - center < end_radial_offset
    arc_radial = np.where(on_cap, radius, arc_radial)
    arc_z = np.where(on_cap, half_height, arc_z)
    arc_distance = np.sqrt((radial - arc_radial) ** 2 + (z_abs - arc_z) ** 2)
    cap_radial = np.minimum(radial, radius)
    cap_distance = np.sqrt((radial - cap_radial) ** 2 + (z_abs - half_height) ** 2)
    distance = np.minimum(cap_distance, arc_distance)
    profile_radius = center + np.sqrt(np.maximum(barrel_radius * barrel_radius - z_abs * z_abs, 0.0))
    inside = (z_abs <= half_height) & (radial <= profile_radius)
    return np.where(inside, -distance, distance)


def primitive_sdf(shape_type: int, scale, points) -> np.ndarray:
    """`_query_primitive_sdf`: signed distance of points [N,3] to the primitive in its local frame (z-up capsule / cylinder /
    cone, incl. barrel cylinders)."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    s = [float(x) for x in scale]
    t = int(shape_type)
    if t == GeoType.SPHERE:
        return np.linalg.norm(p, axis=1) - s[0]
    if t == GeoType.BOX:
        q = np.abs(p) - np.array(s[:3])
        return np.linalg.norm(np.maximum(q, 0.0), axis=1) + np.minimum(q.max(axis=1), 0.0)
    if t == GeoType.CAPSULE:
        r, hh = s[0], s[1]
        dz = np.where(p[:, 2] > hh, p[:, 2] - hh, np.where(p[:, 2] < -hh, p[:, 2] + hh, 0.0))
        return np.sqrt(p[:, 0] ** 2 + p[:, 1] ** 2 + dz ** 2) - r
    if t == GeoType.CYLINDER:
        if len(s) > 2 and s[2] > 0.0:
            return _barrel_cylinder_z(s[0], s[1], s[2], p)
        dx = np.linalg.norm(p[:, :2], axis=1) - s[0]
        dy = np.abs(p[:, 2]) - s[1]
        return np.minimum(np.maximum(dx, dy), 0.0) + np.sqrt(np.maximum(dx, 0.0) ** 2 + np.maximum(dy, 0.0) ** 2)
    if t == GeoType.ELLIPSOID:
        eps = 1.0e-8
        r = np.maximum(np.abs(np.array(s[:3])), eps)
        k0 = np.linalg.norm(p / r, axis=1)
        k1 = np.linalg.norm(p / (r * r), axis=1)
        return np.where(k1 > eps, k0 * (k0 - 1.0) / np.maximum(k1, eps), -r.min())
    if t == GeoType.CONE:
        return _capped_cone_z(s[0], 0.0, s[1], p)
    raise NotImplementedError(f"no analytic SDF for shape type {shape_type}")


def primitive_extents(shape_type: int, scale):
    """`get_primitive_extents` (sdf_utils.py:852-888)."""
    s = [float(x) for x in scale]
    t = int(shape_type)
    if t == GeoType.SPHERE:
        e = [s[0]] * 3
    elif t in (GeoType.BOX, GeoType.ELLIPSOID):
        e = s[:3]
    elif t == GeoType.CAPSULE:
        e = [s[0], s[0], s[1] + s[0]]
    elif t == GeoType.CYLINDER and len(s) > 2 and s[2] > 0.0:  # barrel: the side arc bulges past the end radius
        r = s[0] + s[2] - (s[2] ** 2 - s[1] ** 2) ** 0.5
        e = [r, r, s[1]]
    elif t in (GeoType.CYLINDER, GeoType.CONE):
        e = [s[0], s[0], s[1]]
    else:
        raise NotImplementedError(f"Extents not implemented for shape type: {shape_type}")
    return -np.asarray(e, dtype=np.float64), np.asarray(e, dtype=np.float64)


# ------------------------------------------------------------------------------------------------
# mesh source: exact distance + generalized winding number
# ------------------------------------------------------------------------------------------------
def _point_triangle_distance_sq(p, a, b, c):
    """Squared distance of points p [N,3] to triangles (a, b, c) [T,3] -> [N,T] (Ericson's region test, vectorised)."""
    ab, ac = b - a, c - a                                    # [T,3]
    ap = p[:, None, :] - a[None]                             # [N,T,3]
    d1 = np.einsum("tk,ntk->nt", ab, ap)
    d2 = np.einsum("tk,ntk->nt", ac, ap)
    bp = p[:, None, :] - b[None]
    d3 = np.einsum("tk,ntk->nt", ab, bp)
    d4 = np.einsum("tk,ntk->nt", ac, bp)
    cp = p[:, None, :] - c[None]
    d5 = np.einsum("tk,ntk->nt", ab, cp)
    d6 = np.einsum("tk,ntk->nt", ac, cp)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        denom = va + vb + vc
        v = np.where(denom != 0.0, vb / denom, 0.0)
        w = np.where(denom != 0.0, vc / denom, 0.0)
        t_ab = np.where(d1 - d3 != 0.0, d1 / (d1 - d3), 0.0)
        t_ac = np.where(d2 - d6 != 0.0, d2 / (d2 - d6), 0.0)
        t_bc = np.where((d4 - d3) + (d5 - d6) != 0.0, (d4 - d3) / ((d4 - d3) + (d5 - d6)), 0.0)
    # barycentric coordinates (u, v, w) of the closest point, region by region (later masks override earlier ones)
    bv, bw = v, w
    m = (va <= 0.0) & ((d4 - d3) >= 0.0) & ((d5 - d6) >= 0.0)
    bv, bw = np.where(m, 1.0 - t_bc, bv), np.where(m, t_bc, bw)
    m = (vb <= 0.0) & (d2 >= 0.0) & (d6 <= 0.0)
    bv, bw = np.where(m, 0.0, bv), np.where(m, t_ac, bw)
    m = (vc <= 0.0) & (d1 >= 0.0) & (d3 <= 0.0)
    bv, bw = np.where(m, t_ab, bv), np.where(m, 0.0, bw)
    m = (d6 >= 0.0) & (d5 <= d6)
    bv, bw = np.where(m, 0.0, bv), np.where(m, 1.0, bw)
    m = (d3 >= 0.0) & (d4 <= d3)
    bv, bw = np.where(m, 1.0, bv), np.where(m, 0.0, bw)
    m = (d1 <= 0.0) & (d2 <= 0.0)
    bv, bw = np.where(m, 0.0, bv), np.where(m, 0.0, bw)
    closest = a[None] + bv[..., None] * ab[None] + bw[..., None] * ac[None]
    diff = p[:, None, :] - closest
    return np.einsum("ntk,ntk->nt", diff, diff)


def _winding_number(p, a, b, c):
    """Generalized winding number of points [N,3] w.r.t. the triangle soup (van Oosterom & Strackee solid angles)."""
    A, B, C = a[None] - p[:, None, :], b[None] - p[:, None, :], c[None] - p[:, None, :]
    la, lb, lc = np.linalg.norm(A, axis=2), np.linalg.norm(B, axis=2), np.linalg.norm(C, axis=2)
    num = np.einsum("ntk,ntk->nt", A, np.cross(B, C))
    den = la * lb * lc + np.einsum("ntk,ntk->nt", A, B) * lc + np.einsum("ntk,ntk->nt", B, C) * la + \
        np.einsum("ntk,ntk->nt", C, A) * lb
    return np.arctan2(num, den).sum(axis=1) / (2.0 * np.pi)


def mesh_sdf(vertices, indices, points, winding_threshold: float = 0.5, chunk: int = 4096) -> np.ndarray:
    """`get_distance_to_mesh`: unsigned distance to the closest triangle, negative where the winding number exceeds the
    threshold (inside)."""
    v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    tri = np.asarray(indices, dtype=np.int64).reshape(-1, 3)
    a, b, c = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    out = np.empty(len(p))
    step = max(1, chunk * 64 // max(len(tri), 1))
    for s in range(0, len(p), step):
        q = p[s:s + step]
        d = np.sqrt(_point_triangle_distance_sq(q, a, b, c).min(axis=1))
        inside = _winding_number(q, a, b, c) > winding_threshold
        out[s:s + step] = np.where(inside, -d, d)
    return out


# ------------------------------------------------------------------------------------------------
# sparse construction
# ------------------------------------------------------------------------------------------------
def _interp_coarse(bg, bx, by, bz, lx, ly, lz, inv_cps):
    """`_interp_coarse_sdf` :204-245 for arrays of (block, local) indices; bg is [bz, by, bx]-indexed."""
    sz, sy, sx = bg.shape
    fx, fy, fz = bx + lx * inv_cps, by + ly * inv_cps, bz + lz * inv_cps
    x0 = np.clip(np.floor(fx).astype(np.int64), 0, sx - 2)
    y0 = np.clip(np.floor(fy).astype(np.int64), 0, sy - 2)
    z0 = np.clip(np.floor(fz).astype(np.int64), 0, sz - 2)
    tx, ty, tz = np.clip(fx - x0, 0.0, 1.0), np.clip(fy - y0, 0.0, 1.0), np.clip(fz - z0, 0.0, 1.0)
    g = lambda dz, dy, dx: bg[z0 + dz, y0 + dy, x0 + dx]  # noqa: E731
    c00 = g(0, 0, 0) * (1.0 - tx) + g(0, 0, 1) * tx
    c10 = g(0, 1, 0) * (1.0 - tx) + g(0, 1, 1) * tx
    c01 = g(1, 0, 0) * (1.0 - tx) + g(1, 0, 1) * tx
    c11 = g(1, 1, 0) * (1.0 - tx) + g(1, 1, 1) * tx
    c0 = c00 * (1.0 - ty) + c10 * ty
    c1 = c01 * (1.0 - ty) + c11 * ty
    return c0 * (1.0 - tz) + c1 * tz


def build_sparse_sdf(query, grid_size, cell_size, min_corner, max_corner, *, subgrid_size: int = 8,
                     narrow_band_thickness: float = 0.1, quantization_mode: int = QuantizationMode.UINT16,
                     linearization_error_threshold: float | None = None) -> dict:
    """`_build_sparse_sdf` with `query(points[N,3]) -> sdf[N]` as the source.  Returns the reference's sparse-data dict
    (coarse_sdf [d+1,h+1,w+1], subgrid_data [T,T,T], subgrid_start_slots [w,h,d], ...)."""
    cell_size = np.asarray(cell_size, dtype=np.float64)
    min_corner = np.asarray(min_corner, dtype=np.float64)
    max_corner = np.asarray(max_corner, dtype=np.float64)
    ncell = np.asarray(grid_size, dtype=np.int64) - 1
    w, h, d = [int((n + subgrid_size - 1) // subgrid_size) for n in ncell]
    total = w * h * d
    f32 = np.float32
    # the kernels evaluate positions in float32 (wp.vec3): min_corner + float(index) * cell_size
    mc32, cs32 = min_corner.astype(f32), cell_size.astype(f32)

    def pos(ix, iy, iz):  # float32 like the kernels
        return np.stack([mc32[0] + ix.astype(f32) * cs32[0], mc32[1] + iy.astype(f32) * cs32[1],
                         mc32[2] + iz.astype(f32) * cs32[2]], axis=-1)

    # coarse (background) grid at the subgrid corners (build_coarse_sdf_kernel :453-485), x fastest
    bz, by, bx = np.meshgrid(np.arange(d + 1), np.arange(h + 1), np.arange(w + 1), indexing="ij")
    coarse = np.asarray(query(pos((bx * subgrid_size).ravel(), (by * subgrid_size).ravel(), (bz * subgrid_size).ravel())),
                        dtype=f32).reshape(d + 1, h + 1, w + 1)

    # occupancy at the subgrid centres (check_subgrid_occupied_kernel)
    half_subgrid = subgrid_size * 0.5 * cell_size
    subgrid_radius = float(np.linalg.norm(half_subgrid))
    lo_thr, hi_thr = f32(-narrow_band_thickness - subgrid_radius), f32(narrow_band_thickness + subgrid_radius)
    sz, sy, sx = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    sx, sy, sz = sx.ravel(), sy.ravel(), sz.ravel()  # subgrid index = z*w*h + y*w + x
    centre = np.stack([mc32[0] + ((sx * subgrid_size).astype(f32) + f32(subgrid_size) * f32(0.5)) * cs32[0],
                       mc32[1] + ((sy * subgrid_size).astype(f32) + f32(subgrid_size) * f32(0.5)) * cs32[1],
                       mc32[2] + ((sz * subgrid_size).astype(f32) + f32(subgrid_size) * f32(0.5)) * cs32[2]], axis=-1)
    sd = np.asarray(query(centre), dtype=f32)
    required = np.where(np.sign(sd) > 0.0, sd < hi_thr, sd > lo_thr)  # _is_in_narrow_band :248-253

    # samples of the occupied subgrids, (subgrid_size+1)^3 each, local index x fastest
    spd = subgrid_size + 1
    lz, ly, lx = np.meshgrid(np.arange(spd), np.arange(spd), np.arange(spd), indexing="ij")
    lx, ly, lz = lx.ravel(), ly.ravel(), lz.ravel()
    occ = np.flatnonzero(required)
    values = {}
    if len(occ):
        gx = (sx[occ, None] * subgrid_size + lx[None]).ravel()
        gy = (sy[occ, None] * subgrid_size + ly[None]).ravel()
        gz = (sz[occ, None] * subgrid_size + lz[None]).ravel()
        vals = np.asarray(query(pos(gx, gy, gz)), dtype=f32).reshape(len(occ), spd ** 3)
        values = dict(zip(occ.tolist(), vals))

    # linearity demotion (accumulate_subgrid_linearity_error_kernel + _apply_subgrid_linearity_kernel)
    if linearization_error_threshold is None:
        linearization_error_threshold = float(1e-6 * np.linalg.norm(max_corner - min_corner))
    is_linear = np.zeros(total, dtype=bool)
    if linearization_error_threshold > 0.0 and len(occ):
        inv_cps = f32(1.0) / f32(subgrid_size)
        for s in occ:
            cv = _interp_coarse(coarse, f32(sx[s]), f32(sy[s]), f32(sz[s]), lx.astype(f32), ly.astype(f32), lz.astype(f32),
                                inv_cps)
            if float(np.max(np.abs(values[int(s)] - cv.astype(f32)))) < linearization_error_threshold:
                is_linear[s] = True
        required = required & ~is_linear

    addresses = np.cumsum(required) - required  # exclusive scan
    num_required = int(required.sum())
    global_min = -narrow_band_thickness - subgrid_radius
    global_max = narrow_band_thickness + subgrid_radius
    sdf_range = global_max - global_min
    if sdf_range < 1e-10:
        sdf_range = 1.0
    slots = np.full((w, h, d), SLOT_EMPTY, dtype=np.uint32)
    if num_required == 0:
        sub = np.zeros((1, 1, 1), dtype=np.float32)
        tex_size, final_min, final_range = 1, 0.0, 1.0
    else:
        tpd = max(1, int(np.ceil(num_required ** (1.0 / 3.0))))
        while tpd ** 3 < num_required:
            tpd += 1
        tex_size = tpd * spd
        dtype = {QuantizationMode.FLOAT32: np.float32, QuantizationMode.UINT16: np.uint16,
                 QuantizationMode.UINT8: np.uint8}[quantization_mode]
        sub = np.zeros((tex_size, tex_size, tex_size), dtype=dtype)  # [z, y, x]
        inv_range = f32(1.0 / sdf_range)
        for s in np.flatnonzero(required):
            adr = int(addresses[s])
            az = adr // (tpd * tpd)
            ay = (adr - az * tpd * tpd) // tpd
            ax = adr - az * tpd * tpd - ay * tpd
            slots[sx[s], sy[s], sz[s]] = np.uint32(ax | (ay << 10) | (az << 20))
            v = values[int(s)].reshape(spd, spd, spd)
            if quantization_mode == QuantizationMode.FLOAT32:
                block = v
            else:
                vn = np.clip((v - f32(global_min)) * inv_range, f32(0.0), f32(1.0))
                block = (vn * f32(65535.0 if quantization_mode == QuantizationMode.UINT16 else 255.0)).astype(dtype)
            sub[az * spd:(az + 1) * spd, ay * spd:(ay + 1) * spd, ax * spd:(ax + 1) * spd] = block
        final_min, final_range = (0.0, 1.0) if quantization_mode == QuantizationMode.FLOAT32 else (global_min, sdf_range)
    lin = np.flatnonzero(is_linear)
    slots[sx[lin], sy[lin], sz[lin]] = SLOT_LINEAR
    return {
        "coarse_sdf": coarse.astype(np.float32), "subgrid_data": sub, "subgrid_start_slots": slots, "coarse_dims": (w, h, d),
        "subgrid_tex_size": tex_size, "num_subgrids": num_required, "min_extents": min_corner,
        "max_extents": min_corner + np.array([w, h, d], dtype=float) * subgrid_size * cell_size, "cell_size": cell_size,
        "subgrid_size": subgrid_size, "quantization_mode": quantization_mode, "subgrids_min_sdf_value": final_min,
        "subgrids_sdf_value_range": final_range, "subgrid_required": required.astype(np.int32),
    }


# ------------------------------------------------------------------------------------------------
# TextureSDF: the sampled representation (TextureSDFData :126-160) + float32 sampler
# ------------------------------------------------------------------------------------------------
@dataclass
class TextureSDF:
    coarse: np.ndarray          # float32 [cz+1, cy+1, cx+1] ("coarse_texture", z-major)
    subgrid: np.ndarray         # float32 / uint16 / uint8 [T, T, T] ("subgrid_texture"); raw stored values
    slots: np.ndarray           # uint32 [cx, cy, cz] ("subgrid_start_slots")
    box_lower: np.ndarray       # float32 [3]
    box_upper: np.ndarray
    inv_dx: np.ndarray          # float32 [3]
    subgrid_size: int
    voxel_size: np.ndarray
    voxel_radius: float
    min_value: float            # subgrids_min_sdf_value
    value_range: float          # subgrids_sdf_value_range
    scale_baked: bool = False

    @property
    def quantization_mode(self):
        return {np.dtype(np.float32): QuantizationMode.FLOAT32, np.dtype(np.uint16): QuantizationMode.UINT16,
                np.dtype(np.uint8): QuantizationMode.UINT8}[self.subgrid.dtype]

    def _texel_scale(self):
        """Texture reads of integer formats return normalised [0, 1] floats."""
        return {QuantizationMode.FLOAT32: np.float32(1.0), QuantizationMode.UINT16: np.float32(1.0 / 65535.0),
                QuantizationMode.UINT8: np.float32(1.0 / 255.0)}[self.quantization_mode]

    def _locate(self, f):
        """`_locate_cell_coords` + slot lookup for fine-grid coordinates f [N,3] (float32)."""
        f32 = np.float32
        cx, cy, cz = self.slots.shape
        ssf = f32(self.subgrid_size)
        fv = np.array([cx, cy, cz], dtype=f32) * ssf                       # fine_verts
        fc = np.clip(f, f32(0.0), fv)
        ncell = fv.astype(np.int64)
        i = np.clip(np.floor(fc).astype(np.int64), 0, ncell - 1)
        t = fc - i.astype(f32)
        f2c = f32(1.0 / self.subgrid_size)
        base = np.clip((i.astype(f32) * f2c).astype(np.int64), 0, np.array([cx, cy, cz]) - 1)
        slot = self.slots[base[:, 0], base[:, 1], base[:, 2]]
        return i, t, base, slot

    def _corners(self, f):
        """8 corner values [N,8] (v000, v100, v010, v110, v001, v101, v011, v111) + interpolation weights [N,3] + needs_scale."""
        f32 = np.float32
        i, t, base, slot = self._locate(f)
        n = len(f)
        corners = np.zeros((n, 8), dtype=f32)
        coarse_cell = slot >= SLOT_LINEAR
        t = t.copy()
        if coarse_cell.any():
            k = np.flatnonzero(coarse_cell)
            cf = (i[k].astype(f32) + t[k]) * f32(1.0 / self.subgrid_size)
            t[k] = cf - base[k].astype(f32)
            bx, by, bz = base[k, 0], base[k, 1], base[k, 2]
            for c, (dx, dy, dz) in enumerate(((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1))):
                corners[k, c] = self.coarse[bz + dz, by + dy, bx + dx]
        fine = ~coarse_cell
        if fine.any():
            k = np.flatnonzero(fine)
            s = slot[k].astype(np.int64)
            blk = np.stack([s & 0x3FF, (s >> 10) & 0x3FF, (s >> 20) & 0x3FF], axis=1)
            spd = self.subgrid_size + 1
            o = blk * spd + (i[k] - base[k] * self.subgrid_size)
            scale = self._texel_scale()
            for c, (dx, dy, dz) in enumerate(((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1))):
                corners[k, c] = self.subgrid[o[:, 2] + dz, o[:, 1] + dy, o[:, 0] + dx].astype(f32) * scale
        return corners, t, fine

    def sample(self, points) -> np.ndarray:
        """`texture_sample_sdf`: float32 signed distance at local points [N,3]."""
        f32 = np.float32
        p = np.asarray(points, dtype=f32).reshape(-1, 3)
        clamped = np.minimum(np.maximum(p, self.box_lower), self.box_upper)
        diff = p - clamped
        diff_sq = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
        f = (clamped - self.box_lower) * self.inv_dx
        v, t, fine = self._corners(f)
        tx, ty, tz = t[:, 0], t[:, 1], t[:, 2]
        c00 = v[:, 0] + (v[:, 1] - v[:, 0]) * tx
        c10 = v[:, 2] + (v[:, 3] - v[:, 2]) * tx
        c01 = v[:, 4] + (v[:, 5] - v[:, 4]) * tx
        c11 = v[:, 6] + (v[:, 7] - v[:, 6]) * tx
        c0 = c00 + (c10 - c00) * ty
        c1 = c01 + (c11 - c01) * ty
        val = c0 + (c1 - c0) * tz
        val = np.where(fine, val * f32(self.value_range) + f32(self.min_value), val)
        return (val + np.sqrt(diff_sq)).astype(f32)

    def sample_grad(self, points):
        """`texture_sample_sdf_grad`: (distance [N], gradient [N,3]); de-quantises the corners before blending like
        `_read_cell_corners` :854-933."""
        f32 = np.float32
        p = np.asarray(points, dtype=f32).reshape(-1, 3)
        clamped = np.minimum(np.maximum(p, self.box_lower), self.box_upper)
        diff = p - clamped
        diff_mag = np.sqrt((diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2])
        f = (clamped - self.box_lower) * self.inv_dx
        v, t, fine = self._corners(f)
        v = np.where(fine[:, None], v * f32(self.value_range) + f32(self.min_value), v)
        tx, ty, tz = t[:, 0], t[:, 1], t[:, 2]
        c00 = v[:, 0] + (v[:, 1] - v[:, 0]) * tx
        c10 = v[:, 2] + (v[:, 3] - v[:, 2]) * tx
        c01 = v[:, 4] + (v[:, 5] - v[:, 4]) * tx
        c11 = v[:, 6] + (v[:, 7] - v[:, 6]) * tx
        c0 = c00 + (c10 - c00) * ty
        c1 = c01 + (c11 - c01) * ty
        val = c0 + (c1 - c0) * tz
        ox, oy, oz = f32(1.0) - tx, f32(1.0) - ty, f32(1.0) - tz
        gx = oy * oz * (v[:, 1] - v[:, 0]) + ty * oz * (v[:, 3] - v[:, 2]) + oy * tz * (v[:, 5] - v[:, 4]) + ty * tz * (v[:, 7] - v[:, 6])
        gy = ox * oz * (v[:, 2] - v[:, 0]) + tx * oz * (v[:, 3] - v[:, 1]) + ox * tz * (v[:, 6] - v[:, 4]) + tx * tz * (v[:, 7] - v[:, 5])
        gz = ox * oy * (v[:, 4] - v[:, 0]) + tx * oy * (v[:, 5] - v[:, 1]) + ox * ty * (v[:, 6] - v[:, 2]) + tx * ty * (v[:, 7] - v[:, 3])
        grad = np.stack([gx, gy, gz], axis=1) * self.inv_dx
        out = diff_mag > 0.0
        val = np.where(out, val + diff_mag, val)
        with np.errstate(divide="ignore", invalid="ignore"):
            grad = np.where(out[:, None], diff / np.where(out, diff_mag, f32(1.0))[:, None], grad)
        return val.astype(f32), grad.astype(f32)

    def sample_at_voxel(self, ijk) -> np.ndarray:
        """`texture_sample_sdf_at_voxel`: one texel read at integer fine-grid vertices [N,3] (coarse cells interpolate)."""
        f32 = np.float32
        ijk = np.asarray(ijk, dtype=np.int64).reshape(-1, 3)
        cx, cy, cz = self.slots.shape
        base = np.clip((ijk.astype(f32) * f32(1.0 / self.subgrid_size)).astype(np.int64), 0, np.array([cx, cy, cz]) - 1)
        slot = self.slots[base[:, 0], base[:, 1], base[:, 2]]
        out = np.empty(len(ijk), dtype=f32)
        fine = slot < SLOT_LINEAR
        if fine.any():
            k = np.flatnonzero(fine)
            s = slot[k].astype(np.int64)
            blk = np.stack([s & 0x3FF, (s >> 10) & 0x3FF, (s >> 20) & 0x3FF], axis=1)
            o = blk * (self.subgrid_size + 1) + (ijk[k] - base[k] * self.subgrid_size)
            raw = self.subgrid[o[:, 2], o[:, 1], o[:, 0]].astype(f32) * self._texel_scale()
            out[k] = raw * f32(self.value_range) + f32(self.min_value)
        if (~fine).any():
            k = np.flatnonzero(~fine)
            out[k] = self.sample(self.box_lower + ijk[k].astype(f32) * self.voxel_size.astype(f32))
        return out


def texture_sdf_from_sparse(sparse: dict, scale_baked: bool = False) -> TextureSDF:
    """`create_sparse_sdf_textures` :2269-2334 (scalar layout: the x-paired texel packing is a CUDA fetch optimisation)."""
    cs = np.asarray(sparse["cell_size"], dtype=np.float64)
    return TextureSDF(
        coarse=np.ascontiguousarray(sparse["coarse_sdf"], dtype=np.float32), subgrid=np.ascontiguousarray(sparse["subgrid_data"]),
        slots=np.ascontiguousarray(sparse["subgrid_start_slots"], dtype=np.uint32),
        box_lower=np.asarray(sparse["min_extents"], dtype=np.float32), box_upper=np.asarray(sparse["max_extents"], dtype=np.float32),
        inv_dx=(1.0 / cs).astype(np.float32), subgrid_size=int(sparse["subgrid_size"]), voxel_size=cs.astype(np.float32),
        voxel_radius=float(0.5 * np.linalg.norm(cs)), min_value=float(sparse["subgrids_min_sdf_value"]),
        value_range=float(sparse["subgrids_sdf_value_range"]), scale_baked=scale_baked)


def _create_from_source(query, min_ext, max_ext, *, narrow_band_range, max_resolution, target_voxel_size, subgrid_size,
                        quantization_mode, scale_baked, return_sparse_data=False):
    """`_create_texture_sdf_from_source` :2337-2398."""
    min_ext, max_ext = np.asarray(min_ext, dtype=np.float64), np.asarray(max_ext, dtype=np.float64)
    ext = max_ext - min_ext
    longest = float(np.max(ext))
    if longest < 1e-10:
        return (None, None) if return_sparse_data else None
    if target_voxel_size is not None:
        if target_voxel_size <= 0.0:
            raise ValueError("target_voxel_size must be > 0")
        derived = int(np.ceil(longest / float(target_voxel_size)))
        max_resolution = max(8, ((derived + 7) // 8) * 8)
    elif max_resolution is None:
        max_resolution = 64
    max_resolution = int(max_resolution)
    if max_resolution <= 0:
        raise ValueError("max_resolution must be > 0")
    if max_resolution >= (1 << 16):
        raise ValueError(f"max_resolution must be less than {1 << 16}")
    cell = longest / max_resolution
    dims = np.ceil(ext / cell).astype(int) + 1
    cell_size = ext / (dims - 1)
    nb = max(abs(narrow_band_range[0]), abs(narrow_band_range[1]))
    sparse = build_sparse_sdf(query, dims, cell_size, min_ext, max_ext, subgrid_size=subgrid_size, narrow_band_thickness=nb,
                              quantization_mode=quantization_mode)
    sdf = texture_sdf_from_sparse(sparse, scale_baked)
    return (sdf, sparse) if return_sparse_data else sdf


def create_texture_sdf_from_primitive(shape_type, shape_scale, *, margin: float = 0.05, narrow_band_range=(-0.1, 0.1),
                                      max_resolution=None, target_voxel_size=None, subgrid_size: int = 8,
                                      quantization_mode: int = QuantizationMode.UINT16, scale_baked: bool = False,
                                      return_sparse_data: bool = False):
    """`create_texture_sdf_from_primitive` :2492-2551."""
    scale = [float(s) for s in shape_scale]
    if len(scale) != 3 or not np.all(np.isfinite(scale)):
        raise ValueError("shape_scale must hold three finite values")
    lo, hi = primitive_extents(shape_type, scale)
    return _create_from_source(lambda p: primitive_sdf(shape_type, scale, p), lo - margin, hi + margin,
                               narrow_band_range=narrow_band_range, max_resolution=max_resolution,
                               target_voxel_size=target_voxel_size, subgrid_size=subgrid_size,
                               quantization_mode=quantization_mode, scale_baked=scale_baked, return_sparse_data=return_sparse_data)


def create_texture_sdf_from_mesh(vertices, indices, *, margin: float = 0.05, narrow_band_range=(-0.1, 0.1), max_resolution=None,
                                 target_voxel_size=None, subgrid_size: int = 8, quantization_mode: int = QuantizationMode.UINT16,
                                 winding_threshold: float = 0.5, scale_baked: bool = False, return_sparse_data: bool = False):
    """`create_texture_sdf_from_mesh` :2401-2489 (SIGN_MODE_WINDING)."""
    v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    return _create_from_source(lambda p: mesh_sdf(v, indices, p, winding_threshold), v.min(axis=0) - margin, v.max(axis=0) + margin,
                               narrow_band_range=narrow_band_range, max_resolution=max_resolution,
                               target_voxel_size=target_voxel_size, subgrid_size=subgrid_size,
                               quantization_mode=quantization_mode, scale_baked=scale_baked, return_sparse_data=return_sparse_data)


def voxel_resolution_from_aabb(aabb_lower, aabb_upper, voxel_budget: int = 100):
    """Model._shape_voxel_resolution of one shape (builder.py:11544-11570): a near-cubic voxel grid over the shape-local AABB
    with at most `voxel_budget` cells (NUM_VOXEL_DEPTH_SLOTS of the contact reduction)."""
    size = np.maximum(np.asarray(aabb_upper, np.float64) - np.asarray(aabb_lower, np.float64), 1e-6)
    v = max((size[0] * size[1] * size[2] / voxel_budget) ** (1.0 / 3.0), 1e-6)
    nx, ny, nz = (max(1, round(size[k] / v)) for k in range(3))
    while nx * ny * nz > voxel_budget:
        if nx >= ny and nx >= nz and nx > 1:
            nx -= 1
        elif ny >= nz and ny > 1:
            ny -= 1
        elif nz > 1:
            nz -= 1
        else:
            break
    return int(nx), int(ny), int(nz)


def mesh_reduction_tables(meshes, scales):
    """Per-shape tables the contact reduction's voxel slots read (builder.py:11600-11611 for GeoType.MESH):
    shape_collision_aabb_lower / _upper (vertices * scale) and _shape_voxel_resolution.  `meshes`: one vertex array per shape."""
    lo, hi, res = [], [], []
    for verts, scale in zip(meshes, scales):
        verts, scale = np.asarray(verts, np.float64), np.asarray(scale, np.float64)[:3]
        a, b = verts.min(axis=0) * scale, verts.max(axis=0) * scale
        a, b = np.minimum(a, b), np.maximum(a, b)
        lo.append(a)
        hi.append(b)
        res.append(voxel_resolution_from_aabb(a, b))
    return np.asarray(lo, np.float32), np.asarray(hi, np.float32), np.asarray(res, np.int32)
