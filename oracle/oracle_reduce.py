"""TEST INFRASTRUCTURE ONLY (never imported by newton_amd/ or by bench.py's timed path).

CPU restatement, in float32 / uint64 numpy scalars, of what the reference's global contact reduction keeps of an unreduced
mesh-SDF contact list with deterministic packing:

  * slot assignment and ranking      contact_reduction_global.py:1519-1752 (export_and_reduce_contact_centered_two_spatial_depths)
  * value packing                    :447-558 (_make_contact_value_det, _make_spatial_contact_value_det), contact_reduction.py:99-107
  * normal bins, face frames, dirs   contact_reduction.py:203-430 (get_face_normal, get_slot, project_point_to_plane,
                                     get_spatial_direction_2d), voxel index :432-466
  * buffered normal                  :631-683 (encode_oct / decode_oct: the exported normal is the decoded one)
  * export                           :2098-2290 (_roundoff_duplicate_bit_for_slot_pair, export_reduced_contacts_kernel,
                                     exported_flags: a contact leaves once), :141-176 (numerical equivalence)

The reference runs this with atomics over a hashtable; every slot ends at the maximum of the packed values offered to it, so
the outcome does not depend on the arrival order (the provisional-winner / rollback machinery only exists for the races).  The
restatement therefore takes the maximum per slot directly, with the fingerprint in the place of the contact id (fingerprints are
unique within a shape pair).  Pinned by tests/golden/reduce_reference_vectors.npz, the record of the reference's own functions
executed on five contact sets in two arrival orders (tests/golden/make_reduce_reference_vectors.py).
"""
import numpy as np

f32 = np.float32
NUM_NORMAL_BINS, NUM_SPATIAL_DIRECTIONS, NUM_VOXEL_DEPTH_SLOTS = 20, 6, 100
VALUES_PER_KEY = NUM_SPATIAL_DIRECTIONS + 1
SCORE_SHIFT, FINGERPRINT_MASK = 10, (1 << 22) - 1
BETA_THRESHOLD = 0.0001  # contact_reduction_global.py:89
NUM_ENTRIES = NUM_NORMAL_BINS + (NUM_VOXEL_DEPTH_SLOTS + VALUES_PER_KEY - 1) // VALUES_PER_KEY  # 20 normal bins + 15 voxel groups

FACE_NORMALS = np.array([  # contact_reduction.py:170-191
    0.49112338, 0.79465455, 0.35682216, -0.18759243, 0.79465450, 0.57735026, -0.60706190, 0.79465450, 0.0,
    -0.18759237, 0.79465450, -0.57735026, 0.49112340, 0.79465455, -0.35682210, 0.98224690, -0.18759257, 0.0,
    0.79465440, 0.18759239, -0.57735030, 0.30353096, -0.18759252, 0.93417233, 0.79465440, 0.18759243, 0.57735030,
    -0.79465450, -0.18759249, 0.57735030, -0.30353105, 0.18759243, 0.93417240, -0.79465440, -0.18759240, -0.57735030,
    -0.98224690, 0.18759254, 0.0, 0.30353096, -0.18759250, -0.93417233, -0.30353084, 0.18759246, -0.93417240,
    0.18759249, -0.79465440, 0.57735026, -0.49112338, -0.79465450, 0.35682213, -0.49112338, -0.79465455, -0.35682213,
    0.18759243, -0.79465440, -0.57735026, 0.60706200, -0.79465440, 0.0], dtype=np.float32).reshape(20, 3)


def _dot3(a, b):  # wp.dot: left to right
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def spatial_directions():
    """get_spatial_direction_2d for the six indices: cos / sin of float32(i) * float32(2 pi / 6) in float32."""
    step = f32(f32(2.0) * f32(np.pi)) / f32(NUM_SPATIAL_DIRECTIONS)
    return np.array([[np.cos(f32(f32(i) * step), dtype=np.float32), np.sin(f32(f32(i) * step), dtype=np.float32)]
                     for i in range(NUM_SPATIAL_DIRECTIONS)], dtype=np.float32)


SPATIAL_DIRS = spatial_directions()


def get_slot(n):
    up = n[1]
    if up > f32(0.65):
        rng = range(0, 5)
    elif up < f32(-0.65):
        rng = range(15, 20)
    elif up >= f32(0.0):
        rng = range(0, 15)
    else:
        rng = range(5, 20)
    best, best_dot = rng[0], _dot3(n, FACE_NORMALS[rng[0]])
    for i in rng[1:]:
        d = _dot3(n, FACE_NORMALS[i])
        if d > best_dot:
            best_dot, best = d, i
    return best


def face_frame(bin_id):
    fn = FACE_NORMALS[bin_id]
    ref = np.array([0, 1, 0], np.float32) if abs(fn[1]) < f32(0.9) else np.array([1, 0, 0], np.float32)
    d = _dot3(ref, fn)
    u = np.array([f32(ref[k] - f32(d * fn[k])) for k in range(3)], np.float32)
    ln = np.sqrt(_dot3(u, u), dtype=np.float32)
    u = np.array([f32(u[k] / ln) for k in range(3)], np.float32)  # wp.normalize: v / length
    v = np.array([f32(f32(fn[1] * u[2]) - f32(fn[2] * u[1])), f32(f32(fn[2] * u[0]) - f32(fn[0] * u[2])),
                  f32(f32(fn[0] * u[1]) - f32(fn[1] * u[0]))], np.float32)
    return u, v


FACE_FRAMES = [face_frame(b) for b in range(NUM_NORMAL_BINS)]


def float_flip(x):
    i = int(f32(x).view(np.uint32))
    return (i ^ ((0xFFFFFFFF if i >> 31 else 0) | 0x80000000)) & 0xFFFFFFFF


def value_depth(score, fp):  # _make_contact_value_det without the contact id
    return ((float_flip(score) >> SCORE_SHIFT) << 22) | (fp & FINGERPRINT_MASK)


def value_spatial(score, inner, fp):  # _make_spatial_contact_value_det without the contact id
    return (int(bool(inner)) << 43) | ((float_flip(score) >> (SCORE_SHIFT + 1)) << 22) | (fp & FINGERPRINT_MASK)


def voxel_index(p, lo, hi, res):
    rel = [f32(0.0)] * 3
    for k in range(3):
        size = f32(hi[k] - lo[k])
        if size > f32(1e-6):
            rel[k] = f32(f32(p[k] - lo[k]) / size)
    v = [min(max(int(f32(rel[k] * f32(res[k]))), 0), int(res[k]) - 1) for k in range(3)]
    return v[0] + v[1] * int(res[0]) + v[2] * int(res[0]) * int(res[1])


def encode_oct(n):
    l1 = f32(f32(abs(n[0]) + abs(n[1])) + abs(n[2]))
    if l1 < f32(1.0e-20):
        return np.zeros(2, np.float32)
    inv = f32(f32(1.0) / l1)
    ox, oy, oz = f32(n[0] * inv), f32(n[1] * inv), f32(n[2] * inv)
    if oz < 0:
        sx, sy = (f32(-1.0) if ox < 0 else f32(1.0)), (f32(-1.0) if oy < 0 else f32(1.0))
        ox, oy = f32(f32(f32(1.0) - abs(oy)) * sx), f32(f32(f32(1.0) - abs(ox)) * sy)
    return np.array([ox, oy], np.float32)


def decode_oct(e):
    nz = f32(f32(f32(1.0) - abs(e[0])) - abs(e[1]))
    nx, ny = e[0], e[1]
    if nz < 0:
        sx, sy = (f32(-1.0) if nx < 0 else f32(1.0)), (f32(-1.0) if ny < 0 else f32(1.0))
        nx, ny = f32(f32(f32(1.0) - abs(ny)) * sx), f32(f32(f32(1.0) - abs(nx)) * sy)
    v = np.array([nx, ny, nz], np.float32)
    ln = np.sqrt(_dot3(v, v), dtype=np.float32)
    return np.array([f32(v[k] / ln) for k in range(3)], np.float32) if ln > 0 else np.zeros(3, np.float32)


def _near_ulps(a, b):
    if abs(f32(a - b)) > f32(1.0e-8):
        return False
    return abs(float_flip(a) - float_flip(b)) <= 16


def reduce_contacts(c):
    """c: dict of arrays as tests/golden/reduce_cases.pack builds it.  -> dict(pair, fp, pos, normal, depth, index) of the
    surviving contacts sorted by (shape a, shape b, fingerprint); `index` points into the input list."""
    n = len(c["fp"])
    table = {}  # (shape a, shape b, entry) -> [value] * 7
    by_fp = {}
    for i in range(n):
        depth, inner_d, outer_d = f32(c["depth"][i]), f32(c["inner"][i]), f32(c["outer"][i])
        if not depth < outer_d:
            continue
        use_inner = bool(depth < inner_d)
        pair, fp = (int(c["pair"][i][0]), int(c["pair"][i][1])), int(c["fp"][i])
        by_fp[(pair, fp)] = i
        nrm = c["normal"][i].astype(np.float32)
        b = get_slot(nrm)
        u, v = FACE_FRAMES[b]
        p2 = (_dot3(c["centered"][i], u), _dot3(c["centered"][i], v))
        slots = table.setdefault((pair, b), [0] * VALUES_PER_KEY)
        for d in range(NUM_SPATIAL_DIRECTIONS):
            score = f32(f32(p2[0] * SPATIAL_DIRS[d][0]) + f32(p2[1] * SPATIAL_DIRS[d][1]))
            slots[d] = max(slots[d], value_spatial(score, use_inner, fp))
        if use_inner:
            slots[NUM_SPATIAL_DIRECTIONS] = max(slots[NUM_SPATIAL_DIRECTIONS], value_depth(f32(-depth), fp))
            vox = min(max(voxel_index(c["local"][i], c["aabb_lo"][i], c["aabb_hi"][i], c["res"][i]), 0), NUM_VOXEL_DEPTH_SLOTS - 1)
            vs = table.setdefault((pair, NUM_NORMAL_BINS + vox // VALUES_PER_KEY), [0] * VALUES_PER_KEY)
            vs[vox % VALUES_PER_KEY] = max(vs[vox % VALUES_PER_KEY], value_depth(f32(-depth), fp))
    return _export(c, table, by_fp)


def _export(c, table, by_fp):
    """export_reduced_contacts_kernel (:2098-2290): roundoff twins inside an entry, every surviving contact once, sorted by
    (shape a, shape b, fingerprint); the exported normal is the decoded octahedral code."""
    # what the buffer holds of a contact: position, depth, the octahedral code of the normal
    oct_code = {k: encode_oct(c["normal"][i].astype(np.float32)) for k, i in by_fp.items()}
    keep = set()
    for (pair, _entry), slots in table.items():
        fps = [(s & FINGERPRINT_MASK) if s else None for s in slots]
        suppressed = 0
        for sb in range(1, VALUES_PER_KEY):
            for sa in range(sb):
                if fps[sa] is None or fps[sb] is None or fps[sa] == fps[sb]:
                    continue
                ia, ib = by_fp[(pair, fps[sa])], by_fp[(pair, fps[sb])]
                same = all(_near_ulps(c["pos"][ia][k], c["pos"][ib][k]) for k in range(3)) and \
                    _near_ulps(c["depth"][ia], c["depth"][ib]) and \
                    all(_near_ulps(oct_code[(pair, fps[sa])][k], oct_code[(pair, fps[sb])][k]) for k in range(2))
                if same:
                    suppressed |= (1 << sa) if fps[sb] < fps[sa] else (1 << sb)
        for s in range(VALUES_PER_KEY):
            if fps[s] is not None and not suppressed & (1 << s):
                keep.add((pair, fps[s]))
    keys = sorted(keep)
    idx = np.array([by_fp[k] for k in keys], np.int64)
    return dict(pair=np.array([k[0] for k in keys], np.int32).reshape(-1, 2), fp=np.array([k[1] for k in keys], np.int32),
                pos=c["pos"][idx].reshape(-1, 3), depth=c["depth"][idx],
                normal=np.array([decode_oct(oct_code[k]) for k in keys], np.float32).reshape(-1, 3), index=idx)


def reduce_buffered_contacts(c):
    """reduce_contact_in_hashtable (:1246-1346) over a buffered contact list -- the variant behind write_contact_to_reducer
    (mesh-plane, mesh / heightfield triangles), NOT the centred two-depth one above.  c: pair, pos, normal, depth, fp and, per
    contact, shape a's world transform `xform_a` [7] with its local AABB / voxel resolution (aabb_lo, aabb_hi, res).
      * the normal that picks the bin is the buffered one (encode_oct -> decode_oct), the point is projected uncentred;
      * the six directional slots only take contacts with depth < beta * |aabb diagonal| (beta = 1e-4), value (score, fp);
      * the bin's max-depth slot and the voxel slot (point in shape a's frame) take every contact, value (-depth, fp)."""
    import oracle_mesh_plane as omp  # noqa: PLC0415  (transform helpers in the pinned operand order)

    n = len(c["fp"])
    table, by_fp = {}, {}
    for i in range(n):
        depth = f32(c["depth"][i])
        pair, fp = (int(c["pair"][i][0]), int(c["pair"][i][1])), int(c["fp"][i])
        by_fp[(pair, fp)] = i
        pos = c["pos"][i].astype(np.float32)
        nrm = decode_oct(encode_oct(c["normal"][i].astype(np.float32)))
        lo, hi = c["aabb_lo"][i].astype(np.float32), c["aabb_hi"][i].astype(np.float32)
        b = get_slot(nrm)
        u, v = FACE_FRAMES[b]
        p2 = (_dot3(pos, u), _dot3(pos, v))
        slots = table.setdefault((pair, b), [0] * VALUES_PER_KEY)
        diag = (hi - lo).astype(np.float32)
        if depth < f32(f32(BETA_THRESHOLD) * np.sqrt(_dot3(diag, diag), dtype=np.float32)):
            for d in range(NUM_SPATIAL_DIRECTIONS):
                score = f32(f32(p2[0] * SPATIAL_DIRS[d][0]) + f32(p2[1] * SPATIAL_DIRS[d][1]))
                slots[d] = max(slots[d], value_depth(score, fp))
        slots[NUM_SPATIAL_DIRECTIONS] = max(slots[NUM_SPATIAL_DIRECTIONS], value_depth(f32(-depth), fp))
        local = np.asarray(omp.transform_point(omp.transform_inverse(c["xform_a"][i]), pos), np.float32)
        vox = min(max(voxel_index(local, lo, hi, c["res"][i]), 0), NUM_VOXEL_DEPTH_SLOTS - 1)
        vs = table.setdefault((pair, NUM_NORMAL_BINS + vox // VALUES_PER_KEY), [0] * VALUES_PER_KEY)
        vs[vox % VALUES_PER_KEY] = max(vs[vox % VALUES_PER_KEY], value_depth(f32(-depth), fp))
    return _export(c, table, by_fp)


def _q_rot_inv(q, v):
    """wp.quat_rotate_inv in float32 (oracle/wp_builtins.h order: v*(2w^2-1) - cross(qv, v)*w*2 + qv*dot(qv, v)*2)."""
    q, v = np.asarray(q, np.float32), np.asarray(v, np.float32)
    qv, w = q[:3], q[3]
    c = np.array([f32(f32(qv[1] * v[2]) - f32(qv[2] * v[1])), f32(f32(qv[2] * v[0]) - f32(qv[0] * v[2])),
                  f32(f32(qv[0] * v[1]) - f32(qv[1] * v[0]))], np.float32)
    k = f32(f32(f32(f32(2.0) * w) * w) - f32(1.0))
    d = _dot3(qv, v)
    return np.array([f32(f32(f32(v[i] * k) - f32(f32(c[i] * w) * f32(2.0))) + f32(f32(qv[i] * d) * f32(2.0))) for i in range(3)],
                    np.float32)


def reduce_inputs_from_mesh_sdf_contacts(rows, pairs, shape_transform, shape_data, shape_gap, shape_sdf_index, sdfs, aabb_lo,
                                         aabb_hi, voxel_res):
    """What mesh_sdf_collision_global_reduce_kernel passes to the reducer for every unreduced contact (sdf_contact.py:1889-1947,
    non-speculative: base gap == gap).  `rows`: (pair index, key, point, normal, distance, ...) as oracle_sdf.mesh_sdf_collide or
    the unreduced device kernel return them."""
    X, D = np.asarray(shape_transform, np.float32), np.asarray(shape_data, np.float32)
    out = {k: [] for k in ("pair", "pos", "normal", "depth", "fp", "centered", "inner", "outer", "local", "aabb_lo", "aabb_hi", "res")}
    for row in rows:
        pair_idx, key, pw, nrm, dist = int(row[0]), int(row[1]), np.asarray(row[2], np.float32), np.asarray(row[3], np.float32), f32(row[4])
        s0, s1 = (int(x) for x in np.asarray(pairs).reshape(-1, 2)[pair_idx])
        mode = (key >> 1) & 1
        tri, sd = (s0, s1) if mode == 0 else (s1, s0)
        t = sdfs[int(shape_sdf_index[sd])]
        sdf_scale = np.ones(3, np.float32) if t.scale_baked else D[sd, :3]
        eps = f32(1.0e-10)
        g = np.array([s if abs(s) > eps else (eps if s >= 0.0 else -eps) for s in sdf_scale], np.float32)
        min_scale = f32(np.min(np.abs(g)))
        gap_sum = f32(f32(shape_gap[s0]) + f32(shape_gap[s1]))
        margin_sum = f32(D[tri, 3] + D[sd, 3])
        mid = ((X[tri, :3] + X[sd, :3]) * f32(0.5)).astype(np.float32)
        out["pair"].append((s0, s1))
        out["pos"].append(pw)
        out["normal"].append(nrm)
        out["depth"].append(dist)
        out["fp"].append(key)
        out["centered"].append((pw - mid).astype(np.float32))
        out["inner"].append(f32(margin_sum + min(f32(f32(t.voxel_radius) * min_scale), gap_sum)))
        out["outer"].append(f32(margin_sum + gap_sum))
        out["local"].append(_q_rot_inv(X[tri, 3:], (pw - X[tri, :3]).astype(np.float32)))
        out["aabb_lo"].append(aabb_lo[tri])
        out["aabb_hi"].append(aabb_hi[tri])
        out["res"].append(voxel_res[tri])
    dt = dict(pair=np.int32, fp=np.int32, res=np.int32)
    return {k: np.asarray(v, dt.get(k, np.float32)).reshape(len(rows), -1 if k not in ("depth", "fp", "inner", "outer") else 1)
            .squeeze(-1) if k in ("depth", "fp", "inner", "outer") else np.asarray(v, dt.get(k, np.float32)).reshape(len(rows), -1)
            for k, v in out.items()}
