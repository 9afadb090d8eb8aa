// nt_model_build.hip -- host-side construction of the nt_model descriptor from Newton's flat Model arrays (C ABI:
// nt_model_create / nt_model_get / nt_model_refresh_params / nt_model_destroy / nt_model_last_error).
//
// A Newton binding hands over the arrays `newton.Model` already holds (newton/_src/sim/model.py:808-1364, the rigid subset) as
// HOST pointers; this unit derives the env-uniform topology tables, the incidence lists, the analytic / convex pair partition
// and the env-major SoA parameter tables, and places them in HIP device memory (or host memory, for inspection and tests).
// It is the C restatement of the Python host logic in newton_amd/model.py (EnvTemplate, pack_param_arrays, params_uniform,
// choose_contact_scratch); tests/test_model_build.py holds the two against each other table by table.
// Host code only: no kernels in this unit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/newton_hip.h"

namespace {

thread_local std::string g_error;

struct Fail {
    nt_status code;
    std::string what;
};
[[noreturn]] void unsupported(const std::string& w) { throw Fail{NT_ERR_UNSUPPORTED, w}; }
[[noreturn]] void invalid(const std::string& w) { throw Fail{NT_ERR_INVALID_ARG, w}; }

enum { GEO_PLANE = 1, GEO_HFIELD = 2, GEO_SPHERE = 3, GEO_CAPSULE = 4, GEO_ELLIPSOID = 5, GEO_CYLINDER = 6, GEO_BOX = 7, GEO_MESH = 8, GEO_CONE = 9, GEO_CONVEX_MESH = 10 };
constexpr int SHAPE_HYDROELASTIC = 1 << 4;  // ShapeFlags.HYDROELASTIC
constexpr int BODY_KINEMATIC = 2;

bool analytic_pair(int ta, int tb) {  // narrow_phase.py:642-655: pairs with a closed-form primitive routine
    if (ta > tb) std::swap(ta, tb);
    if (ta == GEO_PLANE) return tb == GEO_SPHERE || tb == GEO_CAPSULE || tb == GEO_ELLIPSOID || tb == GEO_CYLINDER || tb == GEO_BOX;
    if (ta == GEO_SPHERE) return tb == GEO_SPHERE || tb == GEO_CAPSULE || tb == GEO_CYLINDER || tb == GEO_BOX;
    return ta == GEO_CAPSULE && tb == GEO_CAPSULE;
}
bool convex_type(int t) {
    return t == GEO_PLANE || t == GEO_SPHERE || t == GEO_CAPSULE || t == GEO_ELLIPSOID || t == GEO_CYLINDER || t == GEO_BOX ||
           t == GEO_CONE || t == GEO_CONVEX_MESH;
}

}  // namespace

struct nt_model_handle {
    nt_model desc;
    bool on_device;
    // host copies (the tables themselves when !on_device)
    std::vector<std::vector<int32_t>> itab;
    std::vector<std::vector<float>> ftab;
    std::vector<void*> dev;
    std::vector<int64_t> pair_order;
    std::vector<int32_t> sdf_pairs;  // pairs that leave the tiles: [n][2] template shape ids
    std::vector<uint8_t> sdf_kind, sdf_has_edges;

    const int32_t* put(const std::vector<int32_t>& v) {
        itab.push_back(v.empty() ? std::vector<int32_t>(1, 0) : v);
        return (const int32_t*)place(itab.back().data(), itab.back().size() * 4);
    }
    const float* put(const std::vector<float>& v) {
        ftab.push_back(v.empty() ? std::vector<float>(1, 0.0f) : v);
        return (const float*)place(ftab.back().data(), ftab.back().size() * 4);
    }
    const void* place(const void* host, size_t bytes) {
        if (!on_device) return host;
        void* d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess) throw Fail{NT_ERR_LAUNCH, "hipMalloc failed"};
        dev.push_back(d);
        if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) throw Fail{NT_ERR_LAUNCH, "hipMemcpy failed"};
        return d;
    }
    ~nt_model_handle() {
        for (void* d : dev) (void)hipFree(d);
    }
};

namespace {

// [E*n][ncomp] AoS -> [ncomp][n][ES] SoA (environment index fastest)
std::vector<float> soa(const std::vector<float>& aos, int E, int ES, int n, int ncomp) {
    if (n == 0) return std::vector<float>((size_t)ES, 0.0f);
    std::vector<float> out((size_t)ncomp * n * ES, 0.0f);
    for (int e = 0; e < E; ++e)
        for (int s = 0; s < n; ++s)
            for (int c = 0; c < ncomp; ++c) out[((size_t)c * n + s) * ES + e] = aos[((size_t)e * n + s) * ncomp + c];
    return out;
}

struct Params {
    std::vector<float> body, gravity, joint, dof, shape, gshape;
    int uniform;
};

// pack_param_arrays + params_uniform (newton_amd/model.py)
Params pack_params(const nt_newton_model& s, const nt_model& d, const std::vector<int32_t>& gshape_id) {
    const int E = d.env_count, ES = d.env_stride, nb = d.nb, nj = d.nj, nd = d.nd, ns = d.ns;
    Params P;
    std::vector<float> body((size_t)E * nb * NT_BODY_PARAM_FLOATS);
    for (int i = 0; i < E * nb; ++i) {
        float* r = &body[(size_t)i * NT_BODY_PARAM_FLOATS];
        memcpy(r, s.body_com + 3 * (size_t)i, 12);
        r[3] = s.body_inv_mass[i];
        memcpy(r + 4, s.body_inertia + 9 * (size_t)i, 36);
        memcpy(r + 13, s.body_inv_inertia + 9 * (size_t)i, 36);
        r[22] = s.body_mass[i];
    }
    P.body = soa(body, E, ES, nb, NT_BODY_PARAM_FLOATS);
    P.gravity.assign((size_t)3 * ES, 0.0f);
    for (int e = 0; e < E; ++e) {
        int w = nb ? s.body_world[(size_t)e * nb] : 0;
        if (w < 0) w += s.gravity_count;  // gravity[-1] is the global world (model.py:1300-1304)
        if (w < 0 || w >= s.gravity_count) invalid("gravity has no row for a body's world");
        for (int k = 0; k < 3; ++k) P.gravity[(size_t)k * ES + e] = s.gravity[3 * (size_t)w + k];
    }
    std::vector<float> joint((size_t)E * nj * NT_JOINT_PARAM_FLOATS);
    for (int i = 0; i < E * nj; ++i) {
        memcpy(&joint[(size_t)i * 14], s.joint_X_p + 7 * (size_t)i, 28);
        memcpy(&joint[(size_t)i * 14 + 7], s.joint_X_c + 7 * (size_t)i, 28);
    }
    P.joint = soa(joint, E, ES, nj, NT_JOINT_PARAM_FLOATS);
    // joint_armature_effective (solver_featherstone.py:269-282): 1e10 on the dofs of joints whose child body is kinematic
    const int ND = E * nd;
    std::vector<float> armature(s.joint_armature, s.joint_armature + ND);
    for (int j = 0; j < E * nj; ++j) {
        int child = s.joint_child[j];
        if (child >= 0 && (s.body_flags[child] & BODY_KINEMATIC)) {
            int d0 = s.joint_qd_start[j], d1 = (j + 1 < E * nj) ? s.joint_qd_start[j + 1] : ND;
            for (int k = d0; k < d1; ++k) armature[k] = 1.0e10f;
        }
    }
    std::vector<float> dof((size_t)ND * NT_DOF_PARAM_FLOATS);
    for (int i = 0; i < ND; ++i) {
        float* r = &dof[(size_t)i * NT_DOF_PARAM_FLOATS];
        memcpy(r, s.joint_axis + 3 * (size_t)i, 12);
        r[3] = s.joint_limit_lower[i]; r[4] = s.joint_limit_upper[i]; r[5] = s.joint_target_ke[i]; r[6] = s.joint_target_kd[i];
        r[7] = s.joint_limit_ke[i]; r[8] = s.joint_limit_kd[i]; r[9] = armature[i]; r[10] = s.joint_damping[i];
    }
    P.dof = soa(dof, E, ES, nd, NT_DOF_PARAM_FLOATS);
    auto shape_row = [&](int i, float* r) {
        memcpy(r, s.shape_transform + 7 * (size_t)i, 28);
        memcpy(r + 7, s.shape_scale + 3 * (size_t)i, 12);
        r[10] = s.shape_margin[i]; r[11] = s.shape_gap[i]; r[12] = s.shape_material_mu[i];
        r[13] = s.shape_material_mu_torsional[i]; r[14] = s.shape_material_mu_rolling[i]; r[15] = s.shape_material_ke[i];
        r[16] = s.shape_material_kd[i]; r[17] = s.shape_material_kf[i]; r[18] = s.shape_material_ka[i];
        r[19] = s.shape_material_restitution[i];
    };
    std::vector<float> shape((size_t)E * ns * NT_SHAPE_PARAM_FLOATS);
    for (int i = 0; i < E * ns; ++i) shape_row(d.shape_local0 + i, &shape[(size_t)i * NT_SHAPE_PARAM_FLOATS]);
    P.shape = soa(shape, E, ES, ns, NT_SHAPE_PARAM_FLOATS);
    P.gshape.resize(gshape_id.size() * NT_SHAPE_PARAM_FLOATS);
    for (size_t g = 0; g < gshape_id.size(); ++g) shape_row(gshape_id[g], &P.gshape[g * NT_SHAPE_PARAM_FLOATS]);
    // bit-identical columns in every environment?
    auto uniform = [&](const std::vector<float>& a, int rows) {
        for (int r = 0; r < rows; ++r)
            for (int e = 1; e < E; ++e)
                if (memcmp(&a[(size_t)r * ES + e], &a[(size_t)r * ES], 4) != 0) return false;
        return true;
    };
    P.uniform = uniform(P.body, nb ? nb * NT_BODY_PARAM_FLOATS : 0) && uniform(P.joint, nj ? nj * NT_JOINT_PARAM_FLOATS : 0) &&
                uniform(P.dof, nd ? nd * NT_DOF_PARAM_FLOATS : 0) && uniform(P.shape, ns ? ns * NT_SHAPE_PARAM_FLOATS : 0);
    return P;
}

// per-world table with an optional per-world offset removed: identical in every world, or NT_ERR_UNSUPPORTED
std::vector<int32_t> uniform_table(const int32_t* a, int E, int n, int offset_per_world, bool keep_negative, const char* what) {
    std::vector<int32_t> out(n);
    for (int e = 0; e < E; ++e)
        for (int i = 0; i < n; ++i) {
            int v = a[(size_t)e * n + i];
            if (!(keep_negative && v < 0)) v -= e * offset_per_world;
            else v = -1;
            if (e == 0) out[i] = v;
            else if (out[i] != v) unsupported(std::string("heterogeneous worlds: ") + what + " differs between worlds");
        }
    return out;
}

void build(const nt_newton_model& s, nt_model_handle& h) {
    nt_model& d = h.desc;
    memset(&d, 0, sizeof d);
    const int W = s.world_count;
    const int E = W > 0 ? W : 1;
    d.env_count = E;
    d.env_stride = ((E + 63) / 64) * 64;
    // bodies / joints: world-major, none in the global world
    auto check_world_major = [&](const int32_t* world, int count, const char* what) {
        if (count % E) unsupported(std::string("heterogeneous worlds: ") + what + " count is not a multiple of world_count");
        if (W == 0) return;
        const int n = count / E;
        for (int i = 0; i < count; ++i) {
            if (world[i] < 0) unsupported(std::string(what) + "s in the global world (-1) are not supported together with worlds");
            if (world[i] != i / (n ? n : 1)) unsupported(std::string(what) + "s must be ordered world-major");
        }
    };
    check_world_major(s.body_world, s.body_count, "body");
    check_world_major(s.joint_world, s.joint_count, "joint");
    // shapes: one contiguous world-major local block, static globals before and / or after it
    const int NS = s.shape_count;
    std::vector<int32_t> shape_world(NS);
    for (int i = 0; i < NS; ++i) shape_world[i] = W == 0 ? (s.shape_body[i] >= 0 ? 0 : -1) : s.shape_world[i];
    std::vector<int32_t> gshape_id;
    int L0 = -1, nlocal = 0;
    for (int i = 0; i < NS; ++i) {
        if (shape_world[i] >= 0) {
            if (L0 < 0) L0 = i;
            if (i != L0 + nlocal) unsupported("env-local shapes must form one contiguous, world-major block");
            ++nlocal;
        } else {
            if (s.shape_body[i] >= 0) unsupported("global shapes must be static (shape_body == -1)");
            gshape_id.push_back(i);
        }
    }
    if (L0 < 0) L0 = 0;
    if (nlocal % E) unsupported("env-local shapes must form one contiguous, world-major block");
    const int nb = s.body_count / E, nj = s.joint_count / E, ns = nlocal / E, ng = (int)gshape_id.size();
    if (W > 0)
        for (int i = 0; i < nlocal; ++i)
            if (shape_world[L0 + i] != i / (ns ? ns : 1)) unsupported("shapes must be ordered world-major");
    d.nb = nb; d.nj = nj; d.ns = ns; d.ng = ng;
    d.nd = s.joint_dof_count / E; d.nc = s.joint_coord_count / E; d.ntq = s.joint_target_q_count / E;
    d.shape_local0 = L0;

    d.body_flags = h.put(uniform_table(s.body_flags, E, nb, 0, false, "body_flags"));
    std::vector<int32_t> joint_type = uniform_table(s.joint_type, E, nj, 0, false, "joint_type");
    d.joint_type = h.put(joint_type);
    d.joint_enabled = h.put(uniform_table(s.joint_enabled, E, nj, 0, false, "joint_enabled"));
    std::vector<int32_t> joint_parent = uniform_table(s.joint_parent, E, nj, nb, true, "joint_parent");
    std::vector<int32_t> joint_child = uniform_table(s.joint_child, E, nj, nb, false, "joint_child");
    d.joint_parent = h.put(joint_parent);
    d.joint_child = h.put(joint_child);
    d.joint_q_start = h.put(uniform_table(s.joint_q_start, E, nj, d.nc, false, "joint_q_start"));
    std::vector<int32_t> qd_start = uniform_table(s.joint_qd_start, E, nj, d.nd, false, "joint_qd_start");
    d.joint_qd_start = h.put(qd_start);
    d.joint_tq_start = h.put(uniform_table(s.joint_target_q_start, E, nj, d.ntq, false, "joint_target_q_start"));
    {
        std::vector<int32_t> lin(nj), ang(nj);
        for (int e = 0; e < E; ++e)
            for (int j = 0; j < nj; ++j) {
                int l = s.joint_dof_dim[2 * ((size_t)e * nj + j)], a = s.joint_dof_dim[2 * ((size_t)e * nj + j) + 1];
                if (e == 0) { lin[j] = l; ang[j] = a; }
                else if (lin[j] != l || ang[j] != a) unsupported("heterogeneous worlds: joint_dof_dim differs between worlds");
            }
        d.joint_lin_count = h.put(lin);
        d.joint_ang_count = h.put(ang);
    }
    // articulations (SolverFeatherstone): env-local joint ranges, the same in every world, contiguous cover of the joints
    {
        const int A = s.articulation_count;
        std::vector<int32_t> art_start(1, 0);
        d.na = 0; d.max_art_dofs = 0;
        if (A && nj && A % E == 0) {
            const int na = A / E;
            std::vector<int32_t> st = uniform_table(s.articulation_start, E, na, nj, false, "articulations");
            std::vector<int32_t> en = uniform_table(s.articulation_end, E, na, nj, false, "articulations");
            bool contiguous = na > 0 && st[0] == 0 && en[na - 1] == nj;
            for (int k = 1; k < na && contiguous; ++k) contiguous = st[k] == en[k - 1];
            if (contiguous) {
                d.na = na;
                art_start = st;
                art_start.push_back(nj);
                auto dof_edge = [&](int j) { return j < nj ? qd_start[j] : d.nd; };
                for (int k = 0; k < na; ++k) d.max_art_dofs = std::max(d.max_art_dofs, dof_edge(art_start[k + 1]) - dof_edge(art_start[k]));
            }
        }
        d.art_start = h.put(art_start);
    }
    // shape tables: local template + the global shapes behind it
    auto shape_table = [&](const int32_t* a, int offset_per_world, bool keep_negative, const char* what) {
        std::vector<int32_t> t = ns ? uniform_table(a + L0, E, ns, offset_per_world, keep_negative, what) : std::vector<int32_t>();
        for (int g : gshape_id) t.push_back(keep_negative ? -1 : a[g]);
        return t;
    };
    std::vector<int32_t> shape_body = shape_table(s.shape_body, nb, true, "shape_body");
    std::vector<int32_t> shape_type = shape_table(s.shape_type, 0, false, "shape_type");
    d.shape_body = h.put(shape_body);
    {
        // The tiles see a triangle mesh as what compute_shape_aabbs makes of it -- a shape with a pre-computed local AABB
        // (collide.py:421-445, the branch MESH and CONVEX_MESH share): their table carries CONVEX_MESH for it.  A MESH never is a
        // tile PAIR: its pairs go to the SDF / vertex legs below or are refused.
        std::vector<int32_t> tile_type = shape_type;
        for (int32_t& t : tile_type)
            if (t == GEO_MESH || t == GEO_HFIELD) t = GEO_CONVEX_MESH;
        d.shape_type = h.put(tile_type);
    }
    std::vector<int32_t> shape_flags_tab = shape_table(s.shape_flags, 0, false, "shape_flags");
    d.shape_flags = h.put(shape_flags_tab);
    d.shape_group = h.put(shape_table(s.shape_collision_group, 0, false, "shape_collision_group"));
    {
        std::vector<int32_t> none_start(NS, -1), none_count(NS, 0);
        std::vector<int32_t> ms = shape_table(s.shape_mesh_start ? s.shape_mesh_start : none_start.data(), 0, false, "convex hull mesh");
        std::vector<int32_t> mc = shape_table(s.shape_mesh_count ? s.shape_mesh_count : none_count.data(), 0, false, "convex hull mesh");
        d.shape_mesh_start = h.put(ms);
        d.shape_mesh_count = h.put(mc);
        std::vector<float> pts(s.mesh_points, s.mesh_points + 3 * (size_t)s.mesh_point_count);
        d.mesh_points = h.put(pts);
        std::vector<float> bounds((size_t)(ns + ng) * 6, 0.0f);
        for (int k = 0; k < ns + ng; ++k) {
            if (mc[k] <= 0) continue;
            for (int c = 0; c < 3; ++c) {
                float lo = pts[3 * (size_t)ms[k] + c], hi = lo;
                for (int v = 1; v < mc[k]; ++v) {
                    float x = pts[3 * (size_t)(ms[k] + v) + c];
                    lo = std::min(lo, x); hi = std::max(hi, x);
                }
                bounds[(size_t)k * 6 + c] = lo; bounds[(size_t)k * 6 + 3 + c] = hi;
            }
        }
        d.shape_mesh_bounds = h.put(bounds);
    }
    d.gshape_id = h.put(gshape_id);

    // candidate pairs of one environment, in Newton's order; global-vs-global pairs (static-static) are dropped
    std::vector<int32_t> pa, pb;
    {
        std::vector<int32_t> grank(NS, -1);
        for (int g = 0; g < ng; ++g) grank[gshape_id[g]] = g;
        auto is_local = [&](int id) { return id >= L0 && id < L0 + E * ns; };
        std::vector<int32_t> la, lb, pw;
        for (int i = 0; i < s.shape_contact_pair_count; ++i) {
            int a = s.shape_contact_pairs[2 * (size_t)i], b = s.shape_contact_pairs[2 * (size_t)i + 1];
            if (a < 0 || a >= NS || b < 0 || b >= NS) invalid("shape_contact_pairs holds an invalid shape id");
            int wa = is_local(a) ? (a - L0) / ns : -1, wb = is_local(b) ? (b - L0) / ns : -1;
            int w = std::max(wa, wb);
            if (w < 0) continue;
            if (wa >= 0 && wb >= 0 && wa != wb) invalid("shape_contact_pairs contains a cross-world pair");
            la.push_back(is_local(a) ? a - L0 - w * ns : ns + grank[a]);
            lb.push_back(is_local(b) ? b - L0 - w * ns : ns + grank[b]);
            pw.push_back(w);
        }
        if (la.size() % E) unsupported("heterogeneous worlds: candidate pair count differs between worlds");
        const int npair = (int)la.size() / E;
        for (size_t i = 0; i < la.size(); ++i) {
            if (pw[i] != (int)(i / (npair ? npair : 1))) unsupported("shape_contact_pairs must be ordered world-major");
            size_t k = i % (npair ? npair : 1);
            if (i >= (size_t)npair && (la[i] != la[k] || lb[i] != lb[k])) unsupported("heterogeneous worlds: candidate pairs differ between worlds");
        }
        la.resize(npair); lb.resize(npair);
        // Pairs that leave the primitive / GJK-MPR path (narrow_phase.py:531-538,618-640): both shapes hydroelastic with SDFs (the
        // SDF-SDF leg when the pipeline enables it), both shapes with a texture SDF and collision edges unless box-box (mesh-SDF
        // edge contacts), a triangle mesh against an INFINITE plane (vertex leg) or against a convex primitive (triangle leg).  Not tile pairs: listed for the pipeline's SDF leg
        // in ascending Newton (shape0, shape1) order, nt_model_sdf_pairs.
        {
            std::vector<int32_t> none_idx(NS, -1), edge_cnt(NS, 0);
            if (s.shape_edge_range)
                for (int i = 0; i < NS; ++i) edge_cnt[i] = s.shape_edge_range[2 * (size_t)i + 1];
            const std::vector<int32_t> sdf_idx = shape_table(s.shape_sdf_index ? s.shape_sdf_index : none_idx.data(), 0, false, "shape SDF index");
            const std::vector<int32_t> edges = shape_table(edge_cnt.data(), 0, false, "shape collision-edge count");
            auto newton_id0 = [&](int l) { return l < ns ? L0 + l : gshape_id[l - ns]; };
            auto has_sdf = [&](int l) { return sdf_idx[l] >= 0 && edges[l] > 0; };
            auto hydro = [&](int l) { return (shape_flags_tab[l] & SHAPE_HYDROELASTIC) != 0 && sdf_idx[l] >= 0; };
            auto infinite_plane = [&](int l) {
                const float* sc = s.shape_scale + 3 * (size_t)newton_id0(l);
                return shape_type[l] == GEO_PLANE && sc[0] == 0.0f && sc[1] == 0.0f;
            };
            // a triangle mesh against a convex primitive (narrow_phase.py:633-638 `shape_pairs_mesh`): the triangle leg, pair kind 3
            auto tri_partner = [&](int l) {
                const int ty = shape_type[l];
                return ty == GEO_SPHERE || ty == GEO_CAPSULE || ty == GEO_ELLIPSOID || ty == GEO_CYLINDER || ty == GEO_BOX || ty == GEO_CONE ||
                       ty == GEO_CONVEX_MESH;
            };
            auto mesh_like = [&](int l) { return shape_type[l] == GEO_MESH || shape_type[l] == GEO_HFIELD; };  // (narrow_phase.py:553-583)
            struct Routed { int id0, id1, a, b, kind, edges; };
            std::vector<Routed> routed;
            std::vector<int32_t> ta, tb;
            std::vector<int64_t> tile_pos;
            for (int p = 0; p < npair; ++p) {
                const int a = la[p], b = lb[p];
                int kind = -1;
                if (hydro(a) && hydro(b)) kind = 1;
                else if (has_sdf(a) && has_sdf(b) && !(shape_type[a] == GEO_BOX && shape_type[b] == GEO_BOX)) kind = 0;
                else if ((infinite_plane(a) && shape_type[b] == GEO_MESH) || (infinite_plane(b) && shape_type[a] == GEO_MESH)) kind = 2;
                else if ((mesh_like(a) && tri_partner(b)) || (mesh_like(b) && tri_partner(a))) kind = 3;
                if (kind < 0) { ta.push_back(a); tb.push_back(b); tile_pos.push_back(p); continue; }
                const int ia = newton_id0(a), ib = newton_id0(b);
                routed.push_back(ia < ib ? Routed{ia, ib, a, b, kind, has_sdf(a) && has_sdf(b)} : Routed{ib, ia, b, a, kind, has_sdf(a) && has_sdf(b)});
            }
            std::stable_sort(routed.begin(), routed.end(), [](const Routed& x, const Routed& y) { return x.id0 != y.id0 ? x.id0 < y.id0 : x.id1 < y.id1; });
            for (const Routed& r : routed) {
                h.sdf_pairs.push_back(r.a); h.sdf_pairs.push_back(r.b);
                h.sdf_kind.push_back((uint8_t)r.kind);
                h.sdf_has_edges.push_back((uint8_t)r.edges);
            }
            la = ta; lb = tb;
            h.pair_order = tile_pos;  // (position of every tile pair in the world's slice; permuted by the partition below)
        }
        const std::vector<int64_t> tile_pos = h.pair_order;
        const int ntile = (int)la.size();
        // The reference writes analytic-primitive contacts in its first narrow-phase kernel and queues every other pair for the
        // GJK/MPR kernel (narrow_phase.py:642-655,1004-1014): device pairs are stored analytic first (stable partition)
        // barrel cylinders (scale.z != 0, builder.py:7050-7089): their plane / sphere pairs have no fixed analytic route
        // (narrow_phase.py:682-686,847) and are stored with the convex pairs; the kernel decides per environment and substep
        std::vector<char> barrel(ns + ng, 0);
        for (int l = 0; l < ns + ng; ++l) {
            if (shape_type[l] != GEO_CYLINDER) continue;
            const int id0 = l < ns ? L0 + l : gshape_id[l - ns];
            barrel[l] = s.shape_scale[3 * (size_t)id0 + 2] != 0.0f;
            for (int w = 1; l < ns && w < d.env_count; ++w)
                if ((s.shape_scale[3 * ((size_t)id0 + (size_t)w * ns) + 2] != 0.0f) != (barrel[l] != 0))
                    unsupported("heterogeneous worlds: a cylinder is a barrel in some worlds and straight in others");
        }
        std::vector<int64_t> order;
        for (int pass = 0; pass < 2; ++pass)
            for (int p = 0; p < ntile; ++p) {
                int ta = shape_type[la[p]], tb = shape_type[lb[p]];
                bool an = analytic_pair(ta, tb) && !barrel[la[p]] && !barrel[lb[p]];
                if (pass == 0 && !an && ((ta == GEO_PLANE && tb == GEO_PLANE) || !(convex_type(ta) && convex_type(tb))))
                    unsupported("a collision pair has no analytic path and is outside the convex (MPR/GJK) scope of this build");
                if (an == (pass == 0)) { order.push_back(tile_pos[p]); pa.push_back(la[p]); pb.push_back(lb[p]); }
                if (pass == 0 && an) d.np_analytic += 1;
            }
        h.pair_order = order;
        d.np = ntile;
        d.cpp = d.np_analytic == d.np ? 4 : 5;
        d.pair_a = h.put(pa);
        d.pair_b = h.put(pb);
    }
    // ordered incidence lists (padded to 2*nj / 2*np entries)
    {
        std::vector<std::vector<int32_t>> bj(nb), bp(nb);
        for (int j = 0; j < nj; ++j) {
            if (joint_parent[j] >= 0) bj[joint_parent[j]].push_back(2 * j);
            if (joint_child[j] >= 0) bj[joint_child[j]].push_back(2 * j + 1);
        }
        for (int p = 0; p < d.np; ++p) {
            int ba = shape_body[pa[p]], bb = shape_body[pb[p]];
            if (ba >= 0) bp[ba].push_back(2 * p);
            if (bb >= 0) bp[bb].push_back(2 * p + 1);
        }
        auto csr = [&](const std::vector<std::vector<int32_t>>& lists, int pad, const int32_t*& start, const int32_t*& list) {
            std::vector<int32_t> st(nb + 1, 0), flat;
            for (int b = 0; b < nb; ++b) {
                st[b + 1] = st[b] + (int)lists[b].size();
                flat.insert(flat.end(), lists[b].begin(), lists[b].end());
            }
            flat.resize(pad, 0);
            start = h.put(st);
            list = h.put(flat);
        };
        csr(bj, 2 * nj, d.body_joint_start, d.body_joint_list);
        csr(bp, 2 * d.np, d.body_pair_start, d.body_pair_list);
    }
    Params P = pack_params(s, d, gshape_id);
    d.body_param = h.put(P.body); d.gravity = h.put(P.gravity); d.joint_param = h.put(P.joint); d.dof_param = h.put(P.dof);
    d.shape_param = h.put(P.shape); d.gshape_param = h.put(P.gshape);
    d.params_uniform = P.uniform;
    // pair-heavy scenes: contact records in HBM when the LDS tile does not fit otherwise (choose_contact_scratch)
    d.contact_scratch_in_hbm = 0;
    if (nt_pick_envs_per_block(&d, 0) == 0) {
        d.contact_scratch_in_hbm = 1;
        if (nt_pick_envs_per_block(&d, 0) == 0) d.contact_scratch_in_hbm = 0;
    }
}

template <typename F>
nt_status guarded(F&& f) {
    try {
        f();
        g_error.clear();
        return NT_OK;
    } catch (const Fail& e) {
        g_error = e.what;
        return e.code;
    } catch (const std::exception& e) {
        g_error = e.what();
        return NT_ERR_INVALID_ARG;
    }
}

}  // namespace

extern "C" {

const char* nt_model_last_error(void) { return g_error.c_str(); }

nt_status nt_model_create(const nt_newton_model* src, int32_t on_device, nt_model_handle** out) {
    if (!src || !out) return NT_ERR_INVALID_ARG;
    *out = nullptr;
    nt_model_handle* h = new nt_model_handle();
    h->on_device = on_device != 0;
    // (the tables are appended while the descriptor is filled: reserve so that host pointers stay valid)
    h->itab.reserve(64);
    h->ftab.reserve(32);
    nt_status rc = guarded([&] { build(*src, *h); });
    if (rc != NT_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return NT_OK;
}

const nt_model* nt_model_get(const nt_model_handle* h) { return h ? &h->desc : nullptr; }

nt_status nt_model_pair_order(const nt_model_handle* h, int64_t* out) {
    if (!h || !out) return NT_ERR_INVALID_ARG;
    for (size_t i = 0; i < h->pair_order.size(); ++i) out[i] = h->pair_order[i];
    return NT_OK;
}

nt_status nt_model_sdf_pairs(const nt_model_handle* h, int32_t* count, int32_t* pairs, uint8_t* kind, uint8_t* has_edges) {
    if (!h || !count) return NT_ERR_INVALID_ARG;
    *count = (int32_t)h->sdf_kind.size();
    for (size_t i = 0; i < h->sdf_kind.size(); ++i) {
        if (pairs) { pairs[2 * i] = h->sdf_pairs[2 * i]; pairs[2 * i + 1] = h->sdf_pairs[2 * i + 1]; }
        if (kind) kind[i] = h->sdf_kind[i];
        if (has_edges) has_edges[i] = h->sdf_has_edges[i];
    }
    return NT_OK;
}

nt_status nt_model_refresh_params(nt_model_handle* h, const nt_newton_model* src) {
    if (!h || !src) return NT_ERR_INVALID_ARG;
    return guarded([&] {
        nt_model& d = h->desc;
        std::vector<int32_t> gshape_id;
        if (h->on_device) {
            gshape_id.resize(d.ng);
            if (d.ng && hipMemcpy(gshape_id.data(), d.gshape_id, (size_t)d.ng * 4, hipMemcpyDeviceToHost) != hipSuccess)
                throw Fail{NT_ERR_LAUNCH, "hipMemcpy failed"};
        } else {
            gshape_id.assign(d.gshape_id, d.gshape_id + d.ng);
        }
        Params P = pack_params(*src, d, gshape_id);
        auto update = [&](const float* dst, const std::vector<float>& v) {
            if (v.empty()) return;
            if (h->on_device) {
                if (hipMemcpy((void*)dst, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) throw Fail{NT_ERR_LAUNCH, "hipMemcpy failed"};
            } else {
                memcpy((void*)dst, v.data(), v.size() * 4);
            }
        };
        update(d.body_param, P.body); update(d.gravity, P.gravity); update(d.joint_param, P.joint); update(d.dof_param, P.dof);
        update(d.shape_param, P.shape); update(d.gshape_param, P.gshape);
        d.params_uniform = P.uniform;
        // the env-uniform FLAG tables a runtime edit may touch (Model.notify_model_changed: body_flags, joint_enabled, shape_flags,
        // shape_collision_group); edits that break their uniformity across worlds answer NT_ERR_UNSUPPORTED like nt_model_create
        const int E = d.env_count;
        auto update_i = [&](const int32_t* dst, const std::vector<int32_t>& v) {
            if (v.empty()) return;
            if (h->on_device) {
                if (hipMemcpy((void*)dst, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) throw Fail{NT_ERR_LAUNCH, "hipMemcpy failed"};
            } else {
                memcpy((void*)dst, v.data(), v.size() * 4);
            }
        };
        auto shape_table = [&](const int32_t* a, const char* what) {
            std::vector<int32_t> t = d.ns ? uniform_table(a + d.shape_local0, E, d.ns, 0, false, what) : std::vector<int32_t>();
            for (int g : gshape_id) t.push_back(a[g]);
            return t;
        };
        if (src->body_flags) update_i(d.body_flags, uniform_table(src->body_flags, E, d.nb, 0, false, "body_flags"));
        if (src->joint_enabled && d.nj) update_i(d.joint_enabled, uniform_table(src->joint_enabled, E, d.nj, 0, false, "joint_enabled"));
        if (src->shape_flags) update_i(d.shape_flags, shape_table(src->shape_flags, "shape_flags"));
        if (src->shape_collision_group) update_i(d.shape_group, shape_table(src->shape_collision_group, "shape_collision_group"));
    });
}

void nt_model_destroy(nt_model_handle* h) { delete h; }

}  // extern "C"
