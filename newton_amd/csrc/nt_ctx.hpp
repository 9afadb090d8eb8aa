// nt_ctx.hpp -- the per-lane context (Ctx: lane -> (environment, slot), typed LDS accessors, body / shape / dof accessors) and the
// HBM <-> LDS staging helpers of the fused gfx950 kernels.  No include guard: nt_kernels.hip includes this file once per arithmetic
// namespace (after nt_layout.hpp), so that Ctx's own arithmetic (w_quad, update_body_derived) follows the namespace's rules.
// The tile code EPB of every kernel template carries the environments per workgroup in its low byte and the
// uniform-parameter flag NT_UNI in bit 8: Ctx<EPB>::N is the count, Ctx<EPB>::UNI the flag.
constexpr int NT_UNI = 256;
template <int EPB>
struct Ctx {
    static constexpr int N = EPB & 255;
    static constexpr bool UNI = (EPB & NT_UNI) != 0;
    const KArgs& a;
    Topo T;
    float* lds;
    float* up;  // UNI: the block-shared parameter copy [L.uni_floats], behind the topology
    LdsLayout L;
    int e, slot, env, nslot;
    int tslot;  // start of the item loop of phases with fewer items than slot-threads.  Identity: dealing consecutive items to
                // DIFFERENT waves (13 bodies x 16 envs on all 8 waves instead of 4) was measured and lost 13 % on the headline
                // (87.4 vs 100.8 M env-steps/s) -- twice the wave-instructions cost more than the second wave per SIMD hides
    int pose_in_off;  // rows of the poses the contact writer converts into (collide.py:166-204: the step's incoming poses): L.bq, or the
                      // snapshot L.xiq while integrate_bodies runs beside the pair phase of a fused rollout
    bool lane_split;    // the XPBD body phases may run on linear / angular lanes (false in the convex kernels: they sit at the register
                        // limit, and the second copy of the body phases costs them more in spills than the idle waves give back)
    bool gworld_ready;  // T.gworld holds the global shapes' world transforms / AABBs (stage_global_world ran, a barrier ago)
    bool lds_records;  // NT_TILE_LDS_RECORDS granted: the contact records of this launch live in L.cr
    bool hbm_out;      // the collide phases write the Contacts buffers in HBM (always, except the non-final substeps of an LDS-record rollout)
    bool aos_records;  // pair-heavy fused rollout: the substep's contact records live in nt_contacts.cr (one line per slot); hbm_out then
                       // only holds for the last substep
    bool big;  // contact records in HBM, manifold polygon scratch per lane (compile-time constant at every construction site)
    int ES;
    bool valid;

    // rows: float rows per env in front of the block-shared topology ints (-1: the XPBD / collide layout)
    NT_DI Ctx(const KArgs& a_, float* lds_, int rows = -1, const bool big_ = false) : a(a_), lds(lds_), big(big_) {
        // (rows >= 0: another solver's layout, no live list; the pair-heavy tile sizes its per-lane polygon scratch by the workgroup
        // the launch code chose: 256 lanes, or NT_BIG_SCENE_LANES_WIDE when they fit)
        L = make_layout(a.m, big_, xpbd_keeps_prestep_state(a.p), UNI, rows < 0, a.tile_opts,
                        big_ && (int)blockDim.x > NT_BIG_SCENE_LANES ? NT_BIG_SCENE_LANES_WIDE : NT_BIG_SCENE_LANES);
        pose_in_off = L.bq.off;
        lds_records = rows < 0 && (a.tile_opts & NT_TILE_LDS_RECORDS) != 0;
        hbm_out = true;
        aos_records = false;
        if (rows < 0) rows = L.rows_per_env;
        e = threadIdx.x % N;
        slot = threadIdx.x / N;
        nslot = a.nslot;
        env = blockIdx.x * N + e;
        ES = a.m.env_stride;
        valid = env < a.m.env_count && slot < nslot;
        tslot = slot;
        const nt_model& m = a.m;
        int* ti = reinterpret_cast<int*>(lds + (size_t)rows * N);
        int o = 0;
        // The 24 env-uniform tables are fetched FIRST -- every thread one element of each, all loads in flight together -- and
        // written to LDS afterwards.  (One copy loop per table compiles to load / s_waitcnt vmcnt(0) / ds_write per table: 24
        // serialised memory round trips, ~25 us per launch -- most of what a per-call collide / step kernel took, 8 % of a
        // 10-substep rollout launch.)  Tables longer than the workgroup finish in a tail loop.
        constexpr int NT_TOPO_TABLES = 23;
        const int32_t* const srcs[NT_TOPO_TABLES] = {
            m.body_flags, m.joint_type, m.joint_enabled, m.joint_parent, m.joint_child, m.joint_q_start, m.joint_qd_start, m.joint_tq_start,
            m.joint_lin_count, m.joint_ang_count, m.shape_body, m.shape_type, m.shape_flags, m.shape_group, m.pair_a, m.pair_b,
            m.body_joint_start, m.body_joint_list, m.body_pair_start, m.body_pair_list, m.shape_mesh_start, m.shape_mesh_count, m.gshape_id};
        const int nsg = m.ns + m.ng;
        const int lens[NT_TOPO_TABLES] = {m.nb, m.nj, m.nj, m.nj, m.nj, m.nj, m.nj, m.nj, m.nj, m.nj, nsg, nsg, nsg, nsg, m.np, m.np,
                                          m.nb + 1, 2 * m.nj /* padded by the host */, m.nb + 1, 2 * m.np /* padded */, nsg, nsg, m.ng};
        const int** const dsts[NT_TOPO_TABLES] = {
            &T.body_flags, &T.joint_type, &T.joint_enabled, &T.joint_parent, &T.joint_child, &T.joint_q_start, &T.joint_qd_start,
            &T.joint_tq_start, &T.joint_lin_count, &T.joint_ang_count, &T.shape_body, &T.shape_type, &T.shape_flags, &T.shape_group,
            &T.pair_a, &T.pair_b, &T.body_joint_start, &T.body_joint_list, &T.body_pair_start, &T.body_pair_list, &T.shape_mesh_start,
            &T.shape_mesh_count, &T.gshape_id};
        int first[NT_TOPO_TABLES];
        const int tid = threadIdx.x;
#pragma unroll
        for (int k = 0; k < NT_TOPO_TABLES; ++k) first[k] = tid < lens[k] ? srcs[k][tid] : 0;
        const int ngf = NT_SHAPE_PARAM_FLOATS * m.ng;
        const float gfirst = tid < ngf ? m.gshape_param[tid] : 0.0f;
#pragma unroll
        for (int k = 0; k < NT_TOPO_TABLES; ++k) {
            if (tid < lens[k]) ti[o + tid] = first[k];
            for (int i = tid + (int)blockDim.x; i < lens[k]; i += blockDim.x) ti[o + i] = srcs[k][i];
            *dsts[k] = ti + o;
            o += lens[k];
        }
        {
            float* g = reinterpret_cast<float*>(ti + o);
            if (tid < ngf) g[tid] = gfirst;
            for (int i = tid + (int)blockDim.x; i < ngf; i += blockDim.x) g[i] = m.gshape_param[i];
            T.gshape = g;
            o += ngf;
        }
        T.gworld = reinterpret_cast<float*>(ti + o);
        o += 13 * m.ng;
        gworld_ready = false;
        lane_split = true;
        T.joint_inc = ti + o;  // (filled by stage_joint_inc, one barrier after the tables above are published)
        o += 2 * m.nj;
        T.hit_count = ti + o;
        T.hit_list = ti + o + 1;
        T.pair_desc = ti + o;  // (the two tables exclude each other)
        if (!m.contact_scratch_in_hbm) {
            for (int i = threadIdx.x; i < m.np; i += blockDim.x) {
                int sa = m.pair_a[i], sb = m.pair_b[i], swapped = 0;
                if (m.shape_type[sa] > m.shape_type[sb]) { const int t_ = sa; sa = sb; sb = t_; swapped = 1; }
                ti[o + 4 * i] = sa; ti[o + 4 * i + 1] = sb;
                ti[o + 4 * i + 2] = m.shape_body[sa];
                ti[o + 4 * i + 3] = (m.shape_body[sb] & 0x3fffffff) | (swapped << 30);
            }
        }
        up = reinterpret_cast<float*>(ti + topo_ints(m));
    }
    // the same lane seen from the other arithmetic namespace (ieee::Ctx <-> fused::Ctx): no staging, every member copied
    template <class OtherCtx>
    NT_DI explicit Ctx(const OtherCtx& o, int /*tag*/)
        : a(o.a), T(o.T), lds(o.lds), up(o.up), L(o.L), e(o.e), slot(o.slot), env(o.env), nslot(o.nslot), tslot(o.tslot),
          pose_in_off(o.pose_in_off), lane_split(o.lane_split), gworld_ready(o.gworld_ready), lds_records(o.lds_records), hbm_out(o.hbm_out), aos_records(o.aos_records), big(o.big),
          ES(o.ES), valid(o.valid) {}
    // LDS element (comp, s) of a slot-major field.  `n` (the slot count of the [comp][n] HBM twin) is not needed here; the
    // argument stays so that every access reads like its global-memory counterpart g(comp, n, s)
    template <int NC>
    NT_DI float& l(Fld<NC> f, int comp, int /*n*/, int s) const { return lds[(f.off + s * NC + comp) * N + e]; }
    // parameter element (fields bp / jp / dp / sp): per-environment row, or the block-shared copy of a uniform tile
    template <int NC>
    NT_DI float pl(Fld<NC> f, int comp, int /*n*/, int s) const {
        if constexpr (UNI) return up[f.off + s * NC + comp];
        else return lds[(f.off + s * NC + comp) * N + e];
    }
    template <int NC>
    NT_DI vec3 plv3(Fld<NC> f, int comp0, int n, int s) const {
        return vec3(pl(f, comp0, n, s), pl(f, comp0 + 1, n, s), pl(f, comp0 + 2, n, s));
    }
    template <int NC>
    NT_DI xform plxf(Fld<NC> f, int comp0, int n, int s) const {
        return xform(plv3(f, comp0, n, s),
                     quat(pl(f, comp0 + 3, n, s), pl(f, comp0 + 4, n, s), pl(f, comp0 + 5, n, s), pl(f, comp0 + 6, n, s)));
    }
    template <int NC>
    NT_DI mat33 plm33(Fld<NC> f, int comp0, int n, int s) const {
        return mat33(pl(f, comp0, n, s), pl(f, comp0 + 1, n, s), pl(f, comp0 + 2, n, s), pl(f, comp0 + 3, n, s),
                     pl(f, comp0 + 4, n, s), pl(f, comp0 + 5, n, s), pl(f, comp0 + 6, n, s), pl(f, comp0 + 7, n, s),
                     pl(f, comp0 + 8, n, s));
    }
    NT_DI size_t g(int comp, int n, int s) const { return (size_t)(comp * n + s) * ES + env; }

    template <int NC>
    NT_DI vec3 lv3(Fld<NC> f, int comp0, int n, int s) const {
        return vec3(l(f, comp0, n, s), l(f, comp0 + 1, n, s), l(f, comp0 + 2, n, s));
    }
    template <int NC>
    NT_DI void st_lv3(Fld<NC> f, int comp0, int n, int s, vec3 v) const {
        l(f, comp0, n, s) = v.x; l(f, comp0 + 1, n, s) = v.y; l(f, comp0 + 2, n, s) = v.z;
    }
    template <int NC>
    NT_DI xform lxf(Fld<NC> f, int comp0, int n, int s) const {
        return xform(lv3(f, comp0, n, s),
                     quat(l(f, comp0 + 3, n, s), l(f, comp0 + 4, n, s), l(f, comp0 + 5, n, s), l(f, comp0 + 6, n, s)));
    }
    template <int NC>
    NT_DI void st_lxf(Fld<NC> f, int n, int s, const xform& t) const {
        l(f, 0, n, s) = t.p.x; l(f, 1, n, s) = t.p.y; l(f, 2, n, s) = t.p.z;
        l(f, 3, n, s) = t.q.x; l(f, 4, n, s) = t.q.y; l(f, 5, n, s) = t.q.z; l(f, 6, n, s) = t.q.w;
    }
    template <int NC>
    NT_DI mat33 lm33(Fld<NC> f, int comp0, int n, int s) const {
        return mat33(l(f, comp0, n, s), l(f, comp0 + 1, n, s), l(f, comp0 + 2, n, s), l(f, comp0 + 3, n, s),
                     l(f, comp0 + 4, n, s), l(f, comp0 + 5, n, s), l(f, comp0 + 6, n, s), l(f, comp0 + 7, n, s),
                     l(f, comp0 + 8, n, s));
    }
    NT_DI vec3 gravity() const { return lv3(L.grav, 0, 1, 0); }
    // component-major [comp][n] arrays at a plain row offset (a solver's own scratch, e.g. SolverFeatherstone's)
    NT_DI float& l(int off, int comp, int n, int s) const { return lds[(off + comp * n + s) * N + e]; }
    NT_DI vec3 lv3(int off, int comp0, int n, int s) const {
        return vec3(l(off, comp0, n, s), l(off, comp0 + 1, n, s), l(off, comp0 + 2, n, s));
    }
    NT_DI void st_lv3(int off, int comp0, int n, int s, vec3 v) const {
        l(off, comp0, n, s) = v.x; l(off, comp0 + 1, n, s) = v.y; l(off, comp0 + 2, n, s) = v.z;
    }
    NT_DI xform lxf(int off, int comp0, int n, int s) const {
        return xform(lv3(off, comp0, n, s),
                     quat(l(off, comp0 + 3, n, s), l(off, comp0 + 4, n, s), l(off, comp0 + 5, n, s), l(off, comp0 + 6, n, s)));
    }
    NT_DI void st_lxf(int off, int n, int s, const xform& t) const {
        l(off, 0, n, s) = t.p.x; l(off, 1, n, s) = t.p.y; l(off, 2, n, s) = t.p.z;
        l(off, 3, n, s) = t.q.x; l(off, 4, n, s) = t.q.y; l(off, 5, n, s) = t.q.z; l(off, 6, n, s) = t.q.w;
    }
    NT_DI vec3 gv3(const float* base, int comp0, int n, int s) const {
        return vec3(base[g(comp0, n, s)], base[g(comp0 + 1, n, s)], base[g(comp0 + 2, n, s)]);
    }

    // inverse of body_joint_list: position of code (j << 1 | side) in the list, -1 for the world side of a root joint.  Reads the
    // block-shared copy of the list, i.e. runs one barrier after the constructor; the XPBD joint phases are barriers later still
    NT_DI void stage_joint_inc() const {
        const int nlist = a.m.nj > 0 ? T.body_joint_start[a.m.nb] : 0;
        int* inv = const_cast<int*>(T.joint_inc);
        for (int i = threadIdx.x; i < 2 * a.m.nj; i += blockDim.x) {
            int at = -1;
            for (int k = 0; k < nlist; ++k)
                if (T.body_joint_list[k] == i) at = k;
            inv[i] = at;
        }
    }
    NT_DI xform body_q(int b) const { return lxf(L.bq, 0, a.m.nb, b); }
    NT_DI xform body_q_in(int b) const { return lxf(Fld<7>{pose_in_off}, 0, a.m.nb, b); }
    NT_DI quat body_rot(int b) const {
        const int nb = a.m.nb;
        return quat(l(L.bq, 3, nb, b), l(L.bq, 4, nb, b), l(L.bq, 5, nb, b), l(L.bq, 6, nb, b));
    }
    NT_DI vec3 body_v(int b) const { return lv3(L.bqd, 0, a.m.nb, b); }
    NT_DI vec3 body_w(int b) const { return lv3(L.bqd, 3, a.m.nb, b); }
    NT_DI float inv_mass(int b) const { return pl(L.bp, BP_INV_MASS, a.m.nb, b); }
    NT_DI mat33 inv_inertia(int b) const { return plm33(L.bp, BP_INV_INERTIA, a.m.nb, b); }
    NT_DI mat33 inertia(int b) const { return plm33(L.bp, BP_INERTIA, a.m.nb, b); }
    NT_DI vec3 com(int b) const { return plv3(L.bp, BP_COM, a.m.nb, b); }
    NT_DI vec3 world_com(int b) const { return lv3(L.bd, 0, a.m.nb, b); }
    // the world-frame inverse inertia tile of body b as a value (six floats fetched once, then any number of quadratic forms)
    struct Wsym {
        float xx, xy, xz, yy, yz, zz;
        NT_DI float quad(vec3 v) const {
            vec3 wv(xx * v.x + xy * v.y + xz * v.z, xy * v.x + yy * v.y + yz * v.z, xz * v.x + yz * v.y + zz * v.z);
            return dot(v, wv);
        }
    };
    NT_DI Wsym w_tile(int b) const {
        const int nb = a.m.nb;
        return Wsym{l(L.bd, 3, nb, b), l(L.bd, 4, nb, b), l(L.bd, 5, nb, b), l(L.bd, 6, nb, b), l(L.bd, 7, nb, b), l(L.bd, 8, nb, b)};
    }
    // a^T (R I^-1 R^T) a for body b (world-frame inverse inertia, symmetric 6-float tile in LDS)
    NT_DI float w_quad(int b, vec3 v) const {
        const int nb = a.m.nb;
        float xx = l(L.bd, 3, nb, b), xy = l(L.bd, 4, nb, b), xz = l(L.bd, 5, nb, b);
        float yy = l(L.bd, 6, nb, b), yz = l(L.bd, 7, nb, b), zz = l(L.bd, 8, nb, b);
        vec3 wv(xx * v.x + xy * v.y + xz * v.z, xy * v.x + yy * v.y + yz * v.z, xz * v.x + yz * v.y + zz * v.z);
        return dot(v, wv);
    }
    NT_DI void update_body_derived(int b) const {
        xform X = body_q(b);
        update_world_com(b, X);
        update_body_w(b, X.q);
    }
    // world COM of body b at pose X (the first three rows of the body-derived tile)
    NT_DI void update_world_com(int b, const xform& X) const { st_lv3(L.bd, 0, a.m.nb, b, xform_point(X, com(b))); }
    // W = R I^-1 R^T of body b at rotation q (rows 3..8 of the tile)
    NT_DI void update_body_w(int b, quat q) const {
        const int nb = a.m.nb;
        mat33 R = quat_to_matrix(q);
        mat33 Ii = inv_inertia(b);
        // T = I^-1 R^T ; W = R T
        vec3 t0 = Ii * vec3(R.m00, R.m01, R.m02), t1 = Ii * vec3(R.m10, R.m11, R.m12), t2 = Ii * vec3(R.m20, R.m21, R.m22);
        vec3 r0(R.m00, R.m01, R.m02), r1(R.m10, R.m11, R.m12), r2(R.m20, R.m21, R.m22);
        l(L.bd, 3, nb, b) = dot(r0, t0); l(L.bd, 4, nb, b) = dot(r0, t1); l(L.bd, 5, nb, b) = dot(r0, t2);
        l(L.bd, 6, nb, b) = dot(r1, t1); l(L.bd, 7, nb, b) = dot(r1, t2); l(L.bd, 8, nb, b) = dot(r2, t2);
    }
    NT_DI float dof(int row, int d) const { return pl(L.dp, row, a.m.nd, d); }
    NT_DI vec3 dof_axis(int d) const { return plv3(L.dp, DP_AXIS, a.m.nd, d); }

    // shape accessors: s < ns local (per-env params in LDS), otherwise the env-uniform global table
    NT_DI float shape_f(int s, int comp) const {
        if (s < a.m.ns) return pl(L.sp, comp, a.m.ns, s);
        return T.gshape[(s - a.m.ns) * NT_SHAPE_PARAM_FLOATS + comp];
    }
    // the same value through ONE load from a selected LDS address (no branch between the two tables: the loads of a phase's operands
    // then leave as one batch)
    NT_DI float shape_f_sel(int s, int comp) const {
        const float* local;
        if constexpr (UNI) local = up + (L.sp.off + s * NC_SP + comp);
        else local = lds + ((L.sp.off + s * NC_SP + comp) * N + e);
        const float* global = T.gshape + ((s - a.m.ns) * NT_SHAPE_PARAM_FLOATS + comp);
        return *(s < a.m.ns ? local : global);
    }
    NT_DI vec3 shape_scale(int s) const { return vec3(shape_f(s, SP_SCALE), shape_f(s, SP_SCALE + 1), shape_f(s, SP_SCALE + 2)); }
    NT_DI xform shape_local_xform(int s) const {
        return xform(vec3(shape_f(s, 0), shape_f(s, 1), shape_f(s, 2)), quat(shape_f(s, 3), shape_f(s, 4), shape_f(s, 5), shape_f(s, 6)));
    }
    NT_DI int newton_shape_id(int s) const {  // flat Newton shape index
        return s < a.m.ns ? a.m.shape_local0 + env * a.m.ns + s : T.gshape_id[s - a.m.ns];
    }
    NT_DI int local_shape_id(int gid) const {
        int rel = gid - a.m.shape_local0 - env * a.m.ns;
        if (rel >= 0 && rel < a.m.ns) return rel;
        int g = 0;
        for (int k = 0; k < a.m.ng; ++k)
            if (T.gshape_id[k] == gid) g = k;
        return a.m.ns + g;
    }
};

// ------------------------------------------------------------------------------------------------
// HBM <-> LDS staging
// ------------------------------------------------------------------------------------------------
// field [ncomp][n][ES] in HBM <-> slot-major rows in LDS
// (loads in batches of eight before the LDS writes: a load / wait / write per row is one memory round trip per row)
constexpr int NT_STAGE_BATCH = 8;
template <int EPB, int NC>
NT_DI void stage_rows(const Ctx<EPB>& c, Fld<NC> f, const float* src, int ncomp, int n) {
    for (int s = c.slot; s < n; s += c.nslot)
        for (int c0 = 0; c0 < ncomp; c0 += NT_STAGE_BATCH) {
            float v[NT_STAGE_BATCH];
#pragma unroll
            for (int k = 0; k < NT_STAGE_BATCH; ++k) v[k] = c0 + k < ncomp ? src[c.g(c0 + k, n, s)] : 0.0f;
#pragma unroll
            for (int k = 0; k < NT_STAGE_BATCH; ++k)
                if (c0 + k < ncomp) c.l(f, c0 + k, n, s) = v[k];
        }
}
template <int EPB, int NC>
NT_DI void unstage_rows(const Ctx<EPB>& c, Fld<NC> f, float* dst, int ncomp, int n) {
    for (int comp = 0; comp < ncomp; ++comp)
        for (int s = c.slot; s < n; s += c.nslot) dst[c.g(comp, n, s)] = c.l(f, comp, n, s);
}
// plain rows (a solver's own [row] arrays)
template <int EPB>
NT_DI void stage_rows(const Ctx<EPB>& c, int lds_off, const float* src, int rows) {
    for (int r0 = c.slot; r0 < rows; r0 += NT_STAGE_BATCH * c.nslot) {
        float v[NT_STAGE_BATCH];
#pragma unroll
        for (int k = 0; k < NT_STAGE_BATCH; ++k) v[k] = r0 + k * c.nslot < rows ? src[(size_t)(r0 + k * c.nslot) * c.ES + c.env] : 0.0f;
#pragma unroll
        for (int k = 0; k < NT_STAGE_BATCH; ++k)
            if (r0 + k * c.nslot < rows) c.lds[(lds_off + r0 + k * c.nslot) * Ctx<EPB>::N + c.e] = v[k];
    }
}
template <int EPB>
NT_DI void unstage_rows(const Ctx<EPB>& c, int lds_off, float* dst, int rows) {
    for (int r = c.slot; r < rows; r += c.nslot) dst[(size_t)r * c.ES + c.env] = c.lds[(lds_off + r) * Ctx<EPB>::N + c.e];
}

template <int EPB>
NT_DI void load_state(const Ctx<EPB>& c, const nt_state& s) {
    if (!c.valid) return;
    stage_rows(c, c.L.bq, s.body_q, 7, c.a.m.nb);
    stage_rows(c, c.L.bqd, s.body_qd, 6, c.a.m.nb);
}
template <int EPB>
NT_DI void store_state(const Ctx<EPB>& c, const nt_state& s) {
    if (!c.valid) return;
    unstage_rows(c, c.L.bq, s.body_q, 7, c.a.m.nb);
    unstage_rows(c, c.L.bqd, s.body_qd, 6, c.a.m.nb);
}
// ---- first-trip batches: the first item a lane owns in a field (s = slot), all components at once.  A kernel's whole tile -- state,
// parameters, controls -- is fetched as ONE batch of loads followed by the LDS writes (load_tile); items beyond the first trip
// (fields longer than the slot-thread count) follow through stage_rows_rest.
template <int NCOMP, int EPB>
NT_DI void first_load(const Ctx<EPB>& c, const float* src, int n, float (&v)[NCOMP]) {
#pragma unroll
    for (int k = 0; k < NCOMP; ++k) v[k] = c.slot < n ? src[c.g(k, n, c.slot)] : 0.0f;
}
template <int NCOMP, int EPB, int NC>
NT_DI void first_store(const Ctx<EPB>& c, Fld<NC> f, int n, const float (&v)[NCOMP]) {
    if (c.slot < n) {
#pragma unroll
        for (int k = 0; k < NCOMP; ++k) c.l(f, k, n, c.slot) = v[k];
    }
}
template <int EPB, int NC>
NT_DI void stage_rows_rest(const Ctx<EPB>& c, Fld<NC> f, const float* src, int ncomp, int n) {
    for (int s = c.slot + c.nslot; s < n; s += c.nslot)
        for (int c0 = 0; c0 < ncomp; c0 += NT_STAGE_BATCH) {
            float v[NT_STAGE_BATCH];
#pragma unroll
            for (int k = 0; k < NT_STAGE_BATCH; ++k) v[k] = c0 + k < ncomp ? src[c.g(c0 + k, n, s)] : 0.0f;
#pragma unroll
            for (int k = 0; k < NT_STAGE_BATCH; ++k)
                if (c0 + k < ncomp) c.l(f, c0 + k, n, s) = v[k];
        }
}

// the block-shared parameter copy of a uniform-parameter tile: one copy per workgroup, read from the tile's first environment (the
// host vouches that all columns are equal)
template <int EPB>
NT_DI void load_uniform_params(const Ctx<EPB>& c) {
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    auto body_value = [&](int comp, int b, size_t col) {
        float v = m.body_param[(size_t)(comp * nb + b) * c.ES + col];
        bool inv_row = comp == BP_INV_MASS || (comp >= BP_INV_INERTIA && comp < BP_INV_INERTIA + 9);
        if (inv_row && (m.body_flags[b] & BODY_KINEMATIC)) v = 0.0f;  // (global copy: the LDS topology is not published yet)
        return v;
    };
    const size_t col = (size_t)blockIdx.x * Ctx<EPB>::N;
    const int nbp = NT_BODY_PARAM_FLOATS * nb, njp = NT_JOINT_PARAM_FLOATS * m.nj, ndp = NT_DOF_PARAM_FLOATS * m.nd,
              nsp = NT_SHAPE_PARAM_FLOATS * m.ns;
    const int r0 = threadIdx.x;
    // first trip of the four tables as one batch of loads, then the rest
    const float vb = r0 < nbp ? body_value(r0 / nb, r0 % nb, col) : 0.0f;
    const float vj = r0 < njp ? m.joint_param[(size_t)r0 * c.ES + col] : 0.0f;
    const float vd = r0 < ndp ? m.dof_param[(size_t)r0 * c.ES + col] : 0.0f;
    const float vs = r0 < nsp ? m.shape_param[(size_t)r0 * c.ES + col] : 0.0f;
    if (r0 < nbp) c.up[c.L.bp.off + (r0 % nb) * NC_BP + r0 / nb] = vb;
    if (r0 < njp) c.up[c.L.jp.off + (r0 % m.nj) * NC_JP + r0 / m.nj] = vj;
    if (r0 < ndp) c.up[c.L.dp.off + (r0 % m.nd) * NC_DP + r0 / m.nd] = vd;
    if (r0 < nsp) c.up[c.L.sp.off + (r0 % m.ns) * NC_SP + r0 / m.ns] = vs;
    for (int r = r0 + blockDim.x; r < nbp; r += blockDim.x) c.up[c.L.bp.off + (r % nb) * NC_BP + r / nb] = body_value(r / nb, r % nb, col);
    for (int r = r0 + blockDim.x; r < njp; r += blockDim.x) c.up[c.L.jp.off + (r % m.nj) * NC_JP + r / m.nj] = m.joint_param[(size_t)r * c.ES + col];
    for (int r = r0 + blockDim.x; r < ndp; r += blockDim.x) c.up[c.L.dp.off + (r % m.nd) * NC_DP + r / m.nd] = m.dof_param[(size_t)r * c.ES + col];
    for (int r = r0 + blockDim.x; r < nsp; r += blockDim.x) c.up[c.L.sp.off + (r % m.ns) * NC_SP + r / m.ns] = m.shape_param[(size_t)r * c.ES + col];
}
// parameters and controls: read once per kernel
template <int EPB>
NT_DI void load_params(const Ctx<EPB>& c, bool with_control) {
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    // body params carry the effective (kinematic => 0) inverse mass / inertia (solver.py:173-187)
    auto body_value = [&](int comp, int b, size_t col) {
        float v = m.body_param[(size_t)(comp * nb + b) * c.ES + col];
        bool inv_row = comp == BP_INV_MASS || (comp >= BP_INV_INERTIA && comp < BP_INV_INERTIA + 9);
        if (inv_row && (m.body_flags[b] & BODY_KINEMATIC)) v = 0.0f;  // (global copy: the LDS topology is not published yet)
        return v;
    };
    if constexpr (Ctx<EPB>::UNI) load_uniform_params(c);
    if (!c.valid) return;
    if constexpr (!Ctx<EPB>::UNI) {
        for (int b = c.slot; b < nb; b += c.nslot)
            for (int c0 = 0; c0 < NT_BODY_PARAM_FLOATS; c0 += NT_STAGE_BATCH) {
                float v[NT_STAGE_BATCH];
#pragma unroll
                for (int k = 0; k < NT_STAGE_BATCH; ++k) v[k] = c0 + k < NT_BODY_PARAM_FLOATS ? body_value(c0 + k, b, c.env) : 0.0f;
#pragma unroll
                for (int k = 0; k < NT_STAGE_BATCH; ++k)
                    if (c0 + k < NT_BODY_PARAM_FLOATS) c.lds[(c.L.bp.off + b * NC_BP + c0 + k) * Ctx<EPB>::N + c.e] = v[k];
            }
        stage_rows(c, c.L.jp, m.joint_param, NT_JOINT_PARAM_FLOATS, m.nj);
        stage_rows(c, c.L.dp, m.dof_param, NT_DOF_PARAM_FLOATS, m.nd);
        stage_rows(c, c.L.sp, m.shape_param, NT_SHAPE_PARAM_FLOATS, m.ns);
    }
    stage_rows(c, c.L.grav, m.gravity, 3, 1);
    if (with_control) {
        stage_rows(c, c.L.cf, c.a.c.joint_f, 1, m.nd);
        stage_rows(c, c.L.ctq, c.a.c.joint_target_q, 1, m.ntq);
        stage_rows(c, c.L.ctqd, c.a.c.joint_target_qd, 1, m.nd);
    }
}

// state (optional) + parameters + gravity + controls of the lane's environment in one batch of loads (see first_load): what the XPBD /
// collide kernels call instead of load_state + load_params
template <int EPB>
NT_DI void load_tile(const Ctx<EPB>& c, const nt_state* state, bool with_control) {
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    constexpr bool UNI = Ctx<EPB>::UNI;
    if constexpr (UNI) load_uniform_params(c);  // (the block-shared parameter copy: workgroup-strided loops)
    if (!c.valid) return;
    float vq[7], vqd[6], vbp[NT_BODY_PARAM_FLOATS], vjp[NT_JOINT_PARAM_FLOATS], vdp[NT_DOF_PARAM_FLOATS], vsp[NT_SHAPE_PARAM_FLOATS];
    float vg[3] = {0.0f, 0.0f, 0.0f}, vcf = 0.0f, vctq = 0.0f, vctqd = 0.0f;
    int flags = 0;
    if (state) {
        first_load(c, state->body_q, nb, vq);
        first_load(c, state->body_qd, nb, vqd);
    }
    if constexpr (!UNI) {
        first_load(c, m.body_param, nb, vbp);
        if (c.slot < nb) flags = m.body_flags[c.slot];
        first_load(c, m.joint_param, m.nj, vjp);
        first_load(c, m.dof_param, m.nd, vdp);
        first_load(c, m.shape_param, m.ns, vsp);
    }
    if (c.slot == 0) {  // (one item of three components: tiny models run with fewer than three slot-threads per environment)
#pragma unroll
        for (int k = 0; k < 3; ++k) vg[k] = m.gravity[(size_t)k * c.ES + c.env];
    }
    if (with_control) {
        if (c.slot < m.nd) { vcf = c.a.c.joint_f[c.g(0, m.nd, c.slot)]; vctqd = c.a.c.joint_target_qd[c.g(0, m.nd, c.slot)]; }
        if (c.slot < m.ntq) vctq = c.a.c.joint_target_q[c.g(0, m.ntq, c.slot)];
    }
    // ---- writes
    if (state) {
        first_store(c, c.L.bq, nb, vq);
        first_store(c, c.L.bqd, nb, vqd);
    }
    if constexpr (!UNI) {
        if (flags & BODY_KINEMATIC) {  // effective (kinematic => 0) inverse mass / inertia (solver.py:173-187)
            vbp[BP_INV_MASS] = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) vbp[BP_INV_INERTIA + k] = 0.0f;
        }
        first_store(c, c.L.bp, nb, vbp);
        first_store(c, c.L.jp, m.nj, vjp);
        first_store(c, c.L.dp, m.nd, vdp);
        first_store(c, c.L.sp, m.ns, vsp);
    }
    if (c.slot == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) c.l(c.L.grav, k, 1, 0) = vg[k];
    }
    if (with_control) {
        if (c.slot < m.nd) { c.l(c.L.cf, 0, m.nd, c.slot) = vcf; c.l(c.L.ctqd, 0, m.nd, c.slot) = vctqd; }
        if (c.slot < m.ntq) c.l(c.L.ctq, 0, m.ntq, c.slot) = vctq;
    }
    // ---- items beyond the first trip
    if (state) {
        stage_rows_rest(c, c.L.bq, state->body_q, 7, nb);
        stage_rows_rest(c, c.L.bqd, state->body_qd, 6, nb);
    }
    if constexpr (!UNI) {
        for (int b = c.slot + c.nslot; b < nb; b += c.nslot)
            for (int comp = 0; comp < NT_BODY_PARAM_FLOATS; ++comp) {
                float v = m.body_param[(size_t)(comp * nb + b) * c.ES + c.env];
                const bool inv_row = comp == BP_INV_MASS || (comp >= BP_INV_INERTIA && comp < BP_INV_INERTIA + 9);
                if (inv_row && (m.body_flags[b] & BODY_KINEMATIC)) v = 0.0f;
                c.lds[(c.L.bp.off + b * NC_BP + comp) * Ctx<EPB>::N + c.e] = v;
            }
        stage_rows_rest(c, c.L.jp, m.joint_param, NT_JOINT_PARAM_FLOATS, m.nj);
        stage_rows_rest(c, c.L.dp, m.dof_param, NT_DOF_PARAM_FLOATS, m.nd);
        stage_rows_rest(c, c.L.sp, m.shape_param, NT_SHAPE_PARAM_FLOATS, m.ns);
    }
    if (with_control) {
        stage_rows_rest(c, c.L.cf, c.a.c.joint_f, 1, m.nd);
        stage_rows_rest(c, c.L.ctq, c.a.c.joint_target_q, 1, m.ntq);
        stage_rows_rest(c, c.L.ctqd, c.a.c.joint_target_qd, 1, m.nd);
    }
}
