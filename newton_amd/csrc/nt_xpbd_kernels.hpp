// nt_xpbd_kernels.hpp -- the collide / XPBD step / fused rollout kernels: collide phases in namespace ieee (this file), the XPBD
// projection phases in namespace fused (nt_xpbd.hpp compiled with contraction + v_rcp / v_sqrt), both on the same LDS tile.
#pragma once

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
template <int EPB, bool CVX>
NT_DI void do_collide(const Ctx<EPB>& c, bool count_contacts) {
    NT_SKIP_DECL(c.a);
    if (NT_SKIP(1)) return;
    const bool compact = pairs_compacted(c);
    if (compact && c.valid && c.slot == 0) *reinterpret_cast<int*>(&c.l(c.L.hc, 0, 1, 0)) = 0;
    phase_shapes(c);
    __syncthreads();
    NT_TICK(1);
    if (c.big) {
        phase_pairs_big_broad(c);  // pair-heavy tile: compact the candidates first, no candidate staging (19 rows per pair)
        phase_pairs_big_narrow<EPB, CVX>(c);
    } else {
        if (compact) {
            phase_pair_broad_staged(c);
            __syncthreads();
            phase_pair_narrow_staged<EPB, CVX>(c);
        } else {
            phase_pair_eval<EPB, CVX>(c);
        }
        __syncthreads();
        phase_contact_write(c);
    }
    __syncthreads();  // also publishes the contact records (global memory) to the block's contact lanes
    NT_TICK(2);
    // live-contact prefix for the fused solver phases + (last substep / standalone collide) the per-env totals; scratch =
    // the shape-transform rows of the collide scratch, dead once the pairs are done
    const bool one_level = c.a.m.np <= 64;
    if (!one_level) {
        phase_pair_prefix_partials(c, c.L.sx.off);
        __syncthreads();
    }
    phase_pair_prefix_scan(c, c.L.sx.off, count_contacts, one_level);
    __syncthreads();
}

// THREADS / MINW: workgroup size and minimum waves per SIMD the register allocator must leave room for
// (k workgroups of THREADS threads per CU <=> MINW = k * THREADS / 256); chosen per model by the launch code
// One substep of the fused rollout on the staged tiles: { clear_forces; collide; SolverXPBD.step } with the independent
// 13-item phases of the two halves sharing barrier intervals.  The shape phase and the joint-force phase both read only the
// incoming state (collide.py:283-472 / xpbd/kernels.py:945-1075) and write disjoint scratch (layout: forces behind the
// collide scratch), so they run side by side on different waves; the contact-record stage, the live-contact prefix and
// nothing else follow; integrate_bodies comes last because the records are converted into the frames of the incoming
// body poses (collide.py:166-204).  Arithmetic and summation orders are those of do_collide + do_xpbd_step, bit for bit.
// c: the lane's context for the collide phases (namespace ieee); cf: the same lane for the XPBD phases (namespace fused)
template <int EPB, bool CVX>
NT_DI void do_fused_substep(const Ctx<EPB>& c, const fused::Ctx<EPB>& cf, bool last_substep) {
    const nt_model& m = c.a.m;
    NT_SKIP_DECL(c.a);
    const int spw = 64 / Ctx<EPB>::N > 0 ? 64 / Ctx<EPB>::N : 1;  // slots per wave
    const bool restitution = (c.a.p.enable_restitution && c.a.has_contacts) || c.a.p.compute_body_velocity_from_position_delta != 0;
    // -- interval 1: shapes (slots [0, ns)) || joint forces (slots [S0, S0 + nj), S0 on a wave boundary) + body_f_tmp = 0
    const bool compact = pairs_compacted(c);
    // integrate_bodies beside the pair phase (its 13 lanes on the waves the 13 pair lanes leave idle; the pair phase is the longer of
    // the two): the contact writer then converts into the snapshot of the incoming poses (c.pose_in_off) taken in interval 1
    const int I0 = ((m.np + spw - 1) / spw) * spw;
    const bool overlap = c.pose_in_off != c.L.bq.off;
    if (c.valid) {
        if (compact && c.slot == 0) *reinterpret_cast<int*>(&c.l(c.L.hc, 0, 1, 0)) = 0;
        if (c.lds_records && c.slot == 0) *reinterpret_cast<int*>(&c.l(c.L.lc, 0, 1, 0)) = 0;
        if (restitution || overlap)
            for (int r = c.slot; r < (restitution ? 14 : 7) * m.nb; r += c.nslot) c.lds[(c.L.xiq.off + r) * Ctx<EPB>::N + c.e] = c.lds[(c.L.bq.off + r) * Ctx<EPB>::N + c.e];
        if (!NT_SKIP(2)) fused::seed_body_forces(cf, true);
        const int S0 = ((m.ns + spw - 1) / spw) * spw;
        for (int i = c.slot; i < S0 + m.nj; i += c.nslot) {
            if (i < m.ns) {
                if (!NT_SKIP(1)) shape_item(c, i);
            } else if (i >= S0) {
                if (!NT_SKIP(2)) fused::joint_force_item(cf, i - S0);
            }
        }
    }
    __syncthreads();
    NT_TICK(1);
    // -- interval 2: one lane per candidate pair (broad phase test, primitive pair / MPR-GJK manifold, admission)
    if (!NT_SKIP(1)) {
        if (compact) {
            phase_pair_broad_staged(c);
            __syncthreads();
            phase_pair_narrow_staged<EPB, CVX>(c);
        } else if (overlap && c.slot >= I0) {
            if (c.valid && c.slot < I0 + m.nb && !NT_SKIP(2)) fused::xpbd_integrate_item(cf, c.slot - I0, true, true, !fused::xpbd_applies_follow(c.a));
        } else {
            phase_pair_eval<EPB, CVX>(c);
        }
    }
    __syncthreads();
    NT_TICK(2);
    // -- interval 3: contact records of the analytic pairs (one lane per slot) || live-contact prefix (few lanes per env)
    if (!NT_SKIP(1) && c.valid && c.lds_records) {
        // LDS-record tiles: one lane per LIVE contact (the pair lanes built the list), records into L.cr
        const int total = *reinterpret_cast<const int*>(&c.l(c.L.lc, 0, 1, 0));
        for (int i = c.slot; i < total; i += c.nslot) contact_record_item_lds(c, *reinterpret_cast<const int*>(&c.l(c.L.lt, 0, 1, i)));
        if (c.slot == c.nslot - 1) {
            c.l(c.L.px, 0, 1, m.np) = (float)total;
            if (last_substep) c.a.ct.env_count[c.env] = total;  // per-env totals: an API-boundary output
        }
    } else if (!NT_SKIP(1) && c.valid) {
        const int nas = m.np_analytic * m.cpp;
        const int P0 = ((nas + spw - 1) / spw) * spw;
        const bool one_level = m.np <= 64;
        if (one_level) {
            for (int i = c.slot; i < P0 + NT_PREFIX_LANES; i += c.nslot) {
                if (i < nas) contact_write_item(c, i);
                else if (i >= P0) prefix_lane(c, i - P0, c.L.sx.off, last_substep, true);
            }
        } else {
            for (int s = c.slot; s < nas; s += c.nslot) contact_write_item(c, s);
            phase_pair_prefix_partials(c, c.L.sx.off);
        }
    }
    __syncthreads();  // also publishes the contact records (global memory) to the block's contact lanes
    if (!NT_SKIP(1) && m.np > 64 && !c.lds_records) {
        phase_pair_prefix_scan(c, c.L.sx.off, last_substep, false);
        __syncthreads();
    }
    if (c.lds_records && last_substep && c.valid)  // the launch's Contacts output (reads L.cr / L.pm only: no barrier needed behind it)
        for (int s = c.slot; s < m.np * m.cpp; s += c.nslot) contact_export_item_lds(c, s);
    NT_TICK(3);
    // -- interval 4: integrate_bodies (unless it ran beside the pairs)
    if (!overlap) {
        if (!NT_SKIP(2)) fused::phase_xpbd_integrate(cf, !fused::xpbd_applies_follow(c.a));
        __syncthreads();
    }
    NT_TICK(4);
    fused::do_xpbd_step<EPB, true, fused::CwLds, true>(cf, true);
}

template <int EPB, bool CVX, bool BIG = false, int THREADS = ((EPB & 255) <= 8 ? 256 : 512), int MINW = 1>
__global__ void __launch_bounds__(THREADS, MINW) collide_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds, -1, BIG);
    c.lds_records = false;  // (compile-time facts for this kernel: the LDS-record code folds away)
    load_tile(c, &a.s_in, false);
    __syncthreads();
    stage_global_world(c);  // (beside the shape phase; the pair phase behind its barrier reads it)
    c.gworld_ready = true;
    do_collide<EPB, CVX>(c, true);
}

// compute_shape_aabbs alone (models whose every pair belongs to a stage outside the tiles: the launch only exports
// nt_contacts.world_xform / world_aabb_*)
template <int EPB>
__global__ void __launch_bounds__((EPB & 255) <= 8 ? 256 : 512) shapes_export_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds, -1, false);
    c.lds_records = false;
    load_tile(c, &a.s_in, false);
    __syncthreads();
    phase_shapes(c);
}

template <int EPB, bool BIG = false, int THREADS = ((EPB & 255) <= 8 ? 256 : 512), int MINW = 1>
__global__ void __launch_bounds__(THREADS, MINW) xpbd_step_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds, -1, BIG);
    c.lds_records = false;
    const fused::Ctx<EPB> cf(c, 0);
    load_tile(c, &a.s_in, true);
    __syncthreads();
    fused::phase_body_derived(cf);
    c.stage_joint_inc();
    __syncthreads();
    {
        using CW = std::conditional_t<BIG, fused::CwHbm, fused::CwLds>;  // where the per-contact correction records live
        // Without reporting outputs the step runs the contact phases of the fused rollout: the live slots of every pair
        // (a pair's contacts fill its slots from the front: collide.py's writer order) are counted from the Contacts buffers, turned
        // into the live prefix (+ compacted list where the layout has one), and only live contacts are solved / summed -- instead of
        // all np * cpp slots with their shape ids re-read from HBM in every iteration (the standing quadruped: 16 of 52).
        // Rows of the SDF legs do not stand in the way (round 6): the compacted slots are solved first, the rows behind them, the body
        // lanes sum slots then rows -- the order of the uncompacted step (config C5's bin: 1 600 wall slots per world, a handful live;
        // their ids were re-read from HBM in both iterations, one 128-byte line per 4-byte id with one world per workgroup).
        const bool compact = a.has_contacts && a.m.np > 0 && !a.rep.joint_impulse && !a.rep.contact_impulse && !a.s_out.body_parent_f;
        if (compact) {
            const int np = a.m.np, cpp = a.m.cpp;
            if (c.valid)
                for (int p = c.slot; p < np; p += c.nslot) {
                    int n = 0;
                    for (int k = 0; k < cpp; ++k) {
                        const size_t gi = (size_t)(p * cpp + k) * c.ES + c.env;
                        n += a.ct.shape0[gi] != a.ct.shape1[gi] ? 1 : 0;
                    }
                    c.l(c.L.pm, 0, np, p) = (float)n;
                }
            __syncthreads();
            const bool one_level = np <= 64;
            if (!one_level) {
                phase_pair_prefix_partials(c, c.L.sx.off);
                __syncthreads();
            }
            phase_pair_prefix_scan(c, c.L.sx.off, false, one_level);
            __syncthreads();
            if (a.ct.flat.row_start) fused::do_xpbd_step<EPB, true, CW, false, true>(cf, false);
            else fused::do_xpbd_step<EPB, true, CW>(cf, false);
        } else {
            fused::do_xpbd_step<EPB, false, CW>(cf, false);
        }
    }
    store_state(c, a.s_out);
}

// substeps x { clear_forces; collide; step; swap } with state and parameters resident in LDS across substeps.
// Only the final state is stored (into s0 for an even number of substeps, s1 for odd, like the reference's
// pointer swap); body_f of both states is zeroed as clear_forces would leave it.
template <int EPB, bool CVX, bool BIG = false, int THREADS = ((EPB & 255) <= 8 ? 256 : 512), int MINW = 1>
__global__ void __launch_bounds__(THREADS, MINW) xpbd_rollout_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    NT_TICK_START();
    Ctx<EPB> c(a, lds, -1, BIG);
    if constexpr (CVX || BIG) c.lds_records = false;  // (granted to analytic-only staged tiles alone: folds the LDS-record code away here)
    c.lane_split = !(CVX || Ctx<EPB>::N >= 32);  // (a compile-time fact per kernel; 32-environment tiles have no idle waves to split onto)
    if constexpr (!BIG) {  // (the layout holds the snapshot rows, the pairs fit one pass and the integrate lanes fit behind them)
        constexpr int spw = 64 / Ctx<EPB>::N > 0 ? 64 / Ctx<EPB>::N : 1;
        if ((a.tile_opts & NT_TILE_POSE_SNAPSHOT) && a.m.np <= a.nslot && ((a.m.np + spw - 1) / spw) * spw + a.m.nb <= a.nslot)
            c.pose_in_off = c.L.xiq.off;
    }
    // pair-heavy tile: the substeps' contact records go through nt_contacts.cr, the Contacts buffers get the last substep's (the plain
    // position solve only: restitution reads the API layout)
    if constexpr (BIG) c.aos_records = a.ct.cr != nullptr && !xpbd_keeps_prestep_state(a.p);
    const fused::Ctx<EPB> cf(c, 0);
    const int nb = a.m.nb;
    load_tile(c, &a.s_in, true);
    if (c.valid)
        for (int r = c.slot; r < 6 * nb; r += c.nslot) {
            a.s_in.body_f[(size_t)r * c.ES + c.env] = 0.0f;
            a.s_out.body_f[(size_t)r * c.ES + c.env] = 0.0f;
        }
    __syncthreads();
    fused::phase_body_derived(cf);
    c.stage_joint_inc();
    stage_global_world(c);  // (the topology tables it reads were published by the barrier above)
    c.gworld_ready = true;
    __syncthreads();
    NT_TICK(0);
    for (int s = 0; s < a.substeps; ++s) {
        c.hbm_out = !(c.lds_records || c.aos_records) || s == a.substeps - 1;
        if constexpr (BIG) {
            do_collide<EPB, CVX>(c, s == a.substeps - 1);
            fused::do_xpbd_step<EPB, true, fused::CwHbm>(cf, true);
        } else {
            do_fused_substep<EPB, CVX>(c, cf, s == a.substeps - 1);
        }
    }
    store_state(c, (a.substeps & 1) ? a.s_out : a.s_in);
    NT_TICK(9);
}
