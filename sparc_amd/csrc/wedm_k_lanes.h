// wedm_k_lanes.h — wedm_step_lanes<L>: any geometry, the cell-by-cell LDS walk (kernel 10; kernel 2 under stencil_mode 1).
// Its packed form is wedm_lanes2.h.
//
// Included by wedm_kernels.hip (one translation unit per WEDM_PART; see the bottom of that file).
#pragma once

// Any geometry (uniform or one row per environment), L lanes per environment, every cell on the
// predicated formula with the lane's own n_seg / zone / contact indices.  LDS layout and halo
// exchange as in the fused kernels; the chunk length is uniform, C = ceil(n_seg_max / L), so an
// environment with a shorter wire simply leaves the tail of its last chunks unused.
template <int L, bool TRACE, bool F64>
__global__ void __launch_bounds__(256, 2) wedm_step_lanes(const KArgs k) {
    constexpr bool kFrozenOk = true;  // (predicated cells: a frozen lane costs this kernel nothing extra)
    const ColdRef cold = kernarg_cold();
#ifndef WEDM_NO_PIN_LANES
    Hot hv = k.hot;
    pin_hot_in_vgprs(hv);
#else
    const Hot& hv = k.hot;
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int EPB = 256 / L;
    const int tid = threadIdx.x;
    const int el = tid / L, c = tid % L;
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int64_t e = e0 + el;
    const bool live = e < k.num_envs;
    const int nmax = k.n_seg_max;
    const int C = (nmax + L - 1) / L;
    const int64_t stride = cold->s.stride;
    // wire cell i -> chunk i / C, cell i % C -> LDS [cell][256 lanes], lane = environment slot * L + chunk
    const auto wire_slot = [C](int i) { const int ci = i / C; return (i - ci * C) * 256 + ci; };
    copy_wire<L, true>(cold->s.T, stride, e0, k.num_envs, nmax, tid, lds, wire_slot);
    __syncthreads();

    Env s;
    Geom g;
    Persist ps{0.0f, 0.0f, 0.0f, 0};
    load_geom(k.hot, cold, live ? e : 0, g);
    if (live) load_env(cold, e, s);
    else { s.done = WEDM_DEAD_LANE; s.unwind = 0.0; s.h_base = 0.0f; s.h_zone = 0.0f; }
    float* col = lds + tid;
    const bool reinit = live && s.done && WEDM_AUTORESET(cold);  // next-step autoreset (all L lanes of the environment agree)
    if (reinit) {
        reinit_env(cold, e, s, c == 0);
        for (int j = 0; j < C; ++j) col[j * 256] = k.hot.spool;
    }
    unfreeze_wire(k.hot, s);  // keep_stepping_terminated: the DONE row is `terminated` of the last step and freezes nothing
    const bool frozen0 = s.done;
    WEDM_REPORT_FROZEN(frozen0 && live);
    if (!s.done) {
        s.ipk = peak_current(cold, s.mode, e);
        init_persist(k.hot, cold, e, s, ps);
    }
    const uint32_t gid = k.hot.env_id_offset + (uint32_t)e;
    const int cbase = c * C;
    const int n = g.n_seg;  // this lane's environment
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    StencilF64 f64c{0.0, 0.0, 0.0};
    if (F64) { const wedm_params* pp = cold->p; f64c = StencilF64{pp->temp_ref, pp->alpha_rho, pp->dielectric_temperature}; }
    if (c == 0) col[0] = spool;

    const bool tracing = WEDM_TRACING(k);
    int trace_next = k.trace_next, trace_slot = k.trace_slot;
    (void)trace_next; (void)trace_slot;
    for (int it = 0; it < k.n_substeps; ++it) {
        if (__all(s.done) && !tracing) break;
        Coef cf{0.0f, 0.0f, 0, -1};
        QuietTry qt;
        if (!quiet_prelude_t<WEDM_FUSED_DENSE>(hv, cold, g, e, gid, s, qt, cf) && !s.done) cf = scalar_prelude(hv, cold, g, e, gid, s, ps, c == 0, qt);
        freeze_wire(s);
        const float halo_l = (c > 0) ? col[(C - 1) * 256 - 1] : spool;
        const float halo_r = (c < L - 1) ? col[1] : 0.0f;
        float tmax = spool, tm1 = halo_l, tc = col[0];
#ifndef WEDM_LANES_PREDICATED_ONLY
        // Fast walk (float32 stencil, no negative plasma heat in the wave): every cell of the chunk takes the interior
        // formula, stage-major, eight at a time, with ITS OWN coefficients (two range tests against this lane's zone and
        // contact indices per cell); the cells the interior formula is wrong for -- wire cell 0, the last cell, the plasma
        // cell -- are computed by the predicated formula from OLD values before the walk and written after it, and
        // together with the cells past this environment's wire they are kept out of the maximum.  Same results as the
        // predicated walk below (the uniform-geometry kernels rely on the same equivalence), ~23 instead of ~40
        // instructions per cell.
        if (!F64 && !__any(cf.q < 0.0f)) {
            const bool keep = !s.done;
            const bool owns_pl = keep && cf.pidx >= 1 && cf.pidx >= cbase && cf.pidx < cbase + C && cf.pidx < n;
            const bool owns_last = keep && n >= 2 && (n - 1 >= cbase) && (n - 1 < cbase + C);
            float tpl = 0.0f, tlast = 0.0f;
            if (__any(owns_pl)) {
                if (owns_pl) {
                    const int jp = cf.pidx - cbase;
                    float tm = jp > 0 ? col[(jp - 1) * 256] : halo_l;
                    if (cf.pidx == 1) tm = spool;
                    const float tp = jp < C - 1 ? col[(jp + 1) * 256] : halo_r;
                    tpl = stencil_cell(cf.pidx, n, tm, col[jp * 256], tp, g, cf, ps, tref, alpha, tdiel);
                }
            }
            if (owns_last) {
                const int jl = n - 1 - cbase;
                float tm = jl > 0 ? col[(jl - 1) * 256] : halo_l;
                if (n - 1 == 1) tm = spool;
                tlast = stencil_cell(n - 1, n, tm, col[jl * 256], 0.0f, g, cf, ps, tref, alpha, tdiel);
            }
            const float jf_lane = (cf.joule_on && keep) ? cf.jf : 0.0f;
            const bool joule_wave = __any(jf_lane != 0.0f);
            const uint32_t zs = (uint32_t)g.az_start, zw = g.az_end > g.az_start ? (uint32_t)(g.az_end - g.az_start) : 0u;
            const uint32_t cbot = (uint32_t)g.cb, cw = g.ct >= g.cb ? (uint32_t)(g.ct - g.cb + 1) : 0u;
            const uint32_t span = n >= 3 ? (uint32_t)(n - 3) : 0u;
            for (int j0 = 0; j0 < C; j0 += 8) {
                float old[10], tn[8], cv[8], jv[8];
                old[0] = tm1; old[1] = tc;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = j0 + 1 + u;
                    old[u + 2] = row < C ? col[row * 256] : halo_r;
                    const uint32_t i = (uint32_t)(cbase + j0 + u);
                    cv[u] = (i - zs < zw) ? ps.conv_zone : ps.conv_base;   // az_start <= i < az_end
                    jv[u] = (i - cbot < cw) ? jf_lane : 0.0f;               // contact_bottom <= i <= contact_top
                }
                if (joule_wave) tile8_staged<float, true, true>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                else tile8_staged<float, false, true>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                if (keep) {
                    // (rows past this environment's wire keep their value: the write-back copies all n_seg_max rows)
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + u < C) col[(j0 + u) * 256] = (cbase + j0 + u < n) ? tn[u] : old[u + 1];
                }
                float mx[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t im1 = (uint32_t)(cbase + j0 + u) - 1u;  // interior: 1 <= i <= n - 2
                    mx[u] = (n >= 3 && j0 + u < C && im1 <= span) ? tn[u] : spool;
                }
                tmax = fmax_gt(tmax, fmax_gt(fmax_gt(fmax_gt(mx[0], mx[1]), fmax_gt(mx[2], mx[3])),
                                             fmax_gt(fmax_gt(mx[4], mx[5]), fmax_gt(mx[6], mx[7]))));
                tm1 = old[8];
                tc = old[9];
            }
            if (c == 0 && keep) col[0] = spool;
            if (owns_last) { col[(n - 1 - cbase) * 256] = tlast; tmax = fmax_gt(tmax, tlast); }
            if (owns_pl) { col[(cf.pidx - cbase) * 256] = tpl; tmax = fmax_gt(tmax, tpl); }
        } else
#endif
        for (int j0 = 0; j0 < C; j0 += 8) {
            float nx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = j0 + 1 + u;
                nx[u] = row < C ? col[row * 256] : halo_r;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j < C) {
                    const int i = cbase + j;
                    if (i < n && !s.done) {
                        float tn = spool;
                        if (i >= 1) {
                            if (F64) tn = stencil_cell_f64(i, n, (i == 1) ? spool : tm1, tc, nx[u], g, cf, ps, f64c, s.h_base, s.h_zone);
                            else tn = stencil_cell(i, n, (i == 1) ? spool : tm1, tc, nx[u], g, cf, ps, tref, alpha, tdiel);
                        }
                        col[j * 256] = tn;
                        tmax = fmax_gt(tmax, tn);
                    }
                    tm1 = tc;
                    tc = nx[u];
                }
            }
        }
#pragma unroll
        for (int m = 1; m < L; m <<= 1) tmax = fmax_gt(tmax, __shfl_xor(tmax, m));
        unfreeze_wire(hv, s);
        if (!s.done) {
            scalar_epilogue(hv, s, tmax);
            if (s.ctrl) control_step_outputs(cold, e, s, c == 0);
        }
        WEDM_TRACE_POINT(k, it, e, s, c == 0,
                         for (int j = 0; j < C && cbase + j < n; ++j) tT[(int64_t)(cbase + j) * tcnt] = col[j * 256]);
    }

    __syncthreads();
    copy_wire<L, false>(cold->s.T, stride, e0, k.num_envs, nmax, tid, lds, wire_slot);
    if (live && c == 0) {
        if (WEDM_REWARD_ON(cold)) {
            if (!frozen0) write_reward(cold, e, s);
            else cold->s.reward[e] = 0.0f;  // a frozen environment earns nothing (not the previous launch's reward)
        }
        store_time_hi(cold, e, s, (uint32_t)k.n_substeps * (uint32_t)k.hot.dt_us);
        store_env(cold, e, s);
    }
}

