// wedm_lanes2.h — wedm_step_lanes_pk<L>: any geometry (one (h, d) pair per environment: BASELINE config 5), packed float32 walk.
//
// Included by wedm_kernels.hip.  wedm_step_lanes walks a chunk one float32 cell at a time: 11 operations of the interior
// formula + two range tests against the lane's own zone / contact indices + store and maximum predicates per CELL, 23
// wave-instructions per cell (profiles/valu.json, round 3: 213.8 per env-step at 16 384 x <= 450 segments).  Here a lane
// owns TWO virtual chunks A and B of Cv = ceil(n_seg_max / 2L) cells, as in wedm_step_packed, and advances them together in
// float2 registers: the 11 operations cover two cells (v_pk_add / mul / fma_f32), the predicates stay per cell (there is no
// packed compare or select), ~13 instructions per cell.  LDS image and halo exchange are wedm_step_packed's ([row 2 r + v][256
// lanes], rows 2 Cv and 2 Cv + 1 hold the right halos); there is NO tile table -- geometry differs from lane to lane, so
// every tile takes the per-cell-coefficient code and the cells the interior formula is wrong for (wire cell 0, an
// environment's last cell, a plasma cell) are computed by the predicated formula from OLD values before the walk and written
// after it, exactly as wedm_step_lanes does.  Cells past an environment's wire keep their value (the write-back copies all
// n_seg_max rows).  A wave with a negative plasma heat walks cell by cell on the predicated formula (same results).
// F64 (round 4): the same walk with every interior cell in Numba's typing of wire.py:58-123 (cell_f64 of wedm_common.h: 18 float64
// operations per cell, the two cells of a pair one after the other) -- stencil_mode 1 on per-environment geometry.
#pragma once

// One microsecond of the packed any-geometry walk for this lane's two virtual chunks (LDS rows of NT floats); returns the
// lane's maximum (the caller reduces it over the environment's L lanes).  `keep`: the environment is live (a terminated one's
// wire stays as it is).
struct LanesPkGeom {   // per lane, fixed for the launch
    int c, Cv, R, n, baseA, baseB, own_last;
    uint32_t zs, zw, cbot, cw, span;
    bool has_inner;
};
template <int L, int NT, bool F64 = false>
__device__ __forceinline__ float lanes_pk_step(float* col, const LanesPkGeom& q, bool keep, const Geom& g, const Coef& cf, const Persist& ps,
                                               float spool, float tref, float alpha, float tdiel,
                                               const StencilF64& h64 = StencilF64{0.0, 0.0, 0.0}, float h_base = 0.0f, float h_zone = 0.0f) {
    const int c = q.c, Cv = q.Cv, R = q.R, n = q.n, baseA = q.baseA, baseB = q.baseB, own_last = q.own_last;
    const uint32_t zs = q.zs, zw = q.zw, cbot = q.cbot, cw = q.cw, span = q.span;
    const bool has_inner = q.has_inner;
    auto owner = [&](int i) -> int {  // which of this lane's virtual chunks holds wire cell i (0: none, 1: A, 2: B)
        if (i >= baseA && i < baseA + Cv) return 1;
        if (i >= baseB && i < baseB + Cv) return 2;
        return 0;
    };
    // ---- halos (OLD values, read before any store of this step)
    const float halo_l = (c > 0) ? col[(R - 1) * NT - 1] : spool;  // left neighbour lane's B[Cv-1]
    const float halo_r = (c < L - 1) ? col[1] : 0.0f;               // right neighbour lane's A[0]
    const float a_last = col[(R - 2) * NT];                        // own A[Cv-1]: left halo of B
    const float b_first = col[NT];                                 // own B[0]: right halo of A
    col[R * NT] = b_first;
    col[(R + 1) * NT] = halo_r;
    float tmax = spool;

    if (!__any(cf.q < 0.0f)) {
        // full predicated formula for one owned cell, from OLD values (patched cells); rows R, R + 1 are the halo pair
        auto patch_value = [&](int i, int own, bool last) -> float {
            const int v = own - 1, r = i - (v ? baseB : baseA), row = 2 * r + v;
            const float left = col[(r > 0 ? row - 2 : row) * NT];
            float tm = r > 0 ? left : (v ? a_last : halo_l);
            if (i == 1) tm = spool;
            const float tp = last ? 0.0f : col[(row + 2) * NT];
            return rw_cell<F64>(i, n, tm, col[row * NT], tp, g, cf, ps, tref, alpha, tdiel, h64, h_base, h_zone);
        };
        const int own_pl = (keep && cf.pidx >= 1 && cf.pidx < n) ? owner(cf.pidx) : 0;
        float tpl = 0.0f, tlast = 0.0f;
        if (__any(own_pl != 0)) {
            if (own_pl) tpl = patch_value(cf.pidx, own_pl, false);
        }
        if (own_last && keep) tlast = patch_value(n - 1, own_last, true);

        // (float64 typing: the Joule entry is a flag, the factor is cf.jf64; the convection entries are the float32 h_eff themselves)
        const float jf_lane = (cf.joule_on && keep) ? (F64 ? 1.0f : cf.jf) : 0.0f;
        const bool joule_wave = __any(jf_lane != 0.0f);
        const float cz = F64 ? h_zone : ps.conv_zone, cb = F64 ? h_base : ps.conv_base;
        f2 tm1 = {halo_l, a_last};
        f2 tc = {col[0], col[NT]};
        for (int r0 = 0; r0 < Cv; r0 += 8) {
            f2 old[10], tn[8];
            old[0] = tm1; old[1] = tc;
            if (r0 + 8 <= Cv) {  // a full tile: one base address + immediate offsets
                const float* const row = col + (2 * (r0 + 1)) * NT;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    old[u + 2].x = row[(2 * u) * NT];
                    old[u + 2].y = row[(2 * u + 1) * NT];
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int p = r0 + 1 + u;
                    p = p < Cv ? p : Cv;  // pair Cv is the halo pair; later pairs are never used
                    old[u + 2].x = col[(2 * p) * NT];
                    old[u + 2].y = col[(2 * p + 1) * NT];
                }
            }
            // the pairs' coefficients group by group, right before the stage-major group that uses them (all eight up
            // front are 32 registers more than the Env leaves)
            constexpr int W = WEDM_STAGE_W_PACKED;
#pragma unroll
            for (int o = 0; o < 8; o += W) {
                f2 cv[8], jv[8];  // (only entries o .. o + W - 1 are set and read)
#pragma unroll
                for (int u = o; u < o + W; ++u) {
                    const uint32_t ia = (uint32_t)(baseA + r0 + u), ib = (uint32_t)(baseB + r0 + u);
                    cv[u] = f2{(ia - zs < zw) ? cz : cb, (ib - zs < zw) ? cz : cb};
                    jv[u] = f2{(ia - cbot < cw) ? jf_lane : 0.0f, (ib - cbot < cw) ? jf_lane : 0.0f};
                }
                if (F64) {
#pragma unroll
                    for (int u = o; u < o + W; ++u) {
                        const double cA = (double)cv[u].x * g.a64, cB = (double)cv[u].y * g.a64;
                        if (joule_wave) {
                            tn[u].x = cell_f64<true>(old[u].x, old[u + 1].x, old[u + 2].x, g.k64, g.tuf64, cA, h64.tdiel, ps.adv64,
                                                     jv[u].x != 0.0f ? cf.jf64 : 0.0, h64.alpha, h64.tref);
                            tn[u].y = cell_f64<true>(old[u].y, old[u + 1].y, old[u + 2].y, g.k64, g.tuf64, cB, h64.tdiel, ps.adv64,
                                                     jv[u].y != 0.0f ? cf.jf64 : 0.0, h64.alpha, h64.tref);
                        } else {
                            tn[u].x = cell_f64<false>(old[u].x, old[u + 1].x, old[u + 2].x, g.k64, g.tuf64, cA, h64.tdiel, ps.adv64, 0.0, h64.alpha, h64.tref);
                            tn[u].y = cell_f64<false>(old[u].y, old[u + 1].y, old[u + 2].y, g.k64, g.tuf64, cB, h64.tdiel, ps.adv64, 0.0, h64.alpha, h64.tref);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else if (joule_wave) {
                    tile_staged<f2, true, true, W>(old, tn, o, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                } else {
                    tile_staged<f2, false, true, W>(old, tn, o, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                }
            }
            // ONE predicate per cell for store and maximum: interior (1 <= i <= n - 2) and this environment live.  Wire cell
            // 0 and the last cell are written after the walk anyway (spool temperature, the patched value), cells past the
            // wire keep their value (the write-back copies all n_seg_max rows), a terminated environment's wire stays: all of
            // them simply store their OLD value back (a select, not a branch: 41 exec-mask round trips per tile otherwise).
            float mx[8];
            const uint32_t ja = (uint32_t)(baseA + r0) - 1u, jb = (uint32_t)(baseB + r0) - 1u;
            const uint32_t lim = (keep && has_inner) ? span : 0u;
            const bool any_ok = keep && has_inner;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool inA = any_ok && (ja + (uint32_t)u <= lim), inB = any_ok && (jb + (uint32_t)u <= lim);
                if (r0 + u < Cv) {  // (wave-uniform)
                    col[(2 * (r0 + u)) * NT] = inA ? tn[u].x : old[u + 1].x;
                    col[(2 * (r0 + u) + 1) * NT] = inB ? tn[u].y : old[u + 1].y;
                    mx[u] = fmax_gt(inA ? tn[u].x : spool, inB ? tn[u].y : spool);
                } else {
                    mx[u] = spool;
                }
            }
            tmax = fmax_gt(tmax, fmax_gt(fmax_gt(fmax_gt(mx[0], mx[1]), fmax_gt(mx[2], mx[3])),
                                         fmax_gt(fmax_gt(mx[4], mx[5]), fmax_gt(mx[6], mx[7]))));
            tm1 = old[8];
            tc = old[9];
        }
        // ---- patches (after every store of the walk): boundary condition, last cell, plasma cell
        if (c == 0 && keep) col[0] = spool;
        if (own_last && keep) {
            const int v = own_last - 1;
            col[(2 * (n - 1 - (v ? baseB : baseA)) + v) * NT] = tlast;
            tmax = fmax_gt(tmax, tlast);
        }
        if (own_pl) {
            const int v = own_pl - 1;
            col[(2 * (cf.pidx - (v ? baseB : baseA)) + v) * NT] = tpl;
            tmax = fmax_gt(tmax, tpl);
        }
    } else {
        // a negative plasma heat somewhere in the wave: every cell on the predicated formula, chunk A then chunk B (each
        // with its rolling window of OLD values; the halos were taken above)
#pragma unroll 1
        for (int v = 0; v < 2; ++v) {
            const int cbase = v ? baseB : baseA;
            float tm1 = v ? a_last : halo_l, tc = col[v * NT];
            for (int r = 0; r < Cv; ++r) {
                const float nx = col[(2 * (r + 1) + v) * NT];  // (row 2 Cv + v: this chunk's right halo)
                const int i = cbase + r;
                if (i < n && keep) {
                    float tn = spool;
                    if (i >= 1) tn = rw_cell<F64>(i, n, (i == 1) ? spool : tm1, tc, nx, g, cf, ps, tref, alpha, tdiel, h64, h_base, h_zone);
                    col[(2 * r + v) * NT] = tn;
                    tmax = fmax_gt(tmax, tn);
                }
                tm1 = tc;
                tc = nx;
            }
        }
    }
    return tmax;
}

template <int L, bool TRACE, bool F64 = false>
__global__ void __launch_bounds__(256, 2) wedm_step_lanes_pk(const KArgs k) {
    const ColdRef cold = kernarg_cold();
    Hot hv = k.hot;
    pin_mechanics_in_vgprs(hv);
    pin_quiet_in_vgprs(hv);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int EPB = 256 / L;
    const int tid = threadIdx.x;
    const int el = tid / L, c = tid % L;
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int64_t e = e0 + el;
    const bool live = e < k.num_envs;
    const int nmax = k.n_seg_max;
    const int Cv = (nmax + 2 * L - 1) / (2 * L);  // cells per virtual chunk
    const int R = 2 * Cv;                          // data rows per lane; rows R and R + 1 are the halo pair
    const int64_t stride = cold->s.stride;
    // wire cell i -> virtual chunk vc = i / Cv, cell r = i % Cv -> lane vc / 2, row 2 r + vc % 2
    const auto wire_slot = [Cv](int i) { const int vc = i / Cv; return (2 * (i - vc * Cv) + (vc & 1)) * 256 + (vc >> 1); };
    copy_wire<L, true>(cold->s.T, stride, e0, k.num_envs, nmax, tid, lds, wire_slot);
    __syncthreads();

    Env s;
    Geom g;
    Persist ps{0.0f, 0.0f, 0.0f, 0};
    load_geom(k.hot, cold, live ? e : 0, g);
    StencilF64 f64c{0.0, 0.0, 0.0};
    if (F64) { const wedm_params* pp = cold->p; f64c = StencilF64{pp->temp_ref, pp->alpha_rho, pp->dielectric_temperature}; }
    if (live) load_env(cold, e, s);
    else { s.done = WEDM_DEAD_LANE; s.unwind = 0.0; s.h_base = 0.0f; s.h_zone = 0.0f; }
    float* col = lds + tid;
    const bool reinit = live && s.done && WEDM_AUTORESET(cold);  // next-step autoreset (all L lanes of the environment agree)
    if (reinit) {
        reinit_env(cold, e, s, c == 0);
        for (int row = 0; row < R; ++row) col[row * 256] = k.hot.spool;
    }
    unfreeze_wire(k.hot, s);  // keep_stepping_terminated: the DONE row is `terminated` of the last step and freezes nothing
    const bool frozen0 = s.done;
    if (!s.done) {
        s.ipk = peak_current(cold, s.mode, e);
        init_persist(k.hot, cold, e, s, ps);
    }
    const uint32_t gid = k.hot.env_id_offset + (uint32_t)e;
    const int baseA = 2 * c * Cv, baseB = baseA + Cv;  // first wire cell of each virtual chunk
    const int n = g.n_seg;                              // this lane's environment
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    if (c == 0) col[0] = spool;  // wire cell 0 (row 0 of lane 0's chunk A) is held at the spool temperature

    // which of this lane's virtual chunks holds wire cell i (0: none, 1: A, 2: B)
    auto owner = [&](int i) -> int {
        if (i >= baseA && i < baseA + Cv) return 1;
        if (i >= baseB && i < baseB + Cv) return 2;
        return 0;
    };
    const int own_last = (n >= 2) ? owner(n - 1) : 0;
    // range tests as one unsigned compare each: az_start <= i < az_end, contact_bottom <= i <= contact_top, 1 <= i <= n - 2
    const uint32_t zs = (uint32_t)g.az_start, zw = g.az_end > g.az_start ? (uint32_t)(g.az_end - g.az_start) : 0u;
    const uint32_t cbot = (uint32_t)g.cb, cw = g.ct >= g.cb ? (uint32_t)(g.ct - g.cb + 1) : 0u;
    const bool has_inner = n >= 3;
    const uint32_t span = has_inner ? (uint32_t)(n - 3) : 0u;
    const LanesPkGeom pkg{c, Cv, R, n, baseA, baseB, own_last, zs, zw, cbot, cw, span, has_inner};

    const bool tracing = WEDM_TRACING(k);
    int trace_next = k.trace_next, trace_slot = k.trace_slot;
    (void)trace_next; (void)trace_slot;
    for (int it = 0; it < k.n_substeps; ++it) {
        if (__all(s.done) && !tracing) break;
        Coef cf{0.0f, 0.0f, 0, -1};
        QuietTry qt;
        const bool was_quiet = quiet_prelude_t<WEDM_PACKED_DENSE>(hv, cold, g, e, gid, s, qt, cf);
        if (!was_quiet && !s.done) cf = scalar_prelude(hv, cold, g, e, gid, s, ps, c == 0, qt);
        freeze_wire(s);
        const bool keep = !s.done;

        float tmax = lanes_pk_step<L, 256, F64>(col, pkg, keep, g, cf, ps, spool, tref, alpha, tdiel, f64c, s.h_base, s.h_zone);
        tmax = max_over_env_lanes<L>(tmax);
        unfreeze_wire(hv, s);
        if (!s.done) {
            scalar_epilogue(hv, s, tmax);
            if (s.ctrl) control_step_outputs(cold, e, s, c == 0);
        }
        WEDM_TRACE_POINT(k, it, e, s, c == 0,
                         for (int r = 0; r < Cv; ++r) {
                             if (baseA + r < n) tT[(int64_t)(baseA + r) * tcnt] = col[(2 * r) * 256];
                             if (baseB + r < n) tT[(int64_t)(baseB + r) * tcnt] = col[(2 * r + 1) * 256];
                         });
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

// ============================================ served form: the scalar physics of the block's environments on a fourth wave
// wedm_step_lanes_pk's walk on three walker waves (192 / L environments per block), prelude and epilogue once per environment
// on the scalar wave, one microsecond ahead of the walkers where it can prove that the step does not break the wire
// (wedm_served.h: mailbox, protocol, proof).  Three blocks per CU at 168 registers.  No trace point, freeze-on-termination
// only (other launches run wedm_step_lanes_pk).
template <int L>
__global__ void __launch_bounds__(256, WEDM_SERVED_WAVES_PER_EU) wedm_step_lanes_served(const KArgs k) {
    constexpr int NT = 192, EPB = NT / L;
    static_assert(EPB <= 64, "one lane of the scalar wave per environment of the block");
    typedef ServedBox<EPB> Box;
    const ColdRef cold = kernarg_cold();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int nmax = k.n_seg_max;
    const int Cv = (nmax + 2 * L - 1) / (2 * L);  // cells per virtual chunk
    const int R = 2 * Cv;
    const int64_t stride = cold->s.stride;
    volatile Box* const box = (volatile Box*)(lds + (size_t)(R + 2) * NT);
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    SvStamps svs;
    sv_stamps_begin(svs);
    unsigned long long* const stamp_row = k.dbg ? k.dbg + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 12 : nullptr;
    if (tid == 0) { box->cf_seq = 0u; box->tm_seq[0] = 0u; box->tm_seq[1] = 0u; box->tm_seq[2] = 0u; box->tm_seq[3] = 0u; }
    if (tid >= NT) {
        served_scalar_wave<EPB, L>(k, cold, box, e0, tid - NT, svs, stamp_row);
        return;
    }

    // ---------------------------------------------------------------------------------------------------- the walker waves
    const int el = tid / L, c = tid % L, wave = tid >> 6;
    const int64_t e = e0 + el;
    const bool live = e < k.num_envs;
    const auto wire_slot = [Cv](int i) { const int vc = i / Cv; return (2 * (i - vc * Cv) + (vc & 1)) * NT + (vc >> 1); };
    copy_wire_nt<L, NT, true>(cold->s.T, stride, e0, k.num_envs, nmax, tid, lds, wire_slot);
    Geom g;
    load_geom(k.hot, cold, live ? e : 0, g);  // this lane's environment
    float* col = lds + tid;
    const bool reinit = live && WEDM_AUTORESET(cold) && cold->s.i8[(int64_t)WEDM_B_DONE * stride + e] != 0;
    __syncthreads();  // (A)
    if (reinit) {
        for (int row = 0; row < R; ++row) col[row * NT] = spool;
    }
    Persist ps{box->adv[el], 0.0f, 0.0f, 0};
    const int baseA = 2 * c * Cv, baseB = baseA + Cv;
    const int n = g.n_seg;
    if (c == 0) col[0] = spool;  // wire cell 0 is held at the spool temperature
    const int own_last = (n >= 2) ? ((n - 1 >= baseA && n - 1 < baseA + Cv) ? 1 : (n - 1 >= baseB && n - 1 < baseB + Cv) ? 2 : 0) : 0;
    const bool has_inner = n >= 3;
    const LanesPkGeom pkg{c, Cv, R, n, baseA, baseB, own_last,
                          (uint32_t)g.az_start, g.az_end > g.az_start ? (uint32_t)(g.az_end - g.az_start) : 0u,
                          (uint32_t)g.cb, g.ct >= g.cb ? (uint32_t)(g.ct - g.cb + 1) : 0u,
                          has_inner ? (uint32_t)(n - 3) : 0u, has_inner};

    WEDM_SV_LOOP_START();
    for (int it = 0; it < k.n_substeps; ++it) {
        const int slot = it & 1;
        { WEDM_SV_WAIT_BEGIN(); sv_wait(&box->cf_seq, (uint32_t)it + 1u); WEDM_SV_WAIT_END(); }
        WEDM_SV_PHASE_START();
        const int32_t fl = box->flags[slot][el];
        if (fl & SV_STOP) break;  // (block-wide: every lane reads it)
        const Coef cf{box->jf[slot][el], box->q[slot][el], (fl & SV_JOULE) ? 1 : 0, box->pidx[slot][el]};
        ps.conv_base = box->conv_base[slot][el];
        ps.conv_zone = box->conv_zone[slot][el];
        ps.adv_on = (fl & SV_ADV) ? 1 : 0;
        const bool keep = live && !(fl & SV_DONE);
        WEDM_SV_PHASE(pa);
        float tmax = lanes_pk_step<L, NT>(col, pkg, keep, g, cf, ps, spool, tref, alpha, tdiel);
        WEDM_SV_PHASE(pb);
        tmax = max_over_env_lanes<L>(tmax);
        if (c == 0) box->tmax[slot][el] = tmax;
        asm volatile("" ::: "memory");
        if ((tid & 63) == 0) box->tm_seq[wave] = (uint32_t)it + 1u;
        WEDM_SV_PHASE(pc);
    }
    WEDM_SV_LOOP_END();
    sv_stamps_out(svs, stamp_row);
    __syncthreads();  // (B)
    copy_wire_nt<L, NT, false>(cold->s.T, stride, e0, k.num_envs, nmax, tid, lds, wire_slot);
}
