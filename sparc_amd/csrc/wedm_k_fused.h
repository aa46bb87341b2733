// wedm_k_fused.h — wedm_step_fused<L>: uniform geometry, L lanes per environment, wire chunks in LDS, wave-uniform tile table.
//
// Included by wedm_kernels.hip (one translation unit per WEDM_PART; see the bottom of that file).
#pragma once

// ===================================================== fused kernel, L lanes / env

// One interior cell (1 <= i <= n-2), float32 op for op as wire.py:91-120 evaluates it.
// The advection term is always applied: adv == 0 in lanes where the reference skips it
// (d + 0*(..) == d), which keeps the loop free of a per-lane branch.
template <bool JOULE>
__device__ __forceinline__ float interior_cell(float tm1, float tc, float tp1, float k, float tuf, float conv,
                                               float tdiel, float adv, float jfe, float alpha, float tref) {
    float a = sub_twice(tm1, tc);  // T[i-1] - 2*T[i], one rounding
    float d = k * (a + tp1);
    if (JOULE) {
        float rho_T = 1.0f + alpha * (tc - tref);
        d = d + jfe * rho_T;  // jfe == 0 in lanes outside the contacts: d + 0 == d
    }
    d = d - conv * (tc - tdiel);
    d = d + adv * (tm1 - tc);
    return tc + d * tuf;
}

// FROZEN_OK: see wedm_step_packed.  N1: the instantiation for tile tables with a one-change tile that is a boundary tile in
// every microsecond (4 096 x 400 over 16 lanes: the end of the workpiece zone falls inside tile 2 of 4): +4.7 % there; the
// extra code costs tables without such a tile 1-1.5 %, so they run the instantiation without it.
// F64: wedm_params.stencil_mode 1 -- the stencil as Numba types wire.py:58-123 (float64 expressions rounded at each float32
// store), on the tile walk: every tile takes the boundary-tile code (per-cell coefficients, interior formula, end cells
// patched), which is exact for regular tiles too; no stage-major / packed form.  Instantiated with FROZEN_OK only.
template <int L, bool TRACE, bool FROZEN_OK = false, bool N1 = false, bool F64 = false>
__global__ void __launch_bounds__(256, WEDM_FUSED_MIN_BLOCKS) wedm_step_fused(const KArgs k) {
    constexpr bool kFrozenOk = FROZEN_OK;
    // (the N1 instantiation serves small batches with one wave per SIMD: 4 096 x 400 over 16 lanes)
    constexpr bool PREFETCH = N1 && !F64 && WEDM_PREFETCH_N1;
    const ColdRef cold = kernarg_cold();
    Hot hv = k.hot;
    pin_hot_in_vgprs(hv);  // 178 -> 225 VGPRs, SGPR spill traffic in the loop 111 -> 37 instructions: +8 %
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int EPB = 256 / L;  // environments per block
    const int tid = threadIdx.x;
    const int el = tid / L, c = tid % L;
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int64_t e = e0 + el;
    const bool live = e < k.num_envs;
    const WalkTable* __restrict__ wt = k.walk;
    const int C = wt->C;
    const int n = k.hot.n_seg;
    const int64_t stride = cold->s.stride;

    // ---- stage the block's EPB wire columns: 16-byte words of the quad-interleaved block -> LDS
    // wire cell i -> chunk i / C, cell i % C -> LDS [cell][256 lanes], lane = environment slot * L + chunk
    const auto wire_slot = [C](int i) { const int ci = i / C; return (i - ci * C) * 256 + ci; };
    copy_wire<L, true>(cold->s.T, stride, e0, k.num_envs, n, tid, lds, wire_slot);
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
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    StencilF64 f64c{0.0, 0.0, 0.0};
    if (F64) { const wedm_params* pp = cold->p; f64c = StencilF64{pp->temp_ref, pp->alpha_rho, pp->dielectric_temperature}; }
    // one cell by the full predicated formula / one interior cell with coefficients handed in, in the stencil's typing
    // (zone / contacts: whether the cell lies in the workpiece zone / between the contacts)
    auto cell_full = [&](int i, float tm, float tcc, float tp, const Coef& cf, const Persist& ps) -> float {
        if (F64) return stencil_cell_f64(i, n, tm, tcc, tp, g, cf, ps, f64c, s.h_base, s.h_zone);
        return stencil_cell(i, n, tm, tcc, tp, g, cf, ps, tref, alpha, tdiel);
    };
    auto cell_interior = [&](float tm, float tcc, float tp, bool zone, bool contacts, const Coef& cf, const Persist& ps,
                             float jf_lane) -> float {
        if (F64)
            return interior_cell_f64(tm, tcc, tp, g.k64, g.tuf64, (double)(zone ? s.h_zone : s.h_base) * g.a64, f64c.tdiel, ps.adv64,
                                     (contacts && cf.joule_on) ? cf.jf64 : 0.0, f64c.alpha, f64c.tref);
        return interior_cell<true>(tm, tcc, tp, g.k, g.tuf, zone ? ps.conv_zone : ps.conv_base, tdiel, ps.adv,
                                   contacts ? jf_lane : 0.0f, alpha, tref);
    };
    const int n_tiles = wt->n_tiles;
    // per-lane tile membership, gathered ONCE so that walking a tile reads nothing but LDS
    // (scalar loads share lgkmcnt with LDS and would drain the prefetch every tile):
    // bit t of zone_lo/joule_lo = flags of the tile's first cell, *_hi = flags of its last cell
    uint32_t zone_lo = 0u, joule_lo = 0u, zone_hi = 0u, joule_hi = 0u, kind_n = 0u, kind_s = 0u;
    uint32_t split_pack[3] = {0u, 0u, 0u};  // 4 bits per tile (WEDM_MAX_TILES <= 24)
    for (int t = 0; t < n_tiles; ++t) {
        const uint32_t lo = wt->zj[8 * t], hi = wt->zj[8 * t + 7], kd = wt->kind[t];
        split_pack[t >> 3] |= (wt->split[t] & 15u) << ((t & 7) * 4);
        zone_lo |= ((lo >> c) & 1u) << t;
        joule_lo |= ((lo >> (16 + c)) & 1u) << t;
        zone_hi |= ((hi >> c) & 1u) << t;
        joule_hi |= ((hi >> (16 + c)) & 1u) << t;
        kind_n |= (kd == TILE_N ? 1u : 0u) << t;
        kind_s |= (kd == TILE_S ? 1u : 0u) << t;
    }
    kind_n = F64 ? 0u : __builtin_amdgcn_readfirstlane(kind_n);  // (F64: every tile on the boundary-tile code)
    kind_s = __builtin_amdgcn_readfirstlane(kind_s);
    // tiles that take the regular code although they hold a wire end cell / a contact-flag change (see WalkTable)
    const uint32_t kind_ne = F64 ? 0u : __builtin_amdgcn_readfirstlane(wt->kind_ne_mask), kind_nj = F64 ? 0u : __builtin_amdgcn_readfirstlane(wt->kind_nj_mask);
    const uint32_t kind_n1 = (N1 && !F64) ? (__builtin_amdgcn_readfirstlane(wt->kind_n1_mask) & 0x7fffffffu) : 0u;
#pragma unroll
    for (int q = 0; q < 3; ++q) split_pack[q] = __builtin_amdgcn_readfirstlane(split_pack[q]);
    if (c == 0) col[0] = spool;  // wire cell 0 is held at the spool temperature (wire.py:83)
    // the lane that owns the wire's last cell (Neumann boundary, wire.py:95)
    const bool owns_last = (n >= 2) && (n - 1 >= cbase) && (n - 1 < cbase + C);
    const int t_last = (n - 1 - cbase) >> 3;  // the tile of that cell in the owning lane (its last position, where the tile is regular)
    // A chunk whose length is 1 or 2 cells over a multiple of 8 (400 segments: 25 cells over 16 lanes, 50 over 8) would
    // spend a whole tile on that tail, and a tile costs its dependent chain whatever its width (stamped: 811-843 cycles
    // for the 1- / 2-cell tile against 799-809 for a full regular one).  The tail cells are instead computed like the
    // patched cells: by the interior formula from OLD values before the walk (their chains overlap those of the plasma /
    // last cell), written after it; the walk covers the full tiles only.  Bits per tail cell q: zone, contacts,
    // interior, valid (this lane's chunk).
    const int tail = (!F64 && C > 8 && (C & 7) >= 1 && (C & 7) <= 2) ? (C & 7) : 0;
    uint32_t tail_bits = 0u;
    for (int q = 0; q < tail; ++q) {
        const uint32_t zj = wt->zj[C - tail + q], iv = wt->iv[C - tail + q];
        tail_bits |= (((zj >> c) & 1u) | (((zj >> (16 + c)) & 1u) << 1) | (((iv >> c) & 1u) << 2) | (((iv >> (16 + c)) & 1u) << 3)) << (4 * q);
    }

    WEDM_STAMP_DECL;
    const bool tracing = WEDM_TRACING(k);
    int trace_next = k.trace_next, trace_slot = k.trace_slot;
    (void)trace_next; (void)trace_slot;
    for (int it = 0; it < k.n_substeps; ++it) {
        if (__all(s.done) && !tracing) break;
        WEDM_STAMP(st0);
        Coef cf{0.0f, 0.0f, 0, -1};
        QuietTry qt;
        if (!quiet_prelude_t<WEDM_FUSED_DENSE>(hv, cold, g, e, gid, s, qt, cf) && !s.done) cf = scalar_prelude(hv, cold, g, e, gid, s, ps, c == 0, qt);
        WEDM_STAMP(st1);
        freeze_wire(s);

        // ---- halos: OLD neighbour values, read before any lane of this wave stores.  The right
        // halo goes into the chunk's extra LDS row C, so cell C-1 is walked like any other.
        const float halo_l = (c > 0) ? col[(C - 1) * 256 - 1] : spool;
        const float halo_r = (c < L - 1) ? col[1] : 0.0f;
        col[C * 256] = halo_r;

        // a wave with a negative plasma heat (or, without FROZEN_OK, with a frozen environment) walks every cell on the
        // predicated path; results are identical, only slower
        const bool frozen_wave = FROZEN_OK && __any(s.done);
        const bool all_slow = __any(cf.q < 0.0f) || (!FROZEN_OK && __any(s.done));
        const uint32_t slow_now = all_slow ? 0xffffffffu : kind_s;
        // regular tiles of THIS microsecond: a contact-flag change inside a tile only matters while current flows
        const uint32_t n_now = (kind_n | kind_ne | (__any(cf.joule_on && !s.done && cf.jf != 0.0f) ? 0u : kind_nj)) & ~(all_slow ? 0xffffffffu : 0u);

        // ---- patched cells: the plasma cell and the wire's last cell are computed with the
        // full predicated formula from OLD values now and written after the walk
        const bool owns_pl = !s.done && cf.pidx >= 1 && cf.pidx >= cbase && cf.pidx < cbase + C;
        float tpl = 0.0f, tlast = 0.0f;
        if (__any(owns_pl)) {
            if (owns_pl) {
                const int jp = cf.pidx - cbase;
                float tm = jp > 0 ? col[(jp - 1) * 256] : halo_l;
                if (cf.pidx == 1) tm = spool;
                const float tcc = col[jp * 256];
                const float tp = jp < C - 1 ? col[(jp + 1) * 256] : halo_r;
                tpl = cell_full(cf.pidx, tm, tcc, tp, cf, ps);
            }
        }
        if (owns_last && !s.done) {
            const int jl = n - 1 - cbase;
            float tm = jl > 0 ? col[(jl - 1) * 256] : halo_l;
            if (n - 1 == 1) tm = spool;
            tlast = cell_full(n - 1, tm, col[jl * 256], 0.0f, cf, ps);
        }

        // ---- tail cells (see `tail`): new values from OLD ones, now; not on the predicated path, whose last tile covers them
        const bool use_tail = tail != 0 && !all_slow;
        float tt0 = 0.0f, tt1 = 0.0f;
        if (use_tail) {
            const float jfl = (cf.joule_on && !s.done) ? cf.jf : 0.0f;
            const int j0 = C - tail;
            const float a0 = col[(j0 - 1) * 256], b0 = col[j0 * 256], c0 = col[(j0 + 1) * 256];  // row C holds the right halo
            tt0 = interior_cell<true>(a0, b0, c0, g.k, g.tuf, (tail_bits & 1u) ? ps.conv_zone : ps.conv_base, tdiel, ps.adv,
                                      (tail_bits & 2u) ? jfl : 0.0f, alpha, tref);
            if (tail == 2) {
                const float c1 = col[(j0 + 2) * 256];
                tt1 = interior_cell<true>(b0, c0, c1, g.k, g.tuf, (tail_bits & 16u) ? ps.conv_zone : ps.conv_base, tdiel, ps.adv,
                                          (tail_bits & 32u) ? jfl : 0.0f, alpha, tref);
            }
        }
        const int n_walk = use_tail ? n_tiles - 1 : n_tiles;

        float tmax = spool;
        float tm1 = halo_l;
        float tc = col[0];
#ifdef WEDM_ABL_NO_STENCIL
        asm volatile("" ::"v"(cf.jf), "v"(cf.q), "v"(cf.pidx), "v"(ps.conv_base), "v"(ps.conv_zone), "v"(tpl), "v"(tlast));
        if (false) {
#else
        {
#endif
            const float jf_lane = (cf.joule_on && !s.done) ? cf.jf : 0.0f;
            const bool joule_wave = __any(jf_lane != 0.0f);

            // tile t covers cells j = 8t..8t+7; cur[u] = OLD T[j+1+u]; `nxt` is loaded one tile ahead
            // CLAMP = false: all eight rows exist (j + 8 <= C), one base address + immediate offsets
            auto load8 = [&](auto clamp, float (&dst)[8], int j) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int row = j + 1 + u;
                    if (decltype(clamp)::value) row = row < C ? row : C;  // rows past the chunk are never used; row C is the halo
                    dst[u] = col[row * 256];
                }
            };
            auto tile = [&](auto frozen, int t, float (&cur)[8], float (&nxt)[8]) {
                constexpr bool FROZEN = decltype(frozen)::value;  // the copy for a wave with frozen lanes: they do not store
                const int j = 8 * t;
                // PREFETCH (a lone wave per SIMD: nothing else hides the LDS round trip): the NEXT tile's eight rows are
                // requested before this tile is computed -- rows this tile does not store (it stores j .. j + 7, they are
                // j + 9 .. j + 16), so they are still the old values the explicit scheme needs
                if (PREFETCH) { if (t + 1 < n_walk) load8(std::true_type{}, nxt, j + 8); }
                else load8(std::true_type{}, cur, j);  // (an unclamped variant for full tiles pays in the packed kernel only)
                const float conv_lo = ((zone_lo >> t) & 1u) ? ps.conv_zone : ps.conv_base;
                const float jfe_lo = ((joule_lo >> t) & 1u) ? jf_lane : 0.0f;
#ifdef WEDM_STAMPS_TILES
                WEDM_STAMP(tk0);
                // (diagnostic buckets: regular tiles, boundary tiles, and -- in the third -- one-change tiles of the N1
                // instantiation together with the predicated fallback)
                const int tkind = ((n_now >> t) & 1u) ? 0 : ((N1 && (((kind_n1 & ~slow_now) >> t) & 1u)) ? 2 : (!((slow_now >> t) & 1u) ? 1 : 2));
#endif
                if ((n_now >> t) & 1u) {
                    float old[10], tn[8], cv[8], jv[8];
                    old[0] = tm1; old[1] = tc;
#pragma unroll
                    for (int u = 0; u < 8; ++u) old[u + 2] = cur[u];
                    cv[0] = conv_lo; jv[0] = jfe_lo;
                    if (joule_wave && __any(jfe_lo != 0.0f))
                        tile8_staged<float, true, false>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    else
                        tile8_staged<float, false, false>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    // the wire's end cells, where a regular tile holds one (kind_ne / kind_nj): cell 0 stays at the spool
                    // temperature; the last cell is kept out of the maximum here and patched after the walk
                    tn[0] = (c == 0 && t == 0) ? spool : tn[0];
                    const float last_v = (owns_last && t == t_last) ? spool : tn[7];
                    if (!FROZEN || !s.done) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) col[(j + u) * 256] = tn[u];
                    }
                    float m0 = fmax_gt(tn[0], tn[1]), m1 = fmax_gt(tn[2], tn[3]);
                    m0 = fmax_gt(m0, fmax_gt(tn[4], tn[5]));
                    m1 = fmax_gt(m1, fmax_gt(tn[6], last_v));
                    tmax = fmax_gt(tmax, fmax_gt(m0, m1));
                    tm1 = cur[6];
                    tc = cur[7];
                } else if (N1 && (((kind_n1 & ~slow_now) >> t) & 1u)) {
                    // one flag change at `split`, nothing else irregular (end cells apart): stage-major with per-cell
                    // coefficients, stores and maximum as in a regular tile
                    const int split = (int)((split_pack[t >> 3] >> ((t & 7) * 4)) & 15u);
                    const float conv_hi = ((zone_hi >> t) & 1u) ? ps.conv_zone : ps.conv_base;
                    const float jfe_hi = ((joule_hi >> t) & 1u) ? jf_lane : 0.0f;
                    float old[10], tn[8], cv[8], jv[8];
                    old[0] = tm1; old[1] = tc;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        old[u + 2] = cur[u];
                        cv[u] = u < split ? conv_lo : conv_hi;
                        jv[u] = u < split ? jfe_lo : jfe_hi;
                    }
                    if (joule_wave) tile8_staged<float, true, true>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    else tile8_staged<float, false, true>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    tn[0] = (c == 0 && t == 0) ? spool : tn[0];
                    const float last_v = (owns_last && t == t_last) ? spool : tn[7];
                    if (!FROZEN || !s.done) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) col[(j + u) * 256] = tn[u];
                    }
                    float m0 = fmax_gt(tn[0], tn[1]), m1 = fmax_gt(tn[2], tn[3]);
                    m0 = fmax_gt(m0, fmax_gt(tn[4], tn[5]));
                    m1 = fmax_gt(m1, fmax_gt(tn[6], last_v));
                    tmax = fmax_gt(tmax, fmax_gt(m0, m1));
                    tm1 = cur[6];
                    tc = cur[7];
                } else if (!((slow_now >> t) & 1u)) {
                    // TILE_B: interior formula everywhere, one flag change at `split`, boundary and
                    // out-of-wire cells excluded from the max (they are patched / never read)
                    const int split = (int)((split_pack[t >> 3] >> ((t & 7) * 4)) & 15u);
                    const int cnt = (C - j) < 8 ? (C - j) : 8;
                    const float conv_hi = ((zone_hi >> t) & 1u) ? ps.conv_zone : ps.conv_base;
                    const float jfe_hi = ((joule_hi >> t) & 1u) ? jf_lane : 0.0f;
                    const uint32_t im1 = (uint32_t)(cbase + j - 1);  // (i - 1) of the tile's first cell
                    const uint32_t span = (uint32_t)(n - 3);         // interior <=> (i - 1) <= n - 3 (unsigned)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u < cnt) {
                            const float conv = u < split ? conv_lo : conv_hi;
                            const float jfe = u < split ? jfe_lo : jfe_hi;
                            float tn;
                            if (F64) tn = cell_interior(tm1, tc, cur[u], ((u < split ? zone_lo : zone_hi) >> t) & 1u,
                                                        ((u < split ? joule_lo : joule_hi) >> t) & 1u, cf, ps, jf_lane);
                            else tn = interior_cell<true>(tm1, tc, cur[u], g.k, g.tuf, conv, tdiel, ps.adv, jfe, alpha, tref);
                            if (!FROZEN || !s.done) col[(j + u) * 256] = tn;
                            const bool inter = (n >= 3) && (im1 + (uint32_t)u <= span);
                            tmax = inter ? fmax_gt(tmax, tn) : tmax;
                            tm1 = tc;
                            tc = cur[u];
                        }
                    }
                } else {
#pragma unroll 1
                    for (int u = 0; u < 8; ++u) {
                        const int jj = j + u;
                        const uint32_t zj = wt->zj[jj], iv = wt->iv[jj];
                        const bool zbit = (zj >> c) & 1u, jbit = (zj >> (16 + c)) & 1u;
                        const bool inter = ((iv >> c) & 1u) && !all_slow;
                        const bool valid = ((iv >> (16 + c)) & 1u) && !s.done;
                        const float conv = zbit ? ps.conv_zone : ps.conv_base;
                        const float jfe = jbit ? jf_lane : 0.0f;
                        const float tp1 = cur[0];
                        float tn = F64 ? cell_interior(tm1, tc, tp1, zbit, jbit, cf, ps, jf_lane)
                                       : interior_cell<true>(tm1, tc, tp1, g.k, g.tuf, conv, tdiel, ps.adv, jfe, alpha, tref);
                        if (!inter && valid) {  // boundary cells and irregular waves: predicated formula
                            const int i = cbase + jj;
                            tn = (i >= 1) ? cell_full(i, (i == 1) ? spool : tm1, tc, tp1, cf, ps) : spool;
                        }
                        if (valid) {
                            col[jj * 256] = tn;
                            tmax = fmax_gt(tmax, tn);
                        }
                        tm1 = tc;
                        tc = tp1;
                        // rotate the prefetch window (this fallback is rare; keep its code small)
                        float* w = const_cast<float*>(&cur[0]);
                        float first = w[0];
#pragma unroll
                        for (int q = 0; q < 7; ++q) w[q] = w[q + 1];
                        w[7] = first;
                    }
                }
#ifdef WEDM_STAMPS_TILES
                WEDM_STAMP(tk1);
                if (tkind == 0) { accN += tk1 - tk0; ++cntN; } else if (tkind == 1) { accB += tk1 - tk0; ++cntB; } else { accS += tk1 - tk0; ++cntS; }
#endif
            };
            float bufA[8];
            if (PREFETCH) {
                float bufB[8];
                load8(std::true_type{}, bufA, 0);
                if (!FROZEN_OK || !frozen_wave) {
                    for (int t = 0; t < n_walk; t += 2) {
                        tile(std::false_type{}, t, bufA, bufB);
                        if (t + 1 < n_walk) tile(std::false_type{}, t + 1, bufB, bufA);
                    }
                } else {
                    for (int t = 0; t < n_walk; t += 2) {
                        tile(std::true_type{}, t, bufA, bufB);
                        if (t + 1 < n_walk) tile(std::true_type{}, t + 1, bufB, bufA);
                    }
                }
            } else if (!FROZEN_OK || !frozen_wave) {
                for (int t = 0; t < n_walk; ++t) tile(std::false_type{}, t, bufA, bufA);
            } else {
                for (int t = 0; t < n_walk; ++t) tile(std::true_type{}, t, bufA, bufA);
            }
        }
        WEDM_STAMP(st2);
        // ---- patches (after every store of the walk): tail cells, then boundary condition, last cell, plasma cell
        if (use_tail && !s.done) {
            // (valid: the cell exists; interior: it counts for the maximum and is not the wire's last cell, which the
            // patch below writes)
            if (tail_bits & 4u) { col[(C - tail) * 256] = tt0; tmax = fmax_gt(tmax, tt0); }
            if (tail == 2 && (tail_bits & 64u)) { col[(C - 1) * 256] = tt1; tmax = fmax_gt(tmax, tt1); }
        }
        if (c == 0 && !s.done) col[0] = spool;
        if (owns_last && !s.done) {
            col[(n - 1 - cbase) * 256] = tlast;
            tmax = fmax_gt(tmax, tlast);
        }
        if (owns_pl) {
            col[(cf.pidx - cbase) * 256] = tpl;
            tmax = fmax_gt(tmax, tpl);
        }
#pragma unroll
        for (int m = 1; m < L; m <<= 1) tmax = fmax_gt(tmax, __shfl_xor(tmax, m));
        unfreeze_wire(hv, s);
        WEDM_STAMP(st3);
        if (!s.done) {
            scalar_epilogue(hv, s, tmax);
            if (s.ctrl) control_step_outputs(cold, e, s, c == 0);
        }
        WEDM_TRACE_POINT(k, it, e, s, c == 0,
                         for (int j = 0; j < C && cbase + j < n; ++j) tT[(int64_t)(cbase + j) * tcnt] = col[j * 256]);
        WEDM_STAMP(st4);
        WEDM_STAMP_ACC();
    }
    WEDM_STAMP_OUT();

    __syncthreads();
    copy_wire<L, false>(cold->s.T, stride, e0, k.num_envs, n, tid, lds, wire_slot);
    if (live && c == 0) {
        if (WEDM_REWARD_ON(cold)) {
            if (!frozen0) write_reward(cold, e, s);
            else cold->s.reward[e] = 0.0f;  // a frozen environment earns nothing (not the previous launch's reward)
        }
        store_time_hi(cold, e, s, (uint32_t)k.n_substeps * (uint32_t)k.hot.dt_us);
        store_env(cold, e, s);
    }
}


