// wedm_k_packed.h — wedm_step_packed<L>: the fused walk with two virtual chunks per lane advanced in float2 registers.
//
// Included by wedm_kernels.hip (one translation unit per WEDM_PART; see the bottom of that file).
#pragma once

// ============================================ packed fused kernel, L lanes / env, 2 cells / op
// Same walk as wedm_step_fused, but every lane owns TWO virtual chunks A and B of Cv cells and
// advances them together in one float2 register pair, so each v_pk_add_f32 / v_pk_mul_f32 does
// two cells.  With only 1-2 waves per SIMD (the batch fixes the wave count) a wave is limited by
// its own in-order issue, one VALU per 4 cycles, while the SIMD pipe idles half the time: packing
// halves the instructions the wave has to issue.  Rows of A and B are interleaved in the lane's
// LDS column (row 2r = A[r], row 2r+1 = B[r]; rows 2Cv, 2Cv+1 hold the right halos), so a pair
// is one ds_read2st64_b32 / ds_write2st64_b32.  The walk table is the one built for 2L chunks.

template <bool JOULE>
__device__ __forceinline__ f2 interior2(f2 tm1, f2 tc, f2 tp1, float k, float tuf, f2 conv, float tdiel, float adv,
                                        f2 jfe, float alpha, float tref) {
    f2 a = sub_twice(tm1, tc);
    f2 d = k * (a + tp1);
    if (JOULE) {
        f2 rho_T = 1.0f + alpha * (tc - tref);
        d = d + jfe * rho_T;
    }
    d = d - conv * (tc - tdiel);
    d = d + adv * (tm1 - tc);
    return tc + d * tuf;
}

// FROZEN_OK: the instantiation for handles with in-launch autoreset, i.e. batches in which environments terminate at
// different times and wait, frozen, for the next launch.  Without it a wave with a frozen lane walks every cell on the
// predicated path (~4 x slower: 3.65e9 instead of 1.36e10 env-steps/s on a batch that resets 17 % of its environments per
// launch); with it such a wave takes a second copy of the tile code in which the frozen lanes do not store.  A separate
// instantiation, because the mere presence of that copy costs the other waves 2 % (6 % when folded into one copy).
// EXTRA: the instantiation for tile tables that need them: one-change tiles on the stage-major code (see wedm_step_fused's
// N1) and a chunk's 1- or 2-cell tail computed with the patched cells (virtual chunks of 25 cells: 400 segments over 8 lanes).
template <int L, bool TRACE, bool FROZEN_OK = false, bool EXTRA = false>
__global__ void __launch_bounds__(256, WEDM_PACKED_MIN_BLOCKS) wedm_step_packed(const KArgs k) {
    constexpr bool kFrozenOk = FROZEN_OK;
    const ColdRef cold = kernarg_cold();
    Hot hv = k.hot;
    // the constants of the epilogue and of the quiet prelude: what fits in 256 VGPRs without a
    // spill (pinning all of them spills 10 VGPRs and is no faster); +11 % over none
    pin_mechanics_in_vgprs(hv);
    pin_quiet_in_vgprs(hv);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int EPB = 256 / L;
    const int tid = threadIdx.x;
    const int el = tid / L, c = tid % L;
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int64_t e = e0 + el;
    const bool live = e < k.num_envs;
    const WalkTable* __restrict__ wt = k.walk;  // built for 2L virtual chunks
    const int Cv = wt->C;
    const int R = 2 * Cv;  // data rows per lane; rows R and R+1 are the halo pair
    const int n = k.hot.n_seg;
    const int64_t stride = cold->s.stride;

    // ---- stage: wire cell i -> virtual chunk vc = i / Cv, cell r = i % Cv -> lane vc/2, row 2r + vc%2
    const auto wire_slot = [Cv](int i) { const int vc = i / Cv; return (2 * (i - vc * Cv) + (vc & 1)) * 256 + (vc >> 1); };
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
        for (int row = 0; row < R; ++row) col[row * 256] = k.hot.spool;
    }
    unfreeze_wire(k.hot, s);  // keep_stepping_terminated: the DONE row is `terminated` of the last step and freezes nothing
    const bool frozen0 = s.done;
    WEDM_REPORT_FROZEN(frozen0 && live);
    if (!s.done) {
        s.ipk = peak_current(cold, s.mode, e);
        init_persist(k.hot, cold, e, s, ps);
    }
    const uint32_t gid = k.hot.env_id_offset + (uint32_t)e;

    const int baseA = 2 * c * Cv, baseB = baseA + Cv;  // first wire cell of each virtual chunk
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    const int n_tiles = wt->n_tiles;
    // per-lane tile flags for both virtual chunks, gathered once (see wedm_step_fused)
    uint32_t zlA = 0u, zlB = 0u, jlA = 0u, jlB = 0u, zhA = 0u, zhB = 0u, jhA = 0u, jhB = 0u, kind_n = 0u, kind_s = 0u;
    uint32_t split_pack[3] = {0u, 0u, 0u};
    for (int t = 0; t < n_tiles; ++t) {
        const uint32_t lo = wt->zj[8 * t], hi = wt->zj[8 * t + 7], kd = wt->kind[t];
        split_pack[t >> 3] |= (wt->split[t] & 15u) << ((t & 7) * 4);
        zlA |= ((lo >> (2 * c)) & 1u) << t;      zlB |= ((lo >> (2 * c + 1)) & 1u) << t;
        jlA |= ((lo >> (16 + 2 * c)) & 1u) << t; jlB |= ((lo >> (17 + 2 * c)) & 1u) << t;
        zhA |= ((hi >> (2 * c)) & 1u) << t;      zhB |= ((hi >> (2 * c + 1)) & 1u) << t;
        jhA |= ((hi >> (16 + 2 * c)) & 1u) << t; jhB |= ((hi >> (17 + 2 * c)) & 1u) << t;
        kind_n |= (kd == TILE_N ? 1u : 0u) << t;
        kind_s |= (kd == TILE_S ? 1u : 0u) << t;
    }
    kind_n = __builtin_amdgcn_readfirstlane(kind_n);
    kind_s = __builtin_amdgcn_readfirstlane(kind_s);
    // tiles that take the regular code although they hold a wire end cell / a contact-flag change (see WalkTable)
    const uint32_t kind_ne = __builtin_amdgcn_readfirstlane(wt->kind_ne_mask), kind_nj = __builtin_amdgcn_readfirstlane(wt->kind_nj_mask);
    const uint32_t kind_n1 = EXTRA ? (__builtin_amdgcn_readfirstlane(wt->kind_n1_mask) & 0x7fffffffu) : 0u;
#pragma unroll
    for (int q = 0; q < 3; ++q) split_pack[q] = __builtin_amdgcn_readfirstlane(split_pack[q]);
    if (c == 0) col[0] = spool;  // wire cell 0 (row 0 of lane 0's chunk A) is held at the spool temperature

    // which of this lane's virtual chunks holds wire cell i (0: none, 1: A, 2: B)
    auto owner = [&](int i) -> int {
        if (i >= baseA && i < baseA + Cv) return 1;
        if (i >= baseB && i < baseB + Cv) return 2;
        return 0;
    };
    const int own_last = (n >= 2) ? owner(n - 1) : 0;
    // the tile of that cell: a regular tile holds it only as the last cell of chunk B (chunk A's would be followed by
    // cells past the wire's end in the same tile), and not necessarily in the chunk's LAST tile (a further, partial tile
    // of cells past the end may follow)
    const int t_last = (n - 1 - baseB) >> 3;
    // tail cells of the two virtual chunks (see wedm_step_fused): bits per tail cell q and chunk v at 4 (2 q + v):
    // zone, contacts, interior, valid
    const int tail = (EXTRA && Cv > 8 && (Cv & 7) >= 1 && (Cv & 7) <= 2) ? (Cv & 7) : 0;
    uint32_t tail_bits = 0u;
    for (int q = 0; q < tail; ++q) {
        const uint32_t zj = wt->zj[Cv - tail + q], iv = wt->iv[Cv - tail + q];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int vc = 2 * c + v;
            tail_bits |= (((zj >> vc) & 1u) | (((zj >> (16 + vc)) & 1u) << 1) | (((iv >> vc) & 1u) << 2) | (((iv >> (16 + vc)) & 1u) << 3)) << (4 * (2 * q + v));
        }
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
        const bool was_quiet = quiet_prelude_t<WEDM_PACKED_DENSE>(hv, cold, g, e, gid, s, qt, cf);
        if (!was_quiet && !s.done) cf = scalar_prelude(hv, cold, g, e, gid, s, ps, c == 0, qt);
        freeze_wire(s);
        WEDM_STAMP(st1);
#ifdef WEDM_STAMPS
        if (was_quiet) { accN += st1 - st0; ++cntN; } else { accB += st1 - st0; ++cntB; }  // quiet / general prelude
#endif

        // ---- halos (OLD values, read before any store of this step)
        const float halo_l = (c > 0) ? col[(R - 1) * 256 - 1] : spool;  // left neighbour lane's B[Cv-1]
        const float halo_r = (c < L - 1) ? col[1] : 0.0f;               // right neighbour lane's A[0]
        const float a_last = col[(R - 2) * 256];                        // own A[Cv-1]: left halo of B
        const float b_first = col[256];                                 // own B[0]: right halo of A
        col[R * 256] = b_first;
        col[(R + 1) * 256] = halo_r;

        // a wave with a negative plasma heat (or, without FROZEN_OK, with a frozen environment) walks every cell on the
        // predicated path; results are identical, only slower
        const bool frozen_wave = FROZEN_OK && __any(s.done);
        const bool all_slow = __any(cf.q < 0.0f) || (!FROZEN_OK && __any(s.done));
        const uint32_t slow_now = all_slow ? 0xffffffffu : kind_s;
        // regular tiles of THIS microsecond: a contact-flag change inside a tile only matters while current flows
        const uint32_t n_now = (kind_n | kind_ne | (__any(cf.joule_on && !s.done && cf.jf != 0.0f) ? 0u : kind_nj)) & ~(all_slow ? 0xffffffffu : 0u);

        // full predicated formula for one owned cell, from OLD values (patched cells)
        auto patch_value = [&](int i, int own) -> float {
            // (unconditional LDS reads from clamped rows, then selects: a conditional read made the compiler select
            // between an LDS and a private address and fall back to flat loads; the rows after the last pair are the
            // halo pair (b_first, halo_r), exactly what the last cell of A / B needs on its right)
            const int v = own - 1, r = i - (v ? baseB : baseA), row = 2 * r + v;
            const float left = col[(r > 0 ? row - 2 : row) * 256];
            float tm = r > 0 ? left : (v ? a_last : halo_l);
            if (i == 1) tm = spool;
            const float tp = col[(row + 2) * 256];
            return stencil_cell(i, n, tm, col[row * 256], tp, g, cf, ps, tref, alpha, tdiel);
        };
        const int own_pl = (!s.done && cf.pidx >= 1) ? owner(cf.pidx) : 0;
        float tpl = 0.0f, tlast = 0.0f;
        if (__any(own_pl != 0)) {
            if (own_pl) tpl = patch_value(cf.pidx, own_pl);
        }
        if (own_last && !s.done) tlast = patch_value(n - 1, own_last);

        // ---- tail cells: new values from OLD ones, now (not on the predicated path, whose last tile covers them)
        const bool use_tail = EXTRA && tail != 0 && !all_slow;
        float tt[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // [2 q + v]
        if (use_tail) {
            const float jfl = (cf.joule_on && !s.done) ? cf.jf : 0.0f;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q < tail) {
                    const int r = Cv - tail + q;
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const uint32_t b = tail_bits >> (4 * (2 * q + v));
                        // rows 2 Cv and 2 Cv + 1 hold the halo pair: the right neighbour of a chunk's last cell
                        tt[2 * q + v] = interior_cell<true>(col[(2 * (r - 1) + v) * 256], col[(2 * r + v) * 256], col[(2 * (r + 1) + v) * 256],
                                                            g.k, g.tuf, (b & 1u) ? ps.conv_zone : ps.conv_base, tdiel, ps.adv,
                                                            (b & 2u) ? jfl : 0.0f, alpha, tref);
                    }
                }
            }
        }
        const int n_walk = use_tail ? n_tiles - 1 : n_tiles;

        float tmax = spool;
        f2 tm1 = {halo_l, a_last};
        f2 tc = {col[0], col[256]};
#ifdef WEDM_ABL_NO_STENCIL
        asm volatile("" ::"v"(cf.jf), "v"(cf.q), "v"(cf.pidx), "v"(ps.conv_base), "v"(ps.conv_zone), "v"(tpl), "v"(tlast));
        if (false) {
#else
        {
#endif
            const float jf_lane = (cf.joule_on && !s.done) ? cf.jf : 0.0f;
            const bool joule_wave = __any(jf_lane != 0.0f);
            const float cz = ps.conv_zone, cb = ps.conv_base;

            // dst[u] = OLD (A[r0+1+u], B[r0+1+u]); CLAMP = false: all eight pairs exist (r0 + 8 <= Cv),
            // one base address + immediate ds_read2st64 offsets
            auto load8 = [&](auto clamp, f2 (&dst)[8], int r0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int p = r0 + 1 + u;
                    if (decltype(clamp)::value) p = p < Cv ? p : Cv;  // pair Cv is the halo pair; later pairs are never used
                    dst[u].x = col[(2 * p) * 256];
                    dst[u].y = col[(2 * p + 1) * 256];
                }
            };
            auto store2 = [&](int r, f2 v) {
                col[(2 * r) * 256] = v.x;
                col[(2 * r + 1) * 256] = v.y;
            };
            auto tile = [&](auto frozen, int t, f2 (&cur)[8], f2 (&nxt)[8]) {
                constexpr bool FROZEN = decltype(frozen)::value;  // the copy for a wave with frozen lanes: they do not store
                const int r0 = 8 * t;
                // One buffer only: the tile's eight "next" pairs are loaded at the tile's start.  A
                // second (prefetch) buffer cost 16 VGPRs, pushed the kernel into scratch spills
                // (236 B/lane, ~30 GB of L2 traffic per launch) and was 7 % slower; the other wave of
                // the SIMD covers the LDS latency instead.
                (void)nxt;
                if (r0 + 8 <= Cv) load8(std::false_type{}, cur, r0);
                else load8(std::true_type{}, cur, r0);
                const f2 conv_lo = {((zlA >> t) & 1u) ? cz : cb, ((zlB >> t) & 1u) ? cz : cb};
                const f2 jfe_lo = {((jlA >> t) & 1u) ? jf_lane : 0.0f, ((jlB >> t) & 1u) ? jf_lane : 0.0f};
                if ((n_now >> t) & 1u) {
                    f2 old[10], tn[8], cv[8], jv[8];
                    old[0] = tm1; old[1] = tc;
#pragma unroll
                    for (int u = 0; u < 8; ++u) old[u + 2] = cur[u];
                    cv[0] = conv_lo; jv[0] = jfe_lo;
                    if (joule_wave && __any(jfe_lo.x != 0.0f || jfe_lo.y != 0.0f))
                        tile8_staged<f2, true, false>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    else
                        tile8_staged<f2, false, false>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    // the wire's end cells, where a regular tile holds one (kind_ne / kind_nj): cell 0 is the first cell
                    // of lane 0's chunk A and stays at the spool temperature; the last cell is the last cell of the last
                    // lane's chunk B: out of the maximum here, patched after the walk
                    tn[0].x = (c == 0 && t == 0) ? spool : tn[0].x;
                    const float last_y = (own_last == 2 && t == t_last) ? spool : tn[7].y;
                    float m0 = fmax_gt(tn[0].x, tn[0].y), m1 = fmax_gt(tn[1].x, tn[1].y);
                    if (!FROZEN || !s.done) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) store2(r0 + u, tn[u]);
                    }
#pragma unroll
                    for (int u = 2; u < 6; u += 2) {
                        m0 = fmax_gt(m0, fmax_gt(tn[u].x, tn[u].y));
                        m1 = fmax_gt(m1, fmax_gt(tn[u + 1].x, tn[u + 1].y));
                    }
                    m0 = fmax_gt(m0, fmax_gt(tn[6].x, tn[6].y));
                    m1 = fmax_gt(m1, fmax_gt(tn[7].x, last_y));
                    tmax = fmax_gt(tmax, fmax_gt(m0, m1));
                    tm1 = cur[6];
                    tc = cur[7];
                } else if (EXTRA && (((kind_n1 & ~slow_now) >> t) & 1u)) {
                    // one flag change at `split`, nothing else irregular (end cells apart): per-cell coefficients, stores
                    // and maximum as in a regular tile
                    const int split = (int)((split_pack[t >> 3] >> ((t & 7) * 4)) & 15u);
                    const f2 conv_hi = {((zhA >> t) & 1u) ? cz : cb, ((zhB >> t) & 1u) ? cz : cb};
                    const f2 jfe_hi = {((jhA >> t) & 1u) ? jf_lane : 0.0f, ((jhB >> t) & 1u) ? jf_lane : 0.0f};
                    f2 old[10], tn[8], cv[8], jv[8];
                    old[0] = tm1; old[1] = tc;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        old[u + 2] = cur[u];
                        cv[u] = u < split ? conv_lo : conv_hi;
                        jv[u] = u < split ? jfe_lo : jfe_hi;
                    }
                    if (joule_wave) tile8_staged<f2, true, true>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    else tile8_staged<f2, false, true>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    tn[0].x = (c == 0 && t == 0) ? spool : tn[0].x;
                    const float last_y = (own_last == 2 && t == t_last) ? spool : tn[7].y;
                    if (!FROZEN || !s.done) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) store2(r0 + u, tn[u]);
                    }
                    float m0 = fmax_gt(tn[0].x, tn[0].y), m1 = fmax_gt(tn[1].x, tn[1].y);
#pragma unroll
                    for (int u = 2; u < 6; u += 2) {
                        m0 = fmax_gt(m0, fmax_gt(tn[u].x, tn[u].y));
                        m1 = fmax_gt(m1, fmax_gt(tn[u + 1].x, tn[u + 1].y));
                    }
                    m0 = fmax_gt(m0, fmax_gt(tn[6].x, tn[6].y));
                    m1 = fmax_gt(m1, fmax_gt(tn[7].x, last_y));
                    tmax = fmax_gt(tmax, fmax_gt(m0, m1));
                    tm1 = cur[6];
                    tc = cur[7];
                } else if (!((slow_now >> t) & 1u)) {
                    // TILE_B: interior formula everywhere, one flag change at `split`; boundary and
                    // out-of-wire cells stay out of the max (patched afterwards / never read)
                    const int split = (int)((split_pack[t >> 3] >> ((t & 7) * 4)) & 15u);
                    const int cnt = (Cv - r0) < 8 ? (Cv - r0) : 8;
                    const f2 conv_hi = {((zhA >> t) & 1u) ? cz : cb, ((zhB >> t) & 1u) ? cz : cb};
                    const f2 jfe_hi = {((jhA >> t) & 1u) ? jf_lane : 0.0f, ((jhB >> t) & 1u) ? jf_lane : 0.0f};
                    const uint32_t imA = (uint32_t)(baseA + r0 - 1), imB = (uint32_t)(baseB + r0 - 1);
                    const uint32_t span = (uint32_t)(n - 3);
                    f2 old[10], tn[8], cv[8], jv[8];
                    old[0] = tm1; old[1] = tc;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        old[u + 2] = cur[u];
                        cv[u] = u < split ? conv_lo : conv_hi;
                        jv[u] = u < split ? jfe_lo : jfe_hi;
                    }
                    if (joule_wave) tile8_staged<f2, true, true>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    else tile8_staged<f2, false, true>(old, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u < cnt) {
                            if (!FROZEN || !s.done) store2(r0 + u, tn[u]);
                            const bool inA = (n >= 3) && (imA + (uint32_t)u <= span);
                            const bool inB = (n >= 3) && (imB + (uint32_t)u <= span);
                            tmax = inA ? fmax_gt(tmax, tn[u].x) : tmax;
                            tmax = inB ? fmax_gt(tmax, tn[u].y) : tmax;
                        }
                    }
                    // window after the tile: the last REAL pair of the chunk is what the next tile
                    // (if any) needs; a short tile is always the last one, so only full tiles matter
                    tm1 = cur[6];
                    tc = cur[7];
                } else {
                    // TILE_S: per-cell predicated fallback for both components (rare)
#pragma unroll 1
                    for (int u = 0; u < 8; ++u) {
                        const int r = r0 + u;
                        const uint32_t zj = wt->zj[r], iv = wt->iv[r];
                        const f2 tp1 = cur[0];
                        f2 tn;
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            const int vcid = 2 * c + v;
                            const bool zbit = (zj >> vcid) & 1u, jbit = (zj >> (16 + vcid)) & 1u;
                            const bool inter = ((iv >> vcid) & 1u) && !all_slow;
                            const bool valid = ((iv >> (16 + vcid)) & 1u) && !s.done;
                            const float conv = zbit ? cz : cb, jfe = jbit ? jf_lane : 0.0f;
                            const float m = v ? tm1.y : tm1.x, cc = v ? tc.y : tc.x, pp = v ? tp1.y : tp1.x;
                            float x = interior_cell<true>(m, cc, pp, g.k, g.tuf, conv, tdiel, ps.adv, jfe, alpha, tref);
                            if (!inter && valid) {
                                const int i = (v ? baseB : baseA) + r;
                                x = (i >= 1) ? stencil_cell(i, n, (i == 1) ? spool : m, cc, pp, g, cf, ps, tref, alpha, tdiel) : spool;
                            }
                            if (valid) {
                                col[(2 * r + v) * 256] = x;
                                tmax = fmax_gt(tmax, x);
                            }
                            if (v) tn.y = x; else tn.x = x;
                        }
                        tm1 = tc;
                        tc = tp1;
                        f2 first = cur[0];
#pragma unroll
                        for (int q = 0; q < 7; ++q) cur[q] = cur[q + 1];
                        cur[7] = first;
                    }
                }
            };
            f2 bufA[8];
            if (!FROZEN_OK || !frozen_wave) {
                for (int t = 0; t < n_walk; ++t) tile(std::false_type{}, t, bufA, bufA);
            } else {
                for (int t = 0; t < n_walk; ++t) tile(std::true_type{}, t, bufA, bufA);
            }
        }
        WEDM_STAMP(st2);
        // ---- patches (after every store of the walk): tail cells, then boundary condition, last cell, plasma cell
        if (use_tail && !s.done) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q < tail) {
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        if ((tail_bits >> (4 * (2 * q + v))) & 4u) {  // interior: exists, counts, and is not the wire's last cell
                            col[(2 * (Cv - tail + q) + v) * 256] = tt[2 * q + v];
                            tmax = fmax_gt(tmax, tt[2 * q + v]);
                        }
                    }
                }
            }
        }
        if (c == 0 && !s.done) col[0] = spool;
        if (own_last && !s.done) {
            const int v = own_last - 1;
            col[(2 * (n - 1 - (v ? baseB : baseA)) + v) * 256] = tlast;
            tmax = fmax_gt(tmax, tlast);
        }
        if (own_pl) {
            const int v = own_pl - 1;
            col[(2 * (cf.pidx - (v ? baseB : baseA)) + v) * 256] = tpl;
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
                         for (int r = 0; r < Cv; ++r) {
                             if (baseA + r < n) tT[(int64_t)(baseA + r) * tcnt] = col[(2 * r) * 256];
                             if (baseB + r < n) tT[(int64_t)(baseB + r) * tcnt] = col[(2 * r + 1) * 256];
                         });
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


