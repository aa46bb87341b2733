// wedm_k_stream.h — wedm_step_stream<L>: single microseconds (the reference's step() cadence), uniform geometry.
//
// Included by wedm_kernels.hip (one translation unit per WEDM_PART; see the bottom of that file).
#pragma once

// ===================================================== stream kernel (1 us / launch, uniform geometry)
// The reference's own cadence: ONE microsecond per launch, so every byte of state and wire crosses HBM
// once per launch and the roofline really is HBM.  Measured on the MI355X (tools/microbench/rowstream.hip): a
// bare read-modify-write of the 128 x 65 536 wire block in this [segment][environment] layout takes 7.5 us
// (one dword per lane, 9 TB/s out of the Infinity Cache), but the split kernel needs 30 us, because every wave
// is a chain of dependent round trips — state loads, scalar prelude, barrier, batches of rows behind
// `s_waitcnt vmcnt(0)`, barrier, epilogue — at two waves per SIMD, in two rounds of blocks that move in lock
// step.  Stamped variants on the way here (tools/stamps_stream.py): requesting the rows by LDS-DMA costs ~200
// cycles of issue per `global_load_lds_dword` (13 900 cycles for 64 rows), a row-by-row write-back 125 cycles per
// row.  This kernel has ONE memory round trip for everything it reads and no barrier:
//   * L lanes of ONE wave share an environment, as in the fused kernels (scalar physics replicated, halos
//     from the neighbour lane's column, DPP max reduction);
//   * at its very first instructions every lane requests the peak-current table (one entry per lane), the state
//     rows a microsecond reads and then its whole chunk of the wire into registers (CMAX unconditional
//     `global_load_dword`s from clamped addresses: a count the compiler can see, so no conservative waits);
//     the launch's first prelude needs the state only and runs while the wire is still in flight (nothing it reads
//     is queued behind the wire rows: vector loads return in order), then the chunk is dropped into the lane's
//     LDS column and the tile walk of wedm_step_fused runs on it;
//   * in the launch's last microsecond the walk stores every regular and boundary tile straight to global memory;
//     what is left (irregular tiles, patched cells) goes out after the loop, 8 rows at a time; only the state
//     rows a microsecond can have changed are stored (store_env_after_*), and a wave whose steps were all quiet
//     skips the rows the quiet prelude cannot change.
// Every wave is its own pipeline, so the loads, arithmetic and stores of different waves overlap by themselves.
#define WEDM_LDS __attribute__((address_space(3)))
#define WEDM_GLOBAL __attribute__((address_space(1)))
// ONE: the instantiation for launches of exactly one microsecond (the host picks it; no loop over further microseconds,
// and a walk out of registers for the waves that can take it: rest_single below)
// F64 (with ONE only): stencil_mode 1 -- the register walk in Numba's typing of wire.py:58-123 (cell_f64 / rw_quad of
// wedm_common.h); a wave that cannot take the register walk (a frozen environment, a negative plasma heat, a tile with several
// flag changes) walks every cell of its chunk on the per-cell code in that typing.
template <int L, bool TRACE, int CMAX, bool ONE = false, bool F64 = false>
__global__ void __launch_bounds__(256, 2) wedm_step_stream(const KArgs k) {
    static_assert(!F64 || (ONE && !TRACE), "the float64 typing exists for the single-microsecond instantiation only");
    const ColdRef cold = kernarg_cold();
    Hot hv = k.hot;
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

    const int cbase = c * C;
    float* col = lds + tid;
    const int jn = (!live) ? 0 : ((C < n - cbase) ? C : n - cbase);  // cells of this chunk that exist (<= 0: none)
    WEDM_S2_STAMP_DECL;

    // (0) the peak-current table (ignition.py:98-113), entry `lane` in lane `lane`: the wave's first vector load, so that
    // the lookup by the latched mode further down is a cross-lane read of a register that arrived long ago instead of a
    // load queued behind the whole wire (vector loads return in order: the first prelude would wait for every row)
    WEDM_S2_STAMP(10);  // kernel arguments here
    const int tab_i = (tid & 63) <= WEDM_MAX_MODE ? (tid & 63) : WEDM_MAX_MODE;
    const double ipk_entry = cold->tb.mode_current[tab_i];
    // (the same for the crater tables a fresh spark looks up: material.py:98-138)
    const LaneTables ltab{cold->tb.crater_mean[tab_i], cold->tb.crater_std[tab_i], cold->tb.crater_depth[tab_i], cold->tb.crater_valid[tab_i]};
    // (1) the state rows a microsecond reads: requested first, so that the first prelude runs while the wire is in flight
    Env s;
    Geom g;
    Persist ps{0.0f, 0.0f, 0.0f, 0};
    load_geom(k.hot, cold, live ? e : 0, g);
    StencilF64 f64c{0.0, 0.0, 0.0};
    if (F64) { const auto pp = opaque_const(cold->p); f64c = StencilF64{pp->temp_ref, pp->alpha_rho, pp->dielectric_temperature}; }
    WEDM_S2_STAMP(11);  // geometry constants here (two dependent scalar loads)
    double h64[2] = {0.0, 0.0};  // convection coefficients as loaded; converted after the wire rows are requested
    // with an even number of lanes per environment the two lanes of a pair each request ONE row of a pair of rows
    constexpr bool PAIRED = !TRACE && (L % 2 == 0) && WEDM_STREAM_PAIRED_LOADS;
    PairRaw raw;
    if (live) {
        if (TRACE) load_env(cold, e, s);  // frozen environments are sampled too: every row
        else if (PAIRED) load_env_inputs_paired_issue(cold, e, (c & 1) != 0, !k.hot.disable_ignition, raw);
        else load_env_inputs(cold, e, s, !k.hot.disable_ignition, h64, k.hot.done_value == 0);
    } else {
        s.done = WEDM_DEAD_LANE; s.unwind = 0.0; s.h_base = 0.0f; s.h_zone = 0.0f;
    }
    // per-lane tile membership, gathered by the host (build_walk): requested with the rest
    const uint32_t zone_lo = wt->chunk_flags[c][0], joule_lo = wt->chunk_flags[c][1];
    const uint32_t zone_hi = wt->chunk_flags[c][2], joule_hi = wt->chunk_flags[c][3];
    WEDM_S2_STAMP(8);  // state rows requested
    // (2) the wire: the lane's whole chunk into registers, 16 bytes (four consecutive cells of the quad-interleaved block)
    // per load, 32-bit byte offsets from the (wave-uniform) base of T (the host checks that the block is below 4 GB): one
    // v_add per word instead of a 64-bit multiply-add.  The chunk starts on a word (the stream kernel's walk tables
    // round the chunk length up to a multiple of 4).  Words past the chunk repeat its last word (a lane without cells
    // reads word 0): every load is unconditional and from a valid address, so the compiler can count them and waits
    // for each word only where it is used.  CMAX / 4 loads where ABI v3's T[seg][env] needed CMAX.
    static_assert(CMAX % 4 == 0, "whole 16-byte words");
#ifndef WEDM_STREAM_NO_STATE_WAIT
    // The state rows land BEFORE the wire words are requested.  All 2 048 waves of a launch start together, and when a
    // wave queues its wire words right behind its state rows the memory system serves the chip's whole request stream
    // interleaved: a wave's state (23 MB chip-wide) then arrives only while the 33 MB of wire stream in, ~5 us after the
    // launch began, and its prelude -- which needs nothing but the state -- starts that late.  Waiting here costs one
    // short round trip (the state alone is back within ~1.5 us) and puts the first prelude, the general one of an
    // igniting wave included, underneath the arrival of the wire.
    if (!TRACE) {
        if (PAIRED) { pair_raw_loaded_here(raw); }
        else { env_loaded_here(s); asm volatile("" : "+v"(h64[0]), "+v"(h64[1])); }
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    WEDM_S2_STAMP(9);  // state rows landed
    const char* const Tb = (const char*)cold->s.T;
    const uint32_t rowb = (uint32_t)stride * 16u;                                                         // bytes per row of words
    const uint32_t off0 = (uint32_t)((jn > 0 ? (cbase >> 2) : 0) * stride + (live ? e : 0)) * 16u;       // this lane's first word
    f4v w4[CMAX / 4];
    {
        const int qmax = jn > 0 ? ((jn + 3) >> 2) - 1 : 0;
        uint32_t off = off0;
#pragma unroll
        for (int q = 0; q < CMAX / 4; ++q) {
            w4[q] = *(const f4v*)(Tb + off);
            off += (q < qmax) ? rowb : 0u;
        }
    }
    // nothing that USES a loaded state row may be scheduled above this point: the first such use (the compiler hoisted
    // the test of the DONE flag) made the wave wait for the state rows -- a whole memory round trip -- before it had
    // requested its wire rows
    __builtin_amdgcn_sched_barrier(0);
    if (PAIRED && live) load_env_inputs_paired_finish(raw, (c & 1) != 0, s, k.hot.done_value == 0, h64);
    if (!TRACE && live) { s.h_base = (float)h64[0]; s.h_zone = (float)h64[1]; }
    WEDM_S2_STAMP(0);  // everything requested
    // next-step autoreset (all L lanes of the environment agree)
    const bool reinit = live && s.done && WEDM_AUTORESET_SCALAR(cold);
    if (reinit) reinit_env(cold, e, s, c == 0);
    unfreeze_wire(k.hot, s);  // keep_stepping_terminated: the DONE row is `terminated` of the last step and freezes nothing
    const bool frozen0 = s.done;
    double wp0 = 0.0;  // workpiece position at the start of the launch (reward)
    if (WEDM_REWARD_ON_SCALAR(cold) && !frozen0) wp0 = s.wp;
    {
        const bool in_table = s.mode >= 1 && s.mode <= WEDM_MAX_MODE;
        const double from_table = __shfl(ipk_entry, in_table ? s.mode : 0, 64);  // every lane takes part
        if (!s.done) {
            s.ipk = s.mode == 0 ? 60.0 : from_table;
            if (s.mode != 0 && !in_table) s.ipk = peak_current(cold, s.mode, e);  // unknown mode, or None over a stale cache (-1): default_current (cold parameter)
            init_persist<true>(k.hot, cold, e, s, ps);
        }
    }
#ifndef WEDM_STREAM_NO_PIN
    pin_hot_in_vgprs(hv);
#endif
    const uint32_t gid = k.hot.env_id_offset + (uint32_t)e;

    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    const int n_tiles = wt->n_tiles;
    const uint32_t kind_n = __builtin_amdgcn_readfirstlane(wt->kind_n_mask), kind_s = __builtin_amdgcn_readfirstlane(wt->kind_s_mask);
    // tiles that take the regular code although they hold a wire end cell / a contact-flag change (see WalkTable)
    const uint32_t kind_ne = __builtin_amdgcn_readfirstlane(wt->kind_ne_mask), kind_nj = __builtin_amdgcn_readfirstlane(wt->kind_nj_mask);
    uint32_t split_pack[3];  // 4 bits per tile (WEDM_MAX_TILES <= 24)
#pragma unroll
    for (int q = 0; q < 3; ++q) split_pack[q] = __builtin_amdgcn_readfirstlane(wt->split_pack[q]);
    // the lane that owns the wire's last cell (Neumann boundary, wire.py:95)
    const bool owns_last = (n >= 2) && (n - 1 >= cbase) && (n - 1 < cbase + C);
    const int t_last = (n - 1 - cbase) >> 3;  // the tile of that cell in the owning lane (its last position, where the tile is regular)

    const bool tracing = WEDM_TRACING(k);
    int trace_next = k.trace_next, trace_slot = k.trace_slot;
    (void)trace_next; (void)trace_slot;
    const bool no_ragged = L * C == n && k.num_envs % EPB == 0;  // every cell of every lane of the launch exists
    const uint32_t offc = (uint32_t)((cbase >> 2) * stride + (live ? e : 0)) * 16u;  // the word of cell cbase of this environment (stores)
    // byte offset of chunk cell j from offc
    const auto cell_off = [rowb](int j) -> uint32_t { return (uint32_t)(j >> 2) * rowb + (uint32_t)(j & 3) * 4u; };
    bool quiet_only = true;
    int patch0 = -1, patch1 = -1;  // cells patched after the last walk (chunk-local), -1: none
    uint32_t stored = 0u;  // tiles of the last microsecond that went to global memory from the walk itself (wave-uniform)
    // one microsecond = prelude (state only) + the rest (wire walk, epilogue, trace point).  The launch's first
    // prelude runs BEFORE the chunk is dropped into LDS: the wire's rows are still in flight then.
    auto prelude = [&](Coef& cf) {
        QuietTry qt;
#if WEDM_STREAM_DENSE_QUIET
        // (the quiet line also carries sparks that ignited earlier and keep burning or end now: only ignitions, shorts and
        // control-step latches take the general path -- and the issue priority)
        if (!quiet_prelude_t<ONE>(hv, cold, g, e, gid, s, qt, cf)) {
#else
        if (!quiet_prelude(hv, g, gid, s, qt)) {
#endif
#ifndef WEDM_STREAM_NO_SETPRIO
            // A launch ends with its slowest wave, and the slowest waves are the ~2 % whose prelude is the general one (a lane
            // ignites: crater normal, a dozen float64 divisions).  Such a wave takes the issue priority over the other wave of
            // its SIMD, which is not on the launch's critical path, for the rest of its life.
            __builtin_amdgcn_s_setprio(3);
#endif
            quiet_only = false;
            // the crater-table entries of every lane's mode (None / unknown -> I1, material.py:104-113) and of I1, read across
            // lanes from the registers that hold the tables (every lane of the wave is here: the quiet test is wave-uniform)
            const int mm = (s.mode >= 1 && s.mode <= WEDM_MAX_MODE) ? s.mode : 1;
            const LaneTables mine{__shfl(ltab.mean, mm, 64), __shfl(ltab.sd, mm, 64), __shfl(ltab.depth, mm, 64), __shfl(ltab.valid, mm, 64)};
            const LaneTables one{__shfl(ltab.mean, 1, 64), __shfl(ltab.sd, 1, 64), __shfl(ltab.depth, 1, 64), __shfl(ltab.valid, 1, 64)};
            if (!s.done) cf = scalar_prelude<false, true>(hv, cold, g, e, gid, s, ps, c == 0, qt, &mine, &one);
        }
    };
    auto rest = [&](const int it, Coef& cf) {
        const bool last = ONE || it + 1 == k.n_substeps;
        freeze_wire(s);
        // ---- halos: OLD neighbour values, read before any lane of this wave stores.  The right
        // halo goes into the chunk's extra LDS row C, so cell C-1 is walked like any other.
        const float halo_l = (c > 0) ? col[(C - 1) * 256 - 1] : spool;
        const float halo_r = (c < L - 1) ? col[1] : 0.0f;
        col[C * 256] = halo_r;

        // a wave with a frozen environment (or a negative plasma heat) walks every cell on the
        // predicated path; results are identical, only slower
        const bool all_slow = __any(cf.q < 0.0f) || __any(s.done);
        const uint32_t slow_now = (all_slow || F64) ? 0xffffffffu : kind_s;  // (F64: this path is the rare one, every tile cell by cell)
        // regular tiles of THIS microsecond: a contact-flag change inside a tile only matters while current flows
        const uint32_t n_now = (kind_n | kind_ne | (__any(cf.joule_on && !s.done && cf.jf != 0.0f) ? 0u : kind_nj)) & ~((all_slow || F64) ? 0xffffffffu : 0u);

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
                tpl = rw_cell<F64>(cf.pidx, n, tm, tcc, tp, g, cf, ps, tref, alpha, tdiel, f64c, s.h_base, s.h_zone);
            }
        }
        if (owns_last && !s.done) {
            const int jl = n - 1 - cbase;
            float tm = jl > 0 ? col[(jl - 1) * 256] : halo_l;
            if (n - 1 == 1) tm = spool;
            tlast = rw_cell<F64>(n - 1, n, tm, col[jl * 256], 0.0f, g, cf, ps, tref, alpha, tdiel, f64c, s.h_base, s.h_zone);
        }

        float tmax = spool;
        float tm1 = halo_l;
        float tc = col[0];
        {
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
            auto tile = [&](int t, float (&cur)[8], float (&nxt)[8]) {
                const int j = 8 * t;
                (void)nxt;
                load8(std::true_type{}, cur, j);  // (an unclamped variant for full tiles pays in the packed kernel only)
                const float conv_lo = ((zone_lo >> t) & 1u) ? ps.conv_zone : ps.conv_base;
                const float jfe_lo = ((joule_lo >> t) & 1u) ? jf_lane : 0.0f;
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
#pragma unroll
                    for (int u = 0; u < 8; ++u) col[(j + u) * 256] = tn[u];
                    if (last) {  // the launch's last microsecond: the tile also goes straight to global memory, two words
                        char* const Tw = (char*)cold->s.T;
                        const uint32_t off = offc + (uint32_t)(j >> 2) * rowb;
                        *(f4v*)(Tw + off) = f4v{tn[0], tn[1], tn[2], tn[3]};
                        *(f4v*)(Tw + off + rowb) = f4v{tn[4], tn[5], tn[6], tn[7]};
                        stored |= 1u << t;
                    }
                    float m0 = fmax_gt(tn[0], tn[1]), m1 = fmax_gt(tn[2], tn[3]);
                    m0 = fmax_gt(m0, fmax_gt(tn[4], tn[5]));
                    m1 = fmax_gt(m1, fmax_gt(tn[6], last_v));
                    tmax = fmax_gt(tmax, fmax_gt(m0, m1));
                    tm1 = cur[6];
                    tc = cur[7];
                } else if (!F64 && !((slow_now >> t) & 1u)) {
                    // TILE_B: interior formula everywhere, one flag change at `split`, boundary and
                    // out-of-wire cells excluded from the max (they are patched / never read)
                    const int split = (int)((split_pack[t >> 3] >> ((t & 7) * 4)) & 15u);
                    const int cnt = (C - j) < 8 ? (C - j) : 8;
                    const float conv_hi = ((zone_hi >> t) & 1u) ? ps.conv_zone : ps.conv_base;
                    const float jfe_hi = ((joule_hi >> t) & 1u) ? jf_lane : 0.0f;
                    const uint32_t im1 = (uint32_t)(cbase + j - 1);  // (i - 1) of the tile's first cell
                    const uint32_t span = (uint32_t)(n - 3);         // interior <=> (i - 1) <= n - 3 (unsigned)
                    float tnv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u < cnt) {
                            const float conv = u < split ? conv_lo : conv_hi;
                            const float jfe = u < split ? jfe_lo : jfe_hi;
                            float tn = interior_cell<true>(tm1, tc, cur[u], g.k, g.tuf, conv, tdiel, ps.adv, jfe, alpha, tref);
                            col[(j + u) * 256] = tn;
                            tnv[u] = tn;
                            const bool inter = (n >= 3) && (im1 + (uint32_t)u <= span);
                            tmax = inter ? fmax_gt(tmax, tn) : tmax;
                            tm1 = tc;
                            tc = cur[u];
                        }
                    }
                    if (last) {
                        // the launch's last microsecond: every cell of the tile that exists, except wire cell 0
                        // (spool temperature, never rewritten), goes straight to global memory; the cells patched
                        // after the walk are stored again behind these (same lane, same address: in order)
                        char* const Tw = (char*)cold->s.T;
                        const uint32_t offt = offc + (uint32_t)(j >> 2) * rowb;
                        if (no_ragged && cnt == 8) {  // every cell of every lane exists: two unconditional 16-byte stores
                            tnv[0] = (im1 == 0xffffffffu) ? spool : tnv[0];  // wire cell 0
                            *(f4v*)(Tw + offt) = f4v{tnv[0], tnv[1], tnv[2], tnv[3]};
                            *(f4v*)(Tw + offt + rowb) = f4v{tnv[4], tnv[5], tnv[6], tnv[7]};
                        } else {
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (u < cnt && im1 + (uint32_t)u < (uint32_t)(n - 1)) *(float*)(Tw + offt + cell_off(u)) = tnv[u];
                        }
                        stored |= 1u << t;
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
                        float tn;
                        if (F64)
                            tn = cell_f64<true>(tm1, tc, tp1, g.k64, g.tuf64, (double)(zbit ? s.h_zone : s.h_base) * g.a64, f64c.tdiel, ps.adv64,
                                                (jbit && jf_lane != 0.0f) ? cf.jf64 : 0.0, f64c.alpha, f64c.tref);
                        else
                            tn = interior_cell<true>(tm1, tc, tp1, g.k, g.tuf, conv, tdiel, ps.adv, jfe, alpha, tref);
                        if (!inter && valid) {  // boundary cells and irregular waves: predicated formula
                            const int i = cbase + jj;
                            tn = (i >= 1) ? rw_cell<F64>(i, n, (i == 1) ? spool : tm1, tc, tp1, g, cf, ps, tref, alpha, tdiel, f64c, s.h_base, s.h_zone)
                                          : spool;
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
            };
            float bufA[8];
            for (int t = 0; t < n_tiles; ++t) tile(t, bufA, bufA);
        }
        WEDM_S2_STAMP(3);  // walk done
        // ---- patches (after every store of the walk): boundary condition, last cell, plasma cell
        patch0 = (owns_last && !s.done) ? n - 1 - cbase : -1;
        patch1 = owns_pl ? cf.pidx - cbase : -1;
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
        if (!s.done) {
            scalar_epilogue(hv, s, tmax);
            if (s.ctrl) control_step_outputs(cold, e, s, c == 0);
        }
        WEDM_TRACE_POINT(k, it, e, s, c == 0,
                         for (int j = 0; j < C && cbase + j < n; ++j) tT[(int64_t)(cbase + j) * tcnt] = col[j * 256]);
    };
    // A launch of ONE microsecond (the reference's cadence) whose wave has nothing frozen and no tile on the predicated
    // path never reads a NEW temperature again, so the walk runs out of the registers the wire was loaded into: no LDS
    // read, no LDS write of a result, two cells per packed operation (adjacent cells; the shifted neighbour pairs cost a
    // move each), every tile stored to global memory where it is computed.  The OLD chunk still goes to LDS -- one
    // 16-byte write per word -- for the few cells read by a DYNAMIC index: the halos and the neighbours of the patched
    // cells (plasma cell, last cell).
    constexpr bool REGWALK = ONE && !TRACE && CMAX <= 64 && WEDM_STREAM_REGWALK;
    auto rest_single = [&](Coef& cf) {
        if (__any(reinit)) {
#pragma unroll
            for (int q = 0; q < CMAX / 4; ++q) w4[q] = reinit ? f4v{spool, spool, spool, spool} : w4[q];
        }
        if (c == 0) w4[0][0] = spool;  // wire cell 0 is held at the spool temperature (wire.py:83)
        {
            // word q of lane l of this wave -> row 4 q + l / 16 of the wave's own 64 columns, at (l % 16) * 4: the floats a
            // wave touches are the ones of its columns in the [cell][lane] layout, so the other waves of the block may be
            // on either path
            typedef f4v __attribute__((may_alias)) f4v_any;  // (read back below as single floats)
            float* const mine = lds + ((tid >> 4) & 3) * 256 + (tid & ~63) + (tid & 15) * 4;
#pragma unroll
            for (int q = 0; q < CMAX / 4; ++q)
                if (4 * q < C) *(f4v_any*)(mine + q * 1024) = w4[q];
        }
        WEDM_S2_STAMP(1);  // wire in LDS
        // OLD value of cell j of the lane `d` lanes away (same wave: LDS operations of a wave complete in order)
        const auto old_at = [&](int j, int d) -> float {
            const int l = (tid & 63) + d;
            return lds[((j >> 2) * 4 + (l >> 4)) * 256 + (tid & ~63) + (l & 15) * 4 + (j & 3)];
        };
        const float halo_l = (c > 0) ? old_at(C - 1, -1) : spool;
        const float halo_r = (c < L - 1) ? old_at(0, 1) : 0.0f;
        // regular tiles of THIS microsecond: a contact-flag change inside a tile only matters while current flows
        const uint32_t n_now = kind_n | kind_ne | (__any(cf.joule_on && (F64 || cf.jf != 0.0f)) ? 0u : kind_nj);
        // ---- patched cells: full predicated formula from OLD values, stored after the walk
        const bool owns_pl = cf.pidx >= 1 && cf.pidx >= cbase && cf.pidx < cbase + C;
        float tpl = 0.0f, tlast = 0.0f;
        if (__any(owns_pl)) {
            if (owns_pl) {
                const int jp = cf.pidx - cbase;
                float tm = jp > 0 ? old_at(jp - 1, 0) : halo_l;
                if (cf.pidx == 1) tm = spool;
                const float tp = jp < C - 1 ? old_at(jp + 1, 0) : halo_r;
                tpl = rw_cell<F64>(cf.pidx, n, tm, old_at(jp, 0), tp, g, cf, ps, tref, alpha, tdiel, f64c, s.h_base, s.h_zone);
            }
        }
        if (owns_last) {
            const int jl = n - 1 - cbase;
            float tm = jl > 0 ? old_at(jl - 1, 0) : halo_l;
            if (n - 1 == 1) tm = spool;
            tlast = rw_cell<F64>(n - 1, n, tm, old_at(jl, 0), 0.0f, g, cf, ps, tref, alpha, tdiel, f64c, s.h_base, s.h_zone);
        }
        float tmax = spool;
        // (float64 typing: the Joule entry is a flag -- the factor is cf.jf64 -- and the convection entries are the float32 h_eff)
        const float jf_lane = cf.joule_on ? (F64 ? 1.0f : cf.jf) : 0.0f;
        const float czone = F64 ? s.h_zone : ps.conv_zone, cbase_cv = F64 ? s.h_base : ps.conv_base;
        const bool joule_wave = __any(jf_lane != 0.0f);
        char* const Tw = (char*)cold->s.T;
#pragma unroll
        for (int t = 0; t < CMAX / 8; ++t) {
            const int j = 8 * t;
            if (j < C) {
                // o[0..9]: OLD T of cells j-1 .. j+8 (a 4-cell last tile: its cells j+4.. do not exist and are not used)
                float o[10];
                o[0] = t == 0 ? halo_l : w4[t > 0 ? 2 * t - 1 : 0][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) { o[1 + u] = w4[2 * t][u]; o[5 + u] = w4[2 * t + 1][u]; }
                o[5] = (j + 4 == C) ? halo_r : o[5];
                o[9] = (2 * t + 2 < CMAX / 4 && j + 8 != C) ? w4[2 * t + 2 < CMAX / 4 ? 2 * t + 2 : 0][0] : halo_r;
                f2 tm[4], tc[4], tp[4], tn[4], cv[4], jv[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    tm[m] = f2{o[2 * m], o[2 * m + 1]};
                    tc[m] = f2{o[2 * m + 1], o[2 * m + 2]};
                    tp[m] = f2{o[2 * m + 2], o[2 * m + 3]};
                }
                const float conv_lo = ((zone_lo >> t) & 1u) ? czone : cbase_cv;
                const float jfe_lo = ((joule_lo >> t) & 1u) ? jf_lane : 0.0f;
                const uint32_t off = offc + (uint32_t)(j >> 2) * rowb;
                if ((n_now >> t) & 1u) {
                    cv[0] = f2{conv_lo, conv_lo}; jv[0] = f2{jfe_lo, jfe_lo};
                    if (joule_wave && __any(jfe_lo != 0.0f))
                        rw_quad<F64, true, false, WEDM_QUAD_STAGE_W>(tm, tc, tp, tn, g, cv, tdiel, ps, jv, alpha, tref, f64c, cf);
                    else
                        rw_quad<F64, false, false, WEDM_QUAD_STAGE_W>(tm, tc, tp, tn, g, cv, tdiel, ps, jv, alpha, tref, f64c, cf);
                    // the wire's end cells, where a regular tile holds one (kind_ne / kind_nj): cell 0 stays at the spool
                    // temperature; the last cell is kept out of the maximum here and patched after the walk
                    tn[0].x = (c == 0 && t == 0) ? spool : tn[0].x;
                    const float last_v = (owns_last && t == t_last) ? spool : tn[3].y;
                    *(f4v*)(Tw + off) = f4v{tn[0].x, tn[0].y, tn[1].x, tn[1].y};
                    *(f4v*)(Tw + off + rowb) = f4v{tn[2].x, tn[2].y, tn[3].x, tn[3].y};
                    float m0 = fmax_gt(tn[0].x, tn[0].y), m1 = fmax_gt(tn[1].x, tn[1].y);
                    m0 = fmax_gt(m0, fmax_gt(tn[2].x, tn[2].y));
                    m1 = fmax_gt(m1, fmax_gt(tn[3].x, last_v));
                    tmax = fmax_gt(tmax, fmax_gt(m0, m1));
                } else {
                    // TILE_B: interior formula everywhere, one flag change at `split`; boundary and out-of-wire cells
                    // stay out of the maximum (patched after the walk / never stored)
                    const int split = (int)((split_pack[t >> 3] >> ((t & 7) * 4)) & 15u);
                    const int cnt = (C - j) < 8 ? (C - j) : 8;
                    const float conv_hi = ((zone_hi >> t) & 1u) ? czone : cbase_cv;
                    const float jfe_hi = ((joule_hi >> t) & 1u) ? jf_lane : 0.0f;
                    const uint32_t im1 = (uint32_t)(cbase + j - 1);  // (i - 1) of the tile's first cell
                    const uint32_t span = (uint32_t)(n - 3);         // interior <=> (i - 1) <= n - 3 (unsigned)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        cv[m] = f2{2 * m < split ? conv_lo : conv_hi, 2 * m + 1 < split ? conv_lo : conv_hi};
                        jv[m] = f2{2 * m < split ? jfe_lo : jfe_hi, 2 * m + 1 < split ? jfe_lo : jfe_hi};
                    }
                    if (joule_wave) rw_quad<F64, true, true, WEDM_QUAD_STAGE_W>(tm, tc, tp, tn, g, cv, tdiel, ps, jv, alpha, tref, f64c, cf);
                    else rw_quad<F64, false, true, WEDM_QUAD_STAGE_W>(tm, tc, tp, tn, g, cv, tdiel, ps, jv, alpha, tref, f64c, cf);
                    float tnv[8];
#pragma unroll
                    for (int m = 0; m < 4; ++m) { tnv[2 * m] = tn[m].x; tnv[2 * m + 1] = tn[m].y; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool inter = u < cnt && (n >= 3) && (im1 + (uint32_t)u <= span);
                        tmax = inter ? fmax_gt(tmax, tnv[u]) : tmax;
                    }
                    if (no_ragged && cnt == 8) {  // every cell of every lane exists: two unconditional 16-byte stores
                        tnv[0] = (im1 == 0xffffffffu) ? spool : tnv[0];  // wire cell 0
                        *(f4v*)(Tw + off) = f4v{tnv[0], tnv[1], tnv[2], tnv[3]};
                        *(f4v*)(Tw + off + rowb) = f4v{tnv[4], tnv[5], tnv[6], tnv[7]};
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (u < cnt && im1 + (uint32_t)u < (uint32_t)(n - 1)) *(float*)(Tw + off + cell_off(u)) = tnv[u];
                    }
                }
            }
        }
        WEDM_S2_STAMP(3);  // walk done
        // ---- patches, behind the walk's stores (same lane, same address: in order): last cell, then plasma cell
        if (owns_last) {
            *(float*)(Tw + offc + cell_off(n - 1 - cbase)) = tlast;
            tmax = fmax_gt(tmax, tlast);
        }
        if (owns_pl) {
            *(float*)(Tw + offc + cell_off(cf.pidx - cbase)) = tpl;
            tmax = fmax_gt(tmax, tpl);
        }
#pragma unroll
        for (int m = 1; m < L; m <<= 1) tmax = fmax_gt(tmax, __shfl_xor(tmax, m));
        unfreeze_wire(hv, s);
        scalar_epilogue(hv, s, tmax);
        if (s.ctrl) control_step_outputs(cold, e, s, c == 0);
    };
    const bool idle = __all(s.done) && !tracing;  // nothing to advance and nothing to sample
    {
        Coef cf{0.0f, 0.0f, 0, -1};
        if (!idle) prelude(cf);
        WEDM_S2_STAMP(2);  // prelude done (first microsecond)
        bool single = false;
        // (a table with a tile of several flag changes stays on the LDS walk: the predicated per-cell code inside the
        // register walk -- tried before, after and instead of it -- spills the registers that hold the wire:
        // 4 096 x 400 over 16 lanes 19.4 instead of 14.4 us, and 29.7 instead of 20.5 us at 65 536 x 128, which has no such tile)
        if (REGWALK && !idle && kind_s == 0u) {
            freeze_wire(s);
            single = !__any(s.done) && !__any(cf.q < 0.0f);  // (a lane past the batch counts as frozen)
        }
        if (REGWALK && single) {
            // (a branch of its own down to the state stores: what only further microseconds need -- the prelude's pinned
            // constants above all -- is dead during the register walk)
            rest_single(cf);
            WEDM_S2_STAMP(4);
        } else {
            // (3) the chunk into the lane's LDS column (each word is waited for where it is written: one round trip in all)
#pragma unroll
            for (int j = 0; j < CMAX; ++j)
                if (j < C) col[j * 256] = reinit ? k.hot.spool : w4[j >> 2][j & 3];
            if (c == 0) col[0] = k.hot.spool;  // wire cell 0 is held at the spool temperature (wire.py:83)
            WEDM_S2_STAMP(1);  // wire in LDS
            if (!idle) rest(0, cf);
            for (int it = 1; !ONE && it < k.n_substeps && !idle; ++it) {
                if (__all(s.done) && !tracing) break;
                Coef cf{0.0f, 0.0f, 0, -1};
                prelude(cf);
                rest(it, cf);
            }

            WEDM_S2_STAMP(4);  // walk + epilogue done
            // ---- write-back of what the walk did not store itself (boundary / irregular tiles, and the cells patched
            // after the walk: wire cell 0, the last cell, the plasma cell), a tile of 8 rows at a time: 8 LDS reads in
            // flight, then 8 stores, fire and forget; the L lanes of an environment are in one wave: nothing to wait for
            if (!frozen0) {
                char* const Tw = (char*)cold->s.T;
                stored = __builtin_amdgcn_readfirstlane(stored);
#pragma unroll
                for (int t = 0; t < (CMAX + 7) / 8; ++t) {
                    if (8 * t < C && !((stored >> t) & 1u)) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = col[((8 * t + u < C) ? 8 * t + u : C - 1) * 256];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int j = 8 * t + 4 * h;
                            if (j + 3 < jn) {  // a whole word of cells that exist
                                *(f4v*)(Tw + offc + (uint32_t)(j >> 2) * rowb) = f4v{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
                            } else {  // the wire's last, partial word: the cells past the end are padding and keep their value
#pragma unroll
                                for (int u = 0; u < 4; ++u)
                                    if (j + u < jn) *(float*)(Tw + offc + cell_off(j + u)) = v[4 * h + u];
                            }
                        }
                    }
                }
                // cells patched after the walk inside a tile that was already stored
                if (patch0 >= 0 && patch0 < jn && ((stored >> (patch0 >> 3)) & 1u)) *(float*)(Tw + offc + cell_off(patch0)) = col[patch0 * 256];
                if (patch1 >= 0 && patch1 < jn && ((stored >> (patch1 >> 3)) & 1u)) *(float*)(Tw + offc + cell_off(patch1)) = col[patch1 * 256];
            }
        }
    }
    if (live && c == 0 && frozen0 && WEDM_REWARD_ON_SCALAR(cold)) cold->s.reward[e] = 0.0f;  // a frozen environment earns nothing
    if (live && c == 0 && !frozen0) {
        if (WEDM_REWARD_ON_SCALAR(cold)) {
            const double pen = opaque(cold->p)->reward_break_penalty;
            cold->s.reward[e] = (float)(s.wp - wp0) - (float)pen * (s.broken ? 1.0f : 0.0f);
        }
        // (two lanes of an environment storing one row each per instruction -- 13 vector stores instead of 25 -- changes
        // nothing: 20.5 us either way; what a launch's last stores cost is their landing, not their number)
        store_env_after_prelude(cold, e, s, quiet_only);
        store_time_hi(cold, e, s, (uint32_t)k.n_substeps * (uint32_t)k.hot.dt_us);
        store_env_after_epilogue(cold, e, s);
    }
    WEDM_S2_STAMP(5);     // stores issued
    WEDM_S2_STAMP_VM(6);  // stores landed
    WEDM_S2_STAMP_OUT();
}


