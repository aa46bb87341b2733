// wedm_k_regs.h — the register kernels: wedm_step_regs<CELLS, L> (the headline's kernel: wires of at most 128 segments in the
// registers of 1 or 2 lanes) and wedm_step_regs_wide<H, L> (4 / 8 / 16 lanes of one DPP row per environment, 32 cells each).
//
// Included by wedm_kernels.hip (one translation unit per WEDM_PART; see the bottom of that file).
#pragma once

// ============================================ register kernel: one environment per lane, the whole wire in VGPRs
// Wires of at most CELLS (128) segments, uniform geometry; the float32 stencil or (F64) its float64 typing; a TRACE instantiation.
// A lane owns ONE environment and keeps its whole wire in registers from the launch's first microsecond to its last: no
// LDS, no halo exchange, no barrier, and the float64 scalar physics runs once per environment (the LDS kernels run it in
// every lane that shares an environment: 2 at 65 536 x 128).  One wave per SIMD at a 512-register budget.
//   * The wire is held as H = CELLS / 2 packed pairs P[m] = (T[m], T[H + m]) -- the two virtual chunks of the packed LDS
//     kernel -- so the neighbour pairs of P[m] are P[m - 1] and P[m + 1]: no shifted copies.  The table is the one built
//     for two chunks of exactly H cells (build_walk(p, 2, t, H)).
//   * A tile is 8 pairs, updated in place (the OLD pair before the tile is carried along; the OLD T[H - 1] and T[H], the
//     two chunks' halos, are taken at the step's start).
//   * Per microsecond ONE wave-uniform mask says which tiles need more than the regular code without a Joule term: not
//     regular in this microsecond, current in some lane between the contacts, a lane's plasma cell, the wire's last cell.
//     Every other tile is 88 packed operations and a running maximum behind one scalar branch.  The general code of a
//     tile recomputes the odd cells with the predicated formula (compile-time cell index, uniform geometry: scalar
//     predicates), or every cell of a tile that is not regular.
//   * A terminated environment keeps its registers: the walk runs under the mask of the live lanes.
struct cv4 {  // one coefficient pair for the four pairs of a quad (quad_staged with per-cell operands)
    f2 v[4];
    __device__ __forceinline__ explicit cv4(f2 x) : v{x, x, x, x} {}
};

// max(a, b, c) in one instruction.  The compiler cannot see that the halves of a packed result are canonical and puts a
// v_max_f32 x, x in front of every maximum it builds from fmaxf(); for the finite temperatures of a wire the values agree.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}


#ifndef WEDM_REGS_DENSE
#define WEDM_REGS_DENSE WEDM_PACKED_DENSE  // the quiet line also carries sparks that keep burning or end (see WEDM_PACKED_DENSE)
#endif
#ifndef WEDM_REGS_SW2
#define WEDM_REGS_SW2 2
#endif
#ifndef WEDM_REGS_PIN2
#define WEDM_REGS_PIN2 1
#endif
// TRACE: the instantiation with the signal-trace point (a launch into which a sample falls: the reference's logger samples
// after every step, utils/logger.py:110-160); launches without a sample run the instantiation without it.
// F64: stencil_mode 1 -- the walk in the typing Numba gives wire.py:58-123 (cell_f64 above); everything else is the same kernel.
template <int CELLS, int L, bool TRACE = false, bool F64 = false>
__global__ void __launch_bounds__(256, L) wedm_step_regs(const KArgs k) {
    // L = 1: one environment per lane (H = 64 pairs, one wave per SIMD at a 512-register budget);
    // L = 2: two lanes per environment, each with half of the wire (H = 32 pairs, two waves per SIMD, the scalar physics
    //        in both lanes as in the LDS kernels; the halves' halos cross by DPP)
    constexpr int H = CELLS / (2 * L);  // pairs per lane: P[m] = (T[base + m], T[base + H + m])
    static_assert(L == 1 || L == 2, "one or two lanes per environment");
    static_assert(H % 8 == 0 && H / 8 <= 16, "whole tiles");
    constexpr int EPB = 256 / L;
    // pairs per stage of the packed walk: a wave that is alone on its SIMD needs the distance between dependent operations
    constexpr int SW = L == 1 ? 4 : WEDM_REGS_SW2;
    constexpr bool kF64 = F64;
    const ColdRef cold = kernarg_cold();
    Hot hv = k.hot;
    // every-step float64 constants in VGPRs: all of them with 512 registers, the epilogue's and the quiet prelude's with 256
    if (L == 1 || WEDM_REGS_PIN2 == 2) {
        pin_hot_in_vgprs(hv);
    } else if (WEDM_REGS_PIN2 == 1) {
        pin_mechanics_in_vgprs(hv);
        pin_quiet_in_vgprs(hv);
    }
    const int tid = threadIdx.x;
    const int c = tid % L;  // this lane's part of the wire
    const int64_t e = (int64_t)blockIdx.x * EPB + tid / L;
    const bool live = e < k.num_envs;
    const bool writer = c == 0;
    const WalkTable* __restrict__ wt = k.walk;  // 2 L chunks of H cells
    const int n = k.hot.n_seg;
    const int64_t stride = cold->s.stride;
    const int base = c * 2 * H;  // this lane's first cell

    Env s;
    Geom g;
    Persist ps{0.0f, 0.0f, 0.0f, 0};
    load_geom(k.hot, cold, live ? e : 0, g);
    StencilF64 f64c{0.0, 0.0, 0.0};
    if (F64) {  // (uniform geometry: the float64 constants of the typing straight from the parameter block, wave-uniform)
        const wedm_params* pp = cold->p;
        f64c = StencilF64{pp->temp_ref, pp->alpha_rho, pp->dielectric_temperature};
        g.k64 = pp->k_cond; g.tuf64 = pp->tuf; g.a64 = pp->a_surf;
    }
    if (live) load_env(cold, e, s);
    else { s.done = WEDM_DEAD_LANE; s.unwind = 0.0; s.h_base = 0.0f; s.h_zone = 0.0f; }
    // the wire: word q = cells 4 q .. 4 q + 3 of this environment, 16 bytes per lane
    const int nq = (n + 3) >> 2;
    float* const Te = cold->s.T + (live ? e : 0) * 4;
    const int q0 = base / 4;  // this lane's first word
    f2 P[H];
#pragma unroll
    for (int q = 0; q < H / 4; ++q) {
        const f4v a = (q0 + q < nq) ? *(const f4v*)(Te + (int64_t)(q0 + q) * stride * 4) : f4v{0.0f, 0.0f, 0.0f, 0.0f};
        const f4v b = (q0 + H / 4 + q < nq) ? *(const f4v*)(Te + (int64_t)(q0 + H / 4 + q) * stride * 4) : f4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 4; ++u) P[4 * q + u] = f2{a[u], b[u]};
    }
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    const bool reinit = live && s.done && WEDM_AUTORESET(cold);  // next-step autoreset (all lanes of the environment agree)
    if (reinit) reinit_env(cold, e, s, writer);
    if (__any(reinit)) {
#pragma unroll
        for (int m = 0; m < H; ++m) P[m] = reinit ? f2{spool, spool} : P[m];
    }
    if (c == 0) P[0].x = spool;  // wire cell 0 is held at the spool temperature (wire.py:83)
    unfreeze_wire(k.hot, s);  // keep_stepping_terminated: the DONE row is `terminated` of the last step and freezes nothing
    const bool frozen0 = s.done;
    if (!s.done) {
        s.ipk = peak_current(cold, s.mode, e);
        init_persist(k.hot, cold, e, s, ps);
    }
    const uint32_t gid = k.hot.env_id_offset + (uint32_t)e;

    // tile flags of this lane's two chunks (bit t: the tile's first cell lies in the workpiece zone / between the contacts);
    // wave-uniform with one lane per environment
    const int n_tiles = wt->n_tiles;
    uint32_t zoneA = 0u, zoneB = 0u, jouleA = 0u, jouleB = 0u, joule_any = 0u;
    for (int t = 0; t < n_tiles; ++t) {
        const uint32_t lo = wt->zj[8 * t];
        zoneA |= ((lo >> (2 * c)) & 1u) << t;       zoneB |= ((lo >> (2 * c + 1)) & 1u) << t;
        jouleA |= ((lo >> (16 + 2 * c)) & 1u) << t; jouleB |= ((lo >> (17 + 2 * c)) & 1u) << t;
        joule_any |= ((lo >> 16) != 0u ? 1u : 0u) << t;
    }
    if (L == 1) {
        zoneA = __builtin_amdgcn_readfirstlane(zoneA); zoneB = __builtin_amdgcn_readfirstlane(zoneB);
        jouleA = __builtin_amdgcn_readfirstlane(jouleA); jouleB = __builtin_amdgcn_readfirstlane(jouleB);
    }
    joule_any = __builtin_amdgcn_readfirstlane(joule_any);
    const uint32_t kind_n = __builtin_amdgcn_readfirstlane(wt->kind_n_mask);
    const uint32_t kind_ne = __builtin_amdgcn_readfirstlane(wt->kind_ne_mask), kind_nj = __builtin_amdgcn_readfirstlane(wt->kind_nj_mask);
    // the wire's last cell: where a regular tile holds it, it is the last cell of the LAST chunk's tile t_last (chunk B of
    // the environment's last lane)
    const int last_base = (2 * L - 1) * H;
    const uint32_t last_tile = (n > last_base) ? (1u << ((n - 1 - last_base) >> 3)) : 0u;
    const bool owns_last = c == L - 1;

    // the convection coefficient pair (chunk A, chunk B) of every tile: rebuilt where the general prelude may have refreshed
    // the lane's coefficients (the quiet one never does).  F64: the float32 h_eff entries themselves (quad_f64 forms (double)h * A)
    f2 convp[H / 8];
    auto build_conv = [&]() {
        const float cz = F64 ? s.h_zone : ps.conv_zone, cb = F64 ? s.h_base : ps.conv_base;
#pragma unroll
        for (int t = 0; t < H / 8; ++t)
            convp[t] = f2{((zoneA >> t) & 1u) ? cz : cb, ((zoneB >> t) & 1u) ? cz : cb};
    };
    build_conv();
    const bool tracing = WEDM_TRACING(k);
    int trace_next = k.trace_next, trace_slot = k.trace_slot;
    (void)trace_next; (void)trace_slot;

    for (int it = 0; it < k.n_substeps; ++it) {
        if (__all(s.done) && !tracing) break;  // (terminated environments keep being sampled: their frozen state)
        Coef cf{0.0f, 0.0f, 0, -1};
        QuietTry qt;
        const bool was_quiet = quiet_prelude_t<WEDM_REGS_DENSE>(hv, cold, g, e, gid, s, qt, cf);
        if (!was_quiet) {
            if (!s.done) cf = scalar_prelude(hv, cold, g, e, gid, s, ps, writer, qt);
            build_conv();
        }
        freeze_wire(s);
        const bool act = !s.done;
        float tmax = spool;
        // (what the rare code of a tile derives from these -- a lane mask per uniform predicate, one per tile or per cell
        // -- would otherwise be computed once before the loop and kept: a thousand scalar registers spilled into vector
        // lanes and read back on the hot path too.  Opaque per microsecond, the predicates are scalar compares where used.)
        if (L == 1) asm volatile("" : "+s"(zoneA), "+s"(zoneB), "+s"(jouleA), "+s"(jouleB));
        else asm volatile("" : "+v"(zoneA), "+v"(zoneB), "+v"(jouleA), "+v"(jouleB));
        Geom gw = g;  // (uniform geometry: the same in every lane)
        gw.n_seg = __builtin_amdgcn_readfirstlane(g.n_seg); gw.az_start = __builtin_amdgcn_readfirstlane(g.az_start);
        gw.az_end = __builtin_amdgcn_readfirstlane(g.az_end); gw.cb = __builtin_amdgcn_readfirstlane(g.cb);
        gw.ct = __builtin_amdgcn_readfirstlane(g.ct);
        asm volatile("" : "+s"(gw.n_seg), "+s"(gw.az_start), "+s"(gw.az_end), "+s"(gw.cb), "+s"(gw.ct));
        int nw = __builtin_amdgcn_readfirstlane(n);
        asm volatile("" : "+s"(nw));
        // the halos of this lane's two chunks, OLD values: T[base + H - 1] (left of chunk B) and T[base + H] (right of
        // chunk A) are the lane's own; across lanes: the left of chunk A is the previous lane's last cell, the right of
        // chunk B the next lane's first (every lane takes part in the exchange, frozen environments included)
        const float a_last = P[H - 1].x, b_first = P[0].y;
        float halo_l = spool, halo_r = 0.0f;
        if (L == 2) {
            // lane 0 needs lane 1's first cell (its P[0].x); lane 1 needs lane 0's last cell (its P[H - 1].y)
            const float give = c == 0 ? P[H - 1].y : P[0].x;
            const float got = __int_as_float(swap_with_neighbour(__float_as_int(give)));
            halo_l = c == 0 ? spool : got;
            halo_r = c == 0 ? got : 0.0f;
        }
        const float rw_h_base = s.h_base, rw_h_zone = s.h_zone;  // (the predicated float64-typed cell reads the entries themselves)
#include "wedm_regs_walk.inc"
        if (L == 2) tmax = fmax_gt(tmax, __int_as_float(swap_with_neighbour(__float_as_int(tmax))));
        unfreeze_wire(hv, s);
        if (!s.done) {
            scalar_epilogue(hv, s, tmax);
            if (s.ctrl) control_step_outputs(cold, e, s, writer);
        }
        WEDM_TRACE_POINT(k, it, e, s, writer,
                         // (unrolled: a register file has no dynamic index; two running pointers made opaque after every pair,
                         // or the 2 H addresses are all computed up front and kept: 244 spilled registers in the two-lane form)
                         float* pa = tT + (int64_t)base * tcnt; float* pb = pa + (int64_t)H * tcnt;
                         int na = n - base; int nb = na - H;   // cells of this lane's two chunks that exist
                         asm volatile("" : "+v"(na), "+v"(nb));   // (opaque: or the 2 H store predicates are made before the loop and kept)
                         _Pragma("unroll") for (int m = 0; m < H; ++m) {
                             if (m < na) *pa = P[m].x;
                             if (m < nb) *pb = P[m].y;
                             pa += tcnt; pb += tcnt;
                             asm volatile("" : "+v"(pa), "+v"(pb));
                         });
    }

    if (live) {
#pragma unroll
        for (int q = 0; q < 2 * H / 4; ++q) {
            const int m = (q % (H / 4)) * 4;
            const bool hi = q >= H / 4;
            const f4v w = hi ? f4v{P[m].y, P[m + 1].y, P[m + 2].y, P[m + 3].y} : f4v{P[m].x, P[m + 1].x, P[m + 2].x, P[m + 3].x};
            const int cell = base + 4 * q;
            if (cell + 3 < n) {
                *(f4v*)(Te + (int64_t)(q0 + q) * stride * 4) = w;
            } else {  // the wire's last, partial word: the cells past the end are padding and keep their value
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (cell + u < n) Te[(int64_t)(q0 + q) * stride * 4 + u] = w[u];
            }
        }
    }
    if (live && writer) {
        if (WEDM_REWARD_ON(cold)) {
            if (!frozen0) write_reward(cold, e, s);
            else cold->s.reward[e] = 0.0f;  // a frozen environment earns nothing (not the previous launch's reward)
        }
        store_time_hi(cold, e, s, (uint32_t)k.n_substeps * (uint32_t)k.hot.dt_us);
        store_env(cold, e, s);
    }
}


// ============================================ wide register kernel: long wires of a SMALL batch in registers
// Wires of up to 2 H L (512) segments, uniform geometry; the float32 stencil or (F64) its float64 typing.
// The case it is for is 4 096 x 400: a batch that gives the chip one wave per SIMD whatever the kernel, so a launch's
// time is the dependent chain of ONE wave per microsecond, and what shortens the chain is fewer cells per lane and no
// LDS round trip inside it.  L = 16 lanes -- one DPP row -- own an environment; a lane holds 2 H = 32 cells as H = 16
// packed pairs P[m] = (T[base + m], T[base + H + m]) (two virtual chunks, as in wedm_step_regs): two tiles per microsecond.
// What differs from wedm_step_regs:
//   * no walk table.  The wire need not fill the lanes: lane c's cells 32 c .. 32 c + 31 that lie past the wire's end are
//     PADDING -- loaded as zeros, advanced like interior cells (the packed operations compute both halves of a pair
//     anyway), never stored, kept out of the maximum by one select per half and tile, and never read by a real cell
//     (the wire's last cell takes the predicated formula, which has no right neighbour).
//   * zone and contact flags per CELL, from the geometry's indices, as registers: a convection coefficient pair per
//     pair of cells (rebuilt when the general prelude refreshes the coefficients) and a 0 / 1 Joule mask pair; a tile is
//     regular whatever flags change inside it, also the one the wire's end cuts (n_seg not a multiple of 8: its maximum
//     is taken cell by cell).  The few cells that are not interior cells (the last cell, plasma cells) are recomputed
//     by the predicated formula and replace the regular result before the maximum is taken: there is no per-cell
//     fallback walk at all, not even for a negative plasma heat.
//   * halos between the lanes of an environment by DPP row shifts, the maximum over them by DPP quad / row mirrors.
__device__ __forceinline__ float dpp_row_shr1(float old, float x) {  // lane i <- lane i - 1 of its row of 16; lane 0 keeps `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), 0x111, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_row_shl1(float old, float x) {  // lane i <- lane i + 1; lane 15 keeps `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), 0x101, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}

#ifndef WEDM_WIDE_MIN_BLOCKS
#define WEDM_WIDE_MIN_BLOCKS 1
#endif

#ifndef WEDM_WIDE_SW
#define WEDM_WIDE_SW 2  // pairs per stage (4 096 x 400: 1.883e9 with 2, 1.862e9 with 4)
#endif
#ifndef WEDM_WIDE_DENSE
#define WEDM_WIDE_DENSE WEDM_REGS_DENSE
#endif
#ifndef WEDM_WIDE_AUTO_MAX_LANES
#define WEDM_WIDE_AUTO_MAX_LANES 65536  // one block per CU: 4 096 environments x 16 lanes, 16 384 x 4
#endif
// CUT: the instantiation for wires whose end cuts a tile (n_seg not a multiple of 8); the code for that tile costs the
// regular path 2 - 3 % by its presence (registers), so the other wires run the instantiation without it.
// TRACE: the instantiation with the signal-trace point (a launch into which a sample falls); built on the CUT form.
// F64: stencil_mode 1 (the walk in Numba's typing of wire.py:58-123, as in wedm_step_regs)
// MINB: blocks per CU the register budget admits (2: the float64 typing's instantiation for batches of more than one wave per
// SIMD -- a lone wave issues a float64 operation every ~7 cycles, two share the pipe; 256 registers, nothing pinned)
template <int H, int L, bool CUT, bool TRACE = false, bool F64 = false, int MINB = WEDM_WIDE_MIN_BLOCKS>
__global__ void __launch_bounds__(256, MINB) wedm_step_regs_wide(const KArgs k) {
    static_assert(H % 8 == 0 && H <= 32, "whole tiles");
    static_assert(L == 4 || L == 8 || L == 16, "the lanes of an environment lie in one DPP row");
    constexpr int EPB = 256 / L;
    constexpr int SW = WEDM_WIDE_SW;
    const ColdRef cold = kernarg_cold();
    Hot hv = k.hot;
    // (at 256 registers pinned constants are spilled wire cells: 32 768 x 400 in the float64 typing 1.42e9 with every hot constant
    // pinned, 1.62e9 with the epilogue's and the quiet prelude's, 1.93e9 with none)
    if (MINB == 1) pin_hot_in_vgprs(hv);
    const int tid = threadIdx.x;
    const int c = tid % L;  // this lane's part of the wire
    const int64_t e = (int64_t)blockIdx.x * EPB + tid / L;
    const bool live = e < k.num_envs;
    const bool writer = c == 0;
    const int n = k.hot.n_seg;
    const int64_t stride = cold->s.stride;
    const int base = c * 2 * H;  // this lane's first cell

    Env s;
    Geom g;
    Persist ps{0.0f, 0.0f, 0.0f, 0};
    load_geom(k.hot, cold, live ? e : 0, g);
    StencilF64 f64c{0.0, 0.0, 0.0};
    if (F64) {  // (uniform geometry: the float64 constants of the typing straight from the parameter block)
        const wedm_params* pp = cold->p;
        f64c = StencilF64{pp->temp_ref, pp->alpha_rho, pp->dielectric_temperature};
        g.k64 = pp->k_cond; g.tuf64 = pp->tuf; g.a64 = pp->a_surf;
    }
    if (live) load_env(cold, e, s);
    else { s.done = WEDM_DEAD_LANE; s.unwind = 0.0; s.h_base = 0.0f; s.h_zone = 0.0f; }
    // the wire: word q = cells 4 q .. 4 q + 3 of this environment, 16 bytes per lane; words past the end: zeros (padding)
    const int nq = (n + 3) >> 2;
    float* const Te = cold->s.T + (live ? e : 0) * 4;
    const int q0 = base / 4;  // this lane's first word
    f2 P[H];
#pragma unroll
    for (int q = 0; q < H / 4; ++q) {
        const f4v a = (q0 + q < nq) ? *(const f4v*)(Te + (int64_t)(q0 + q) * stride * 4) : f4v{0.0f, 0.0f, 0.0f, 0.0f};
        const f4v b = (q0 + H / 4 + q < nq) ? *(const f4v*)(Te + (int64_t)(q0 + H / 4 + q) * stride * 4) : f4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 4; ++u) P[4 * q + u] = f2{a[u], b[u]};
    }
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    const bool reinit = live && s.done && WEDM_AUTORESET(cold);  // next-step autoreset (all lanes of the environment agree)
    if (reinit) reinit_env(cold, e, s, writer);
    if (__any(reinit)) {
#pragma unroll
        for (int m = 0; m < H; ++m) P[m] = reinit ? f2{spool, spool} : P[m];
    }
    if (c == 0) P[0].x = spool;  // wire cell 0 is held at the spool temperature (wire.py:83)
    unfreeze_wire(k.hot, s);  // keep_stepping_terminated: the DONE row is `terminated` of the last step and freezes nothing
    const bool frozen0 = s.done;
    if (!s.done) {
        s.ipk = peak_current(cold, s.mode, e);
        init_persist(k.hot, cold, e, s, ps);
    }
    const uint32_t gid = k.hot.env_id_offset + (uint32_t)e;

    // cells of this lane's two chunks that exist (0 .. H each), per-cell flags as bit masks (bit m: cell m of the chunk)
    const int nA = min(max(n - base, 0), H), nB = min(max(n - base - H, 0), H);
    uint32_t zoneA = 0u, zoneB = 0u, jouleA = 0u, jouleB = 0u;
#pragma unroll
    for (int m = 0; m < H; ++m) {
        const int ia = base + m, ib = base + H + m;
        zoneA |= (ia >= g.az_start && ia < g.az_end) ? (1u << m) : 0u;
        zoneB |= (ib >= g.az_start && ib < g.az_end) ? (1u << m) : 0u;
        jouleA |= (ia >= g.cb && ia <= g.ct) ? (1u << m) : 0u;
        jouleB |= (ib >= g.cb && ib <= g.ct) ? (1u << m) : 0u;
    }
    // the tile the wire's end cuts, if n_seg is not a multiple of 8 (wave-uniform: uniform geometry): regular code too, with
    // its maximum taken cell by cell over the cells that exist and the last cell patched where it lies
    uint32_t cut = 0u;
    if (CUT) {
#pragma unroll
        for (int t = 0; t < H / 8; ++t)
            cut |= __any((nA > 8 * t && nA < 8 * t + 8) || (nB > 8 * t && nB < 8 * t + 8)) ? (1u << t) : 0u;
        cut = __builtin_amdgcn_readfirstlane(cut);
    }
    // the wire's last cell: in a tile the end does not cut it is the last cell of its tile (n_seg a multiple of 8)
    const int ll = n - 1 - base;  // lane-local index of the last cell, if this lane holds it
    const bool owns_last = ll >= 0 && ll < 2 * H;
    const int lloc = (n - 1) & (2 * H - 1);  // the same index, wave-uniform
    const bool last_in_b = lloc >= H;
    const uint32_t last_tile = ((n & 7) == 0) ? (1u << ((lloc & (H - 1)) >> 3)) : 0u;
    // 0 / 1 Joule mask pairs and the convection coefficient pairs of this lane's cells
    f2 jm[H], convc[H];
#pragma unroll
    for (int m = 0; m < H; ++m) jm[m] = f2{((jouleA >> m) & 1u) ? 1.0f : 0.0f, ((jouleB >> m) & 1u) ? 1.0f : 0.0f};
    auto build_conv = [&]() {  // (F64: the float32 h_eff entries themselves, quad_f64 forms (double)h * A)
        const float cz = F64 ? s.h_zone : ps.conv_zone, cb = F64 ? s.h_base : ps.conv_base;
#pragma unroll
        for (int m = 0; m < H; ++m)
            convc[m] = f2{((zoneA >> m) & 1u) ? cz : cb, ((zoneB >> m) & 1u) ? cz : cb};
    };
    build_conv();
    WEDM_STAMP_DECL;
    const bool tracing = WEDM_TRACING(k);
    int trace_next = k.trace_next, trace_slot = k.trace_slot;
    (void)trace_next; (void)trace_slot;

    for (int it = 0; it < k.n_substeps; ++it) {
        if (__all(s.done) && !tracing) break;  // (terminated environments keep being sampled: their frozen state)
        WEDM_STAMP(st0);
        Coef cf{0.0f, 0.0f, 0, -1};
        QuietTry qt;
        const bool was_quiet = quiet_prelude_t<WEDM_WIDE_DENSE>(hv, cold, g, e, gid, s, qt, cf);
        if (!was_quiet) {
            if (!s.done) cf = scalar_prelude(hv, cold, g, e, gid, s, ps, writer, qt);
            build_conv();
        }
        freeze_wire(s);
        WEDM_STAMP(st1);
        const bool act = !s.done;
        float tmax = spool;
        // halos, OLD values: T[base + H - 1] (left of chunk B) and T[base + H] (right of chunk A) are the lane's own; the
        // left of chunk A is the previous lane's last cell, the right of chunk B the next lane's first (every lane takes
        // part in the exchange, frozen environments and padding lanes included)
        const float a_last = P[H - 1].x, b_first = P[0].y;
        float halo_l = dpp_row_shr1(spool, P[H - 1].y), halo_r = dpp_row_shl1(0.0f, P[0].x);
        if (L < 16) { halo_l = c == 0 ? spool : halo_l; halo_r = c == L - 1 ? 0.0f : halo_r; }
        // PLAIN: no lane of the wave carries current or a plasma heat in this microsecond (the ordinary one): no Joule
        // term, no plasma cell, nothing to look for -- the walk is its two tiles and the wire's last cell
        const bool busy = __any(cf.joule_on != 0 || cf.pidx >= 0 || cf.q != 0.0f);
        auto walk = [&](auto plain_tag) {
            constexpr bool PLAIN = decltype(plain_tag)::value;
            const Coef cz{0.0f, 0.0f, 0, -1};
            const Coef& cw = PLAIN ? cz : cf;
            // (uniform geometry, opaque where it is used: the predicates of the rare per-cell code are computed there
            // instead of once before the loop and kept -- see wedm_step_regs)
            Geom gw = g;
            int nw = n;
            auto prep_gw = [&]() {
                gw.n_seg = __builtin_amdgcn_readfirstlane(g.n_seg); gw.az_start = __builtin_amdgcn_readfirstlane(g.az_start);
                gw.az_end = __builtin_amdgcn_readfirstlane(g.az_end); gw.cb = __builtin_amdgcn_readfirstlane(g.cb);
                gw.ct = __builtin_amdgcn_readfirstlane(g.ct);
                asm volatile("" : "+s"(gw.n_seg), "+s"(gw.az_start), "+s"(gw.az_end), "+s"(gw.cb), "+s"(gw.ct));
                nw = __builtin_amdgcn_readfirstlane(n);
                asm volatile("" : "+s"(nw));
            };
            if (!PLAIN) prep_gw();
            const float jf_lane = (!PLAIN && cf.joule_on) ? (F64 ? 1.0f : cf.jf) : 0.0f;  // (float64 typing: a flag, the factor is cf.jf64)
            const bool joule_wave = !PLAIN && __any(jf_lane != 0.0f);
            // the tiles that hold some lane's plasma cell (a lane's own cells only)
            const int pcell = (!PLAIN && cf.pidx >= 1 && cf.pidx >= base && cf.pidx < base + 2 * H) ? cf.pidx - base : -1;  // lane-local
            uint32_t ptiles = 0u;
            if (!PLAIN && __any(pcell >= 0)) {
                const int pt = pcell >= 0 ? ((pcell & (H - 1)) >> 3) : -1;
#pragma unroll
                for (int t = 0; t < H / 8; ++t) ptiles |= __any(pt == t) ? (1u << t) : 0u;
            }
            const uint32_t odd = ptiles | last_tile | cut;  // regular tiles with cells to patch
            const f2 jfp = f2{jf_lane, jf_lane};
            f2 leftp = f2{halo_l, a_last};  // OLD pair before the tile
#pragma unroll
            for (int t = 0; t < H / 8; ++t) {
                const int j = 8 * t;
                f2 tm[8], tc[8], tp[8], pn[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    tc[u] = P[j + u];
                    tm[u] = u == 0 ? leftp : P[j + u - 1];
                    tp[u] = (j + u + 1 < H) ? P[j + u + 1 < H ? j + u + 1 : 0] : f2{b_first, halo_r};
                }
                leftp = tc[7];
                {
                    f2 tmA[4], tcA[4], tpA[4], pnA[4], tmB[4], tcB[4], tpB[4], pnB[4], cvA[4], cvB[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        tmA[u] = tm[u]; tcA[u] = tc[u]; tpA[u] = tp[u]; tmB[u] = tm[4 + u]; tcB[u] = tc[4 + u]; tpB[u] = tp[4 + u];
                        cvA[u] = convc[j + u]; cvB[u] = convc[j + 4 + u];
                    }
                    if (!PLAIN && joule_wave) {
                        f2 jvA[4], jvB[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { jvA[u] = jm[j + u] * jfp; jvB[u] = jm[j + 4 + u] * jfp; }
                        rw_quad<F64, true, true, SW>(tmA, tcA, tpA, pnA, g, cvA, tdiel, ps, jvA, alpha, tref, f64c, cf);
                        rw_quad<F64, true, true, SW>(tmB, tcB, tpB, pnB, g, cvB, tdiel, ps, jvB, alpha, tref, f64c, cf);
                    } else {
                        rw_quad<F64, false, true, SW>(tmA, tcA, tpA, pnA, g, cvA, tdiel, ps, cvA, alpha, tref, f64c, cw);
                        rw_quad<F64, false, true, SW>(tmB, tcB, tpB, pnB, g, cvB, tdiel, ps, cvB, alpha, tref, f64c, cw);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { pn[u] = pnA[u]; pn[4 + u] = pnB[u]; }
                    if (t == 0) pn[0].x = (c == 0) ? spool : pn[0].x;  // wire cell 0
                    // The cells that are not interior cells take the predicated formula from the same OLD values and replace the
                    // regular result BEFORE the maximum is taken: the maximum is over the true new temperatures, whatever
                    // the sign of a plasma heat (no per-cell fallback for a negative one, as the LDS kernels need).
                    if ((odd >> t) & 1u) {
                        if (PLAIN) prep_gw();
                        // the wire's last cell (last position of its tile)
                        if ((last_tile >> t) & 1u) {
                            const float x = rw_cell<F64>(base + (last_in_b ? H : 0) + j + 7, nw, last_in_b ? tm[7].y : tm[7].x,
                                                         last_in_b ? tc[7].y : tc[7].x, 0.0f, gw, cw, ps, tref, alpha, tdiel, f64c, s.h_base, s.h_zone);
                            const bool hx = owns_last && !last_in_b, hy = owns_last && last_in_b;
                            pn[7].x = hx ? x : pn[7].x; pn[7].y = hy ? x : pn[7].y;
                        }
                        // the wire's last cell inside a tile that the end cuts: the same, at its (uniform) place
                        if ((cut >> t) & 1u) {
                            int lw = lloc;  // (opaque here: or its 16 compares are made before the loop and kept in spilled scalars)
                            asm volatile("" : "+s"(lw));
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                if (lw == j + u) {
                                    const float x = rw_cell<F64>(base + j + u, nw, tm[u].x, tc[u].x, 0.0f, gw, cw, ps, tref, alpha, tdiel, f64c, s.h_base, s.h_zone);
                                    pn[u].x = owns_last ? x : pn[u].x;
                                }
                                if (lw == H + j + u) {
                                    const float x = rw_cell<F64>(base + H + j + u, nw, tm[u].y, tc[u].y, 0.0f, gw, cw, ps, tref, alpha, tdiel, f64c, s.h_base, s.h_zone);
                                    pn[u].y = owns_last ? x : pn[u].y;
                                }
                            }
                        }
                        // plasma cells of the lanes that have one in this tile
                        if (!PLAIN && ((ptiles >> t) & 1u)) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                if (__any(pcell == j + u)) {
                                    const float x = rw_cell<F64>(base + j + u, nw, (base + j + u == 1) ? spool : tm[u].x, tc[u].x, tp[u].x, gw, cf, ps, tref, alpha, tdiel, f64c, s.h_base, s.h_zone);
                                    pn[u].x = (pcell == j + u) ? x : pn[u].x;
                                }
                                if (__any(pcell == H + j + u)) {
                                    const float x = rw_cell<F64>(base + H + j + u, nw, tm[u].y, tc[u].y, tp[u].y, gw, cf, ps, tref, alpha, tdiel, f64c, s.h_base, s.h_zone);
                                    pn[u].y = (pcell == H + j + u) ? x : pn[u].y;
                                }
                            }
                        }
                    }
                    float mx, my;
                    if ((cut >> t) & 1u) {
                        // cell by cell over the cells that exist
                        mx = spool; my = spool;
                        int va = nA, vb = nB;  // (opaque for the same reason)
                        asm volatile("" : "+v"(va), "+v"(vb));
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            mx = (j + u < va) ? fmax_gt(mx, pn[u].x) : mx;
                            my = (j + u < vb) ? fmax_gt(my, pn[u].y) : my;
                        }
                    } else {
                        // the maximum of the chunk halves that exist (a tile is whole or padding here)
                        mx = max3_raw(pn[0].x, pn[1].x, pn[2].x); my = max3_raw(pn[0].y, pn[1].y, pn[2].y);
                        mx = max3_raw(mx, pn[3].x, pn[4].x); my = max3_raw(my, pn[3].y, pn[4].y);
                        mx = max3_raw(mx, pn[5].x, pn[6].x); my = max3_raw(my, pn[5].y, pn[6].y);
                        mx = fmax_gt(mx, pn[7].x); my = fmax_gt(my, pn[7].y);
                        mx = nA > j ? mx : spool; my = nB > j ? my : spool;
                    }
                    tmax = max3_raw(tmax, mx, my);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) P[j + u] = pn[u];
            }
        };
        if (act) {  // (the lanes of terminated environments sit the walk out: their registers stay)
            if (busy) walk(std::false_type{});
            else walk(std::true_type{});
        }
        WEDM_STAMP(st2);
        // the maximum over the lanes of the environment (all lanes take part; frozen and padding lanes hold the spool value)
        tmax = fmax_gt(tmax, dpp_perm<0xB1>(tmax));   // quad_perm [1,0,3,2]
        tmax = fmax_gt(tmax, dpp_perm<0x4E>(tmax));   // quad_perm [2,3,0,1]
        if (L >= 8) tmax = fmax_gt(tmax, dpp_perm<0x141>(tmax));  // row_half_mirror
        if (L >= 16) tmax = fmax_gt(tmax, dpp_perm<0x140>(tmax)); // row_mirror
        unfreeze_wire(hv, s);
        WEDM_STAMP(st3);
        if (!s.done) {
            scalar_epilogue(hv, s, tmax);
            if (s.ctrl) control_step_outputs(cold, e, s, writer);
        }
        WEDM_STAMP(st4);
        WEDM_STAMP_ACC();
        WEDM_TRACE_POINT(k, it, e, s, writer,
                         for (int m = 0; m < H; ++m) {
                             if (base + m < n) tT[(int64_t)(base + m) * tcnt] = P[m].x;
                             if (base + H + m < n) tT[(int64_t)(base + H + m) * tcnt] = P[m].y;
                         });
    }
    WEDM_STAMP_OUT();

    if (live) {
#pragma unroll
        for (int q = 0; q < 2 * H / 4; ++q) {
            const int m = (q % (H / 4)) * 4;
            const bool hi = q >= H / 4;
            const f4v w = hi ? f4v{P[m].y, P[m + 1].y, P[m + 2].y, P[m + 3].y} : f4v{P[m].x, P[m + 1].x, P[m + 2].x, P[m + 3].x};
            const int cell = base + 4 * q;
            if (cell + 3 < n) {
                *(f4v*)(Te + (int64_t)(q0 + q) * stride * 4) = w;
            } else {  // the wire's last, partial word: the cells past the end are padding and keep their value
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (cell + u < n) Te[(int64_t)(q0 + q) * stride * 4 + u] = w[u];
            }
        }
    }
    if (live && writer) {
        if (WEDM_REWARD_ON(cold)) {
            if (!frozen0) write_reward(cold, e, s);
            else cold->s.reward[e] = 0.0f;  // a frozen environment earns nothing (not the previous launch's reward)
        }
        store_time_hi(cold, e, s, (uint32_t)k.n_substeps * (uint32_t)k.hot.dt_us);
        store_env(cold, e, s);
    }
}


