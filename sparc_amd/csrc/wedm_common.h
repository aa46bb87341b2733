// wedm_common.h — what every step kernel shares: build switches, the wave-uniform tile table (WalkTable), the kernel
// arguments, the signal-trace point, wire accessors and the block-cooperative wire copy, the cell-by-cell pass of the
// global-memory kernel, the stage-major tile code (tile_staged / quad_staged), the float64-typed cells (cell_f64 / quad_f64 / rw_quad / rw_cell) and the
// diagnostic phase stamps.
//
// Included by wedm_kernels.hip (one translation unit per WEDM_PART; see the bottom of that file).
#pragma once


// Which kernels let their wave-uniform fast path also carry burning / ending sparks (quiet_prelude_t<true>), A/B-timed
// on the MI355X (tools/r2_run13.sh, r2_run14.sh): the packed kernel gains everywhere (bench workload +1.3 %, 15 um gap
// +4.1 %, closed loop +3.6 %), the unpacked fused and the predicated kernels lose 1-4 % on every workload (registers).
#ifndef WEDM_PACKED_DENSE
#define WEDM_PACKED_DENSE true
#endif
#ifndef WEDM_FUSED_DENSE
#define WEDM_FUSED_DENSE false
#endif
// the fused kernel's N1 instantiation requests a tile's LDS rows one tile ahead (see PREFETCH there)
#ifndef WEDM_FUSED_MIN_BLOCKS
#define WEDM_FUSED_MIN_BLOCKS 2
#endif
#ifndef WEDM_PACKED_MIN_BLOCKS
#define WEDM_PACKED_MIN_BLOCKS 2
#endif
#ifndef WEDM_STREAM_PAIRED_LOADS
#define WEDM_STREAM_PAIRED_LOADS 1
#endif
// the stream kernel walks a launch of ONE microsecond out of the registers the wire was loaded into (see rest_single)
#ifndef WEDM_STREAM_DENSE_QUIET
#define WEDM_STREAM_DENSE_QUIET 1
#endif
#ifndef WEDM_STREAM_REGWALK
#define WEDM_STREAM_REGWALK 1
#endif
#ifndef WEDM_PIN_STAGE
#define WEDM_PIN_STAGE 0
#endif
#ifndef WEDM_PREFETCH_N1
#define WEDM_PREFETCH_N1 0
#endif

// Wave-uniform description of one step's walk over a chunk of C cells (see build_walk()).
// Cell j of chunk c is wire segment i = c*C + j.  The chunk is walked in ceil(C/8) tiles of 8
// cells, each of a kind that is the same for every chunk (TILE_N / TILE_B / TILE_S and the masks
// below that let further tiles take the regular code).
#define WEDM_MAX_C 160  // 160 KB LDS / (256 lanes * 4 B)
#define WEDM_MAX_TILES (WEDM_MAX_C / 8 + 1)
struct WalkTable {
    int32_t C;                             // cells per chunk = ceil(n_seg / L)
    int32_t n_tiles;                       // ceil(C / 8)
    // dword entries so that the (wave-uniform) lookups compile to scalar loads:
    uint32_t zj[WEDM_MAX_TILES * 8];  // bits 0-15: chunk c has cell j inside the workpiece zone;
                                      // bits 16-31: chunk c has cell j between the contacts
    uint32_t iv[WEDM_MAX_TILES * 8];  // bits 0-15: 1 <= c*C + j <= n-2 (interior, j < C);
                                      // bits 16-31: c*C + j < n (valid, j < C)
    uint32_t kind[WEDM_MAX_TILES];    // TILE_N / TILE_B / TILE_S
    uint32_t split[WEDM_MAX_TILES];   // TILE_B: first cell offset that uses the tile's second flag set (8: none)
    // the same, gathered by the host the way the kernels keep it in registers (bit t = tile t): per chunk
    // {zone of the tile's first cell, between the contacts (first cell), zone (last cell), contacts (last cell)},
    // and wave-uniform tile kinds / split offsets (4 bits per tile).  One 16-byte load per lane instead of a
    // loop of dependent table reads per launch (which cost the single-microsecond kernel ~2 us per launch).
    uint32_t chunk_flags[16][4];
    uint32_t kind_n_mask, kind_s_mask;
    uint32_t split_pack[3];
    // tiles that can ALSO take the regular (TILE_N) code: kind_ne_mask = full tiles with one flag set whose only
    // non-interior cells are the wire's end cells (cell 0 = first cell of chunk 0's tile 0, cell n-1 = last cell of the
    // last chunk's last tile: computed by the interior formula like the rest, kept out of the maximum, patched after
    // the walk like every boundary cell); kind_nj_mask = the same where only the between-the-contacts flag changes inside
    // the tile, which matters only in a microsecond in which some lane of the wave carries current.
    uint32_t kind_ne_mask, kind_nj_mask;
    // kind_n1_mask: full tiles, end cells apart all interior, with exactly ONE flag change (bit 31: at least one of them
    // changes the ZONE flag, i.e. is a boundary tile in every microsecond): the N1 instantiation of wedm_step_fused runs
    // them stage-major with per-cell coefficients, without a boundary tile's predicated stores and maxima
    uint32_t kind_n1_mask;
};
// TILE_N: 8 interior cells, one flag set.  TILE_B: every cell takes the interior formula with at
// most one flag change inside the tile; boundary cells (wire cell 0, the last cell, cells past
// the end of the wire) are kept out of the running max and patched afterwards.  TILE_S: per-cell
// predicated fallback (more than one flag change in a tile).
enum { TILE_N = 0, TILE_B = 1, TILE_S = 2 };

struct KArgs {
    Hot hot;    // every-step parameters, by value
    Cold cold;  // device pointers: full wedm_params copy, state/geometry/action blocks, tables
    int32_t num_envs;
    int32_t n_substeps;
    int32_t n_seg_max;
    const WalkTable* walk;  // device copy of the table for the L in use (fused kernel only)
    int32_t trace_next;     // substep index after which the next trace sample is due (INT32_MAX: no trace)
    int32_t trace_slot;     // ring slot of that sample
    wedm_trace_desc trace;  // the bound trace (by value: one kernarg s_load, only in the TRACE instantiations)
    unsigned long long* dbg; // diagnostic builds only (WEDM_STAMPS): per-wave phase cycle sums
};

// The by-value `cold` member as the kernels read it: through the kernarg segment (wedm_device.h).
__device__ __forceinline__ ColdRef kernarg_cold() {
    return ColdRef{(ColdPtr)((const WEDM_AS4 char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KArgs, cold))};
}

// ------------------------------------------------------------ signal trace
// The sample schedule is host-made and identical for every wave: `it == trace_next` is a scalar
// compare per microsecond; the descriptor travels by value in the kernel arguments.  While a
// trace is due in this launch the kernels keep iterating over terminated environments so that
// every slot receives a sample (their frozen state).
// Kernels are instantiated with and without the trace point (template parameter TRACE): the
// inlined sampling code costs the packed kernel 4 more spilled VGPRs (scratch 80 -> 100 B/lane)
// and the global kernel half its occupancy, so launches without a bound trace run the
// instantiation that does not contain it.
#define WEDM_TRACING(k) (TRACE && (k).trace_next < (k).n_substeps)
// CELLS: statement that copies this lane's wire cells, given `tT` (slot base + column) and `tcnt`
#define WEDM_TRACE_POINT(k, it, e, s, SCALAR_LANE, CELLS)                                        \
    if (TRACE && (it) == trace_next) {                                                           \
        const wedm_trace_desc& tr = (k).trace;                                                   \
        const int64_t tcol = trace_column(tr, (e));                                              \
        if (tcol >= 0) {                                                                         \
            if (SCALAR_LANE) trace_scalars(tr, tcol, (s), trace_slot, (k).hot.done_value == 0);  \
            if (tr.T) {                                                                          \
                const int64_t tcnt = tr.env_count;                                               \
                float* tT = tr.T + (int64_t)trace_slot * (k).n_seg_max * tcnt + tcol;            \
                CELLS;                                                                           \
            }                                                                                    \
        }                                                                                        \
        trace_next += tr.every;                                                                  \
        trace_slot = (trace_slot + 1 == tr.capacity) ? 0 : trace_slot + 1;                       \
    }

// ------------------------------------------------------------ T accessors
typedef float f4v __attribute__((ext_vector_type(4)));

// One environment's wire in the quad-interleaved block T[seg >> 2][env][seg & 3] (WEDM_T_INDEX).
struct GlobalT {
    float* base;      // &T[0][e][0]
    int64_t qstride;  // elements between consecutive quads of one environment (4 * stride)
    __device__ __forceinline__ float ld(int i) const { return base[(int64_t)(i >> 2) * qstride + (i & 3)]; }
    __device__ __forceinline__ void st(int i, float v) const { base[(int64_t)(i >> 2) * qstride + (i & 3)] = v; }
    __device__ __forceinline__ f4v ldq(int q) const { return *(const f4v*)(base + (int64_t)q * qstride); }
    __device__ __forceinline__ void stq(int q, f4v v) const { *(f4v*)(base + (int64_t)q * qstride) = v; }
};
__device__ __forceinline__ GlobalT global_wire(float* T, int64_t stride, int64_t e) { return GlobalT{T + 4 * e, 4 * stride}; }

// Block-cooperative copy of the wire cells [0, n) of the block's 256 / L environments between the quad-interleaved
// block in HBM and the kernel's LDS image, 16 bytes per lane and instruction (a wave touches contiguous runs of
// 64 x 16 B).  `slot(i)` = LDS float offset of wire cell i for the block's first environment (the kernel's own
// chunk / row mapping); environment slot `sel` adds sel * L.  Cells of the last quad past n are padding: not copied.
template <int L, bool TO_LDS, class Slot>
__device__ __forceinline__ void copy_wire(float* T, int64_t stride, int64_t e0, int num_envs, int n, int tid, float* lds, Slot slot) {
    constexpr int EPB = 256 / L;
    const int qr = tid / EPB, sel = tid % EPB;  // L quads per iteration
    if (e0 + sel >= num_envs) return;
    float* const base = T + 4 * (e0 + sel);
    const int64_t qstride = 4 * stride;
    const int nq = (n + 3) >> 2;
    for (int q = qr; q < nq; q += L) {
        float* const g = base + (int64_t)q * qstride;
        if (TO_LDS) {
            const f4v v = *(const f4v*)g;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * q + k < n) lds[slot(4 * q + k) + sel * L] = v[k];
        } else if (4 * q + 3 < n) {
            f4v v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = lds[slot(4 * q + k) + sel * L];
            *(f4v*)g = v;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * q + k < n) g[k] = lds[slot(4 * q + k) + sel * L];
        }
    }
}

// One in-place pass of wire.py:58-123 over the lane's wire.  Tiles of 8 cells: the 8
// "next" temperatures are loaded before any of the tile's stores, so every cell sees
// OLD neighbours (explicit Euler) with one load + one store per cell.
template <bool F64, class TA>
__device__ __forceinline__ float stencil_pass(const TA& T, const Geom& g, const Coef& c, const Persist& ps, const Hot& hot,
                                              const StencilF64& f64c, float h_base, float h_zone) {
    const float spool = hot.spool, tref = hot.tref, alpha = hot.alpha, tdiel = hot.tdiel;
    (void)tref; (void)alpha; (void)tdiel;
    const int n = g.n_seg;
    T.st(0, spool);  // boundary condition (wire.py:83,123)
    float tmax = spool;
    if (n <= 1) return tmax;
    float tm1 = spool;
    float tc = T.ld(1);
    for (int i0 = 1; i0 < n; i0 += 8) {
        float nx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int idx = i0 + 1 + u;
            nx[u] = idx < n ? T.ld(idx) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int i = i0 + u;
            if (i < n) {
                float tn;
                if (F64) tn = stencil_cell_f64(i, n, tm1, tc, nx[u], g, c, ps, f64c, h_base, h_zone);
                else tn = stencil_cell(i, n, tm1, tc, nx[u], g, c, ps, tref, alpha, tdiel);
                T.st(i, tn);
                tmax = tn > tmax ? tn : tmax;
                tm1 = tc;
                tc = nx[u];
            }
        }
    }
    return tmax;
}

template <bool TRACE, bool F64, bool REPLAY, class TA>
__device__ __forceinline__ void run_substeps(const KArgs& k, const ColdRef cold, const Geom& g, int64_t e,
                                             uint32_t gid, Env& s, const TA& T) {
    Persist ps;
    init_persist(k.hot, cold, e, s, ps);
    StencilF64 f64c{0.0, 0.0, 0.0};
    if (F64) { const wedm_params* pp = cold->p; f64c = StencilF64{pp->temp_ref, pp->alpha_rho, pp->dielectric_temperature}; }
    const bool tracing = WEDM_TRACING(k);
    int trace_next = k.trace_next, trace_slot = k.trace_slot;
    (void)trace_next; (void)trace_slot;
    for (int it = 0; it < k.n_substeps; ++it) {
        if (!s.done) {
            Coef c = scalar_prelude<REPLAY>(k.hot, cold, g, e, gid, s, ps, true);  // single steps: the quiet test does not pay
            // (keep_stepping_terminated: the wire module returns at once on a broken wire, wire.py:260-261)
            float tmax = s.broken ? s.tmax : stencil_pass<F64>(T, g, c, ps, k.hot, f64c, s.h_base, s.h_zone);
            scalar_epilogue(k.hot, s, tmax);
            if (s.ctrl) control_step_outputs(cold, e, s, true);
        } else if (!tracing) {
            break;
        }
        WEDM_TRACE_POINT(k, it, e, s, true,
                         for (int i = 0; i < g.n_seg; ++i) tT[(int64_t)i * tcnt] = T.ld(i));
    }
}

// np.max over finite temperatures; maps to v_max_f32 / v_max3_f32
__device__ __forceinline__ float fmax_gt(float a, float b) { return __builtin_fmaxf(a, b); }

// max over the L lanes of an environment (L a power of two <= 16, the lanes adjacent and aligned inside one DPP row): quad
// permutes, half-row mirror, row mirror -- one v_max_f32 with a DPP operand per doubling.  (__shfl_xor is a ds_bpermute with
// its address arithmetic: ~4 instructions and an LDS round trip per doubling.)
template <int CTRL>
__device__ __forceinline__ float dpp_lane_perm(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
template <int L>
__device__ __forceinline__ float max_over_env_lanes(float v) {
    static_assert(L == 1 || L == 2 || L == 4 || L == 8 || L == 16, "lanes per environment");
    if (L >= 2) v = fmax_gt(v, dpp_lane_perm<0xB1>(v));    // quad_perm [1,0,3,2]
    if (L >= 4) v = fmax_gt(v, dpp_lane_perm<0x4E>(v));    // quad_perm [2,3,0,1]
    if (L >= 8) v = fmax_gt(v, dpp_lane_perm<0x141>(v));   // row_half_mirror
    if (L >= 16) v = fmax_gt(v, dpp_lane_perm<0x140>(v));  // row_mirror
    return v;
}

#ifdef WEDM_STAMPS
// -DWEDM_STAMPS_REAL: the 100 MHz clock all XCDs share (10 ns per tick: start / end skew across the chip) instead of the
// per-XCD shader clock (phase lengths inside a wave)
#ifdef WEDM_STAMPS_REAL
#define WEDM_S2_CLOCK "s_memrealtime"
#else
#define WEDM_S2_CLOCK "s_memtime"
#endif
#define WEDM_S2_STAMP_DECL unsigned long long sst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    asm volatile(WEDM_S2_CLOCK " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sst[7])::"memory")
#define WEDM_S2_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile(WEDM_S2_CLOCK " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sst[i])::"memory"); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#define WEDM_S2_STAMP_VM(i) do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_waitcnt vmcnt(0)\n\t" WEDM_S2_CLOCK " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sst[i])::"memory"); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#define WEDM_S2_STAMP_OUT() do { if (k.dbg && (threadIdx.x & 63) == 0) { \
    unsigned long long* o = k.dbg + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 12; \
    for (int q = 0; q < 12; ++q) o[q] = sst[q]; } } while (0)
#else
#define WEDM_S2_STAMP_DECL do { } while (0)
#define WEDM_S2_STAMP(i) do { } while (0)
#define WEDM_S2_STAMP_VM(i) do { } while (0)
#define WEDM_S2_STAMP_OUT() do { } while (0)
#endif

typedef float f2 __attribute__((ext_vector_type(2)));

// Eight cells (V = float) or eight packed cell pairs (V = float2) evaluated STAGE-MAJOR: every stage applies one operation
// of interior2() to all eight pairs, and a scheduling barrier separates the stages, so dependent
// packed ops are always >= 8 instructions apart.  Left to itself the scheduler emits the eight
// chains one after the other (each op waiting on the previous, s_nop in between).  Operation
// order and rounding are exactly those of interior2().  old[u], old[u+1], old[u+2] are the OLD
// (tm1, tc, tp1) of pair u.  conv/jfe: one coefficient pair per cell (PERCELL) or per tile.
#define WEDM_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
// W pairs starting at pair `o` of the tile (W = 4: two half-tiles keep the temporaries, and
// with them the scratch spills of the caller's state, small; 4-way ILP already covers the
// packed-op latency).
template <class V, bool JOULE, bool PERCELL, int W>
__device__ __forceinline__ void tile_staged(const V (&old)[10], V (&tn)[8], const int o, float k, float tuf,
                                            const V (&conv)[8], float tdiel, float adv, const V (&jfe)[8],
                                            float alpha, float tref) {
    V a[W], e[W], f[W], r[W];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        a[u] = sub_twice(old[o + u], old[o + u + 1]);  // T[i-1] - 2*T[i] (exact product, one rounding)
        e[u] = old[o + u + 1] - tdiel;           // T[i] - T_dielectric
        f[u] = old[o + u] - old[o + u + 1];      // T[i-1] - T[i]
        if (JOULE) r[u] = old[o + u + 1] - tref;
    }
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) {
        e[u] = (PERCELL ? conv[o + u] : conv[0]) * e[u];
        f[u] = adv * f[u];
        if (JOULE) r[u] = alpha * r[u];
#if WEDM_PIN_STAGE
        // (the optimiser otherwise sinks this product down to its only use, `a - e`, where it folds the negation into the
        // multiply and leaves a three-deep dependent chain with wait states in the stage that was meant to be one add)
        asm volatile("" : "+v"(e[u]));
#endif
    }
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) {
        a[u] = a[u] + old[o + u + 2];
        if (JOULE) r[u] = 1.0f + r[u];
    }
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) {
        a[u] = k * a[u];
        if (JOULE) r[u] = (PERCELL ? jfe[o + u] : jfe[0]) * r[u];
    }
    WEDM_STAGE_FENCE();
    if (JOULE) {
#pragma unroll
        for (int u = 0; u < W; ++u) a[u] = a[u] + r[u];
        WEDM_STAGE_FENCE();
    }
#pragma unroll
    for (int u = 0; u < W; ++u) a[u] = a[u] - e[u];
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) a[u] = a[u] + f[u];
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) a[u] = a[u] * tuf;
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) tn[o + u] = old[o + u + 1] + a[u];
    WEDM_STAGE_FENCE();
}

#ifndef WEDM_STAGE_W
#define WEDM_STAGE_W 4
#endif
#ifndef WEDM_STAGE_W_PACKED
#define WEDM_STAGE_W_PACKED 2  // as fast as 4 (the other wave of the SIMD fills the gaps) and 16 VGPRs cheaper
#endif
template <class V, bool JOULE, bool PERCELL>
__device__ __forceinline__ void tile8_staged(const V (&old)[10], V (&tn)[8], float k, float tuf, const V (&conv)[8],
                                             float tdiel, float adv, const V (&jfe)[8], float alpha, float tref) {
    constexpr int W = sizeof(V) == 8 ? WEDM_STAGE_W_PACKED : WEDM_STAGE_W;
#pragma unroll
    for (int o = 0; o < 8; o += W)
        tile_staged<V, JOULE, PERCELL, W>(old, tn, o, k, tuf, conv, tdiel, adv, jfe, alpha, tref);
}

// Eight ADJACENT cells of one chunk as four packed pairs (cells 2m, 2m+1), stage-major like tile_staged: tm / tc / tp are
// the OLD (T[i-1], T[i], T[i+1]) of both cells of pair m -- tm and tp are the chunk's registers shifted by one cell
// (one v_pk_mov_b32 or two v_mov_b32 each), which is what a register-resident walk pays instead of LDS round trips.
// Operation order and rounding are those of interior_cell().
#ifndef WEDM_QUAD_STAGE_W
#define WEDM_QUAD_STAGE_W 2
#endif
template <bool JOULE, bool PERCELL, int W>
__device__ __forceinline__ void quad_stage_group(const f2 (&tm)[4], const f2 (&tc)[4], const f2 (&tp)[4], f2 (&tn)[4], const int o,
                                                 float k, float tuf, const f2 (&conv)[4], float tdiel, float adv,
                                                 const f2 (&jfe)[4], float alpha, float tref) {
    f2 a[W], e[W], f[W], r[W];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        a[u] = sub_twice(tm[o + u], tc[o + u]);
        e[u] = tc[o + u] - tdiel;
        f[u] = tm[o + u] - tc[o + u];
        if (JOULE) r[u] = tc[o + u] - tref;
    }
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) {
        e[u] = (PERCELL ? conv[o + u] : conv[0]) * e[u];
        f[u] = adv * f[u];
        if (JOULE) r[u] = alpha * r[u];
    }
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) {
        a[u] = a[u] + tp[o + u];
        if (JOULE) r[u] = 1.0f + r[u];
    }
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) {
        a[u] = k * a[u];
        if (JOULE) r[u] = (PERCELL ? jfe[o + u] : jfe[0]) * r[u];
    }
    WEDM_STAGE_FENCE();
    if (JOULE) {
#pragma unroll
        for (int u = 0; u < W; ++u) a[u] = a[u] + r[u];
        WEDM_STAGE_FENCE();
    }
#pragma unroll
    for (int u = 0; u < W; ++u) a[u] = a[u] - e[u];
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) a[u] = a[u] + f[u];
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) a[u] = a[u] * tuf;
    WEDM_STAGE_FENCE();
#pragma unroll
    for (int u = 0; u < W; ++u) tn[o + u] = tc[o + u] + a[u];
    WEDM_STAGE_FENCE();
}
// W pairs per stage: 2 where registers are short (the stream kernel, two waves per SIMD: the other wave fills the gaps),
// 4 where a wave is alone on its SIMD and a dependent packed operation two instructions later would wait (register kernel)
template <bool JOULE, bool PERCELL, int W = WEDM_QUAD_STAGE_W>
__device__ __forceinline__ void quad_staged(const f2 (&tm)[4], const f2 (&tc)[4], const f2 (&tp)[4], f2 (&tn)[4], float k,
                                            float tuf, const f2 (&conv)[4], float tdiel, float adv, const f2 (&jfe)[4],
                                            float alpha, float tref) {
#pragma unroll
    for (int o = 0; o < 4; o += W)
        quad_stage_group<JOULE, PERCELL, W>(tm, tc, tp, tn, o, k, tuf, conv, tdiel, adv, jfe, alpha, tref);
}

#ifndef WEDM_REGS_F64_FENCE
#define WEDM_REGS_F64_FENCE 1  // pairs between scheduling fences in quad_f64 (A/B: 0: 7.9e9, 1: 8.3e9, 2: 7.9e9 at 65 536 x 128)
#endif
// ---- stencil_mode 1 (the stencil as Numba types wire.py:58-123) in the register walk.
// An interior cell of that typing: stencil_cell_f64's sequence of float64 expressions and float32 roundings with the
// coefficients handed in; the Joule block only in the tiles that have current in this microsecond (JOULE), as the reference
// skips it (wire.py:96-100) -- 18 float64 operations per cell without it (8 of them conversions), 25 with it, against 5.5 of
// the packed float32 form.
template <bool JOULE>
__device__ __forceinline__ float cell_f64(float tm1, float tc, float tp1, double k64, double tuf64, double conv64, double tdiel64,
                                          double adv64, double jfe64, double alpha64, double tref64) {
    const double m = (double)tm1, t = (double)tc;
    // (m - 2.0 * t: 2 t is exact, so the difference rounds once -- one fused operation instead of t + t and a subtraction)
    float d = (float)(k64 * (__builtin_fma(-2.0, t, m) + (double)tp1));
    if (JOULE) {
        const double rho_T = 1.0 + alpha64 * (t - tref64);
        d = (float)((double)d + jfe64 * rho_T);
    }
    d = (float)((double)d - conv64 * (t - tdiel64));
    d = (float)((double)d + adv64 * (m - t));
    return (float)(t + (double)d * tuf64);
}

// Four pairs of a tile in that typing.  `hc`: the float32 convection entries h_eff of the two halves (wire.py:205; the
// coefficient is (double)h * A), per pair where PERCELL; `jfl`: non-zero where a cell lies between the contacts of a lane
// that carries current.
template <bool JOULE, bool PERCELL>
__device__ __forceinline__ void quad_f64(const f2 (&tm)[4], const f2 (&tc)[4], const f2 (&tp)[4], f2 (&tn)[4], const Geom& g,
                                         const f2 (&hc)[4], const Persist& ps, const f2 (&jfl)[4], const StencilF64& h, double jf64) {
    double cA = (double)hc[0].x * g.a64, cB = (double)hc[0].y * g.a64;
    double jA = (JOULE && jfl[0].x != 0.0f) ? jf64 : 0.0, jB = (JOULE && jfl[0].y != 0.0f) ? jf64 : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (PERCELL && u > 0) {
            cA = (double)hc[u].x * g.a64; cB = (double)hc[u].y * g.a64;
            jA = (JOULE && jfl[u].x != 0.0f) ? jf64 : 0.0; jB = (JOULE && jfl[u].y != 0.0f) ? jf64 : 0.0;
        }
        tn[u].x = cell_f64<JOULE>(tm[u].x, tc[u].x, tp[u].x, g.k64, g.tuf64, cA, h.tdiel, ps.adv64, jA, h.alpha, h.tref);
#if WEDM_REGS_F64_FENCE < 0
        __builtin_amdgcn_sched_barrier(0);
#endif
        tn[u].y = cell_f64<JOULE>(tm[u].y, tc[u].y, tp[u].y, g.k64, g.tuf64, cB, h.tdiel, ps.adv64, jB, h.alpha, h.tref);
#if WEDM_REGS_F64_FENCE < 0
        __builtin_amdgcn_sched_barrier(0);
#elif WEDM_REGS_F64_FENCE > 0
        if ((u + 1) % WEDM_REGS_F64_FENCE == 0) __builtin_amdgcn_sched_barrier(0);  // (cells in flight: 2 per pair between fences)
#endif
    }
}

// what wedm_regs_walk.inc calls: the packed float32 quad / the predicated float32 cell, or their float64-typed forms
template <bool F64, bool JOULE, bool PERCELL, int SW>
__device__ __forceinline__ void rw_quad(const f2 (&tm)[4], const f2 (&tc)[4], const f2 (&tp)[4], f2 (&tn)[4], const Geom& g,
                                        const f2 (&conv)[4], float tdiel, const Persist& ps, const f2 (&jfe)[4], float alpha,
                                        float tref, const StencilF64& h, const Coef& cf) {
    if (F64) quad_f64<JOULE, PERCELL>(tm, tc, tp, tn, g, conv, ps, jfe, h, cf.jf64);
    else quad_staged<JOULE, PERCELL, SW>(tm, tc, tp, tn, g.k, g.tuf, conv, tdiel, ps.adv, jfe, alpha, tref);
}
template <bool F64>
__device__ __forceinline__ float rw_cell(int i, int n_seg, float tm1, float tc, float tp1, const Geom& g, const Coef& cf,
                                         const Persist& ps, float tref, float alpha, float tdiel, const StencilF64& h,
                                         float h_base, float h_zone) {
    if (F64) return stencil_cell_f64(i, n_seg, tm1, tc, tp1, g, cf, ps, h, h_base, h_zone);
    return stencil_cell(i, n_seg, tm1, tc, tp1, g, cf, ps, tref, alpha, tdiel);
}

// A wave that starts with a terminated (frozen) environment in a kernel instantiation without the frozen-lane tile
// code tells the host (Cold::frozen_seen, host-visible): the next launches of the handle take the FROZEN_OK instantiation.
#define WEDM_REPORT_FROZEN(cond)                                                      \
    do {                                                                              \
        if (!kFrozenOk && __any(cond)) {                                              \
            int32_t* const seen = cold->frozen_seen;                                  \
            if (seen && (threadIdx.x & 63) == 0) *seen = 1;                           \
        }                                                                             \
    } while (0)

// In-kernel phase stamps (diagnostic build -DWEDM_STAMPS only; never in the shipped library).
#ifdef WEDM_STAMPS
#define WEDM_STAMP(var)                                                      \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                   \
    } while (0)
#define WEDM_STAMP_DECL unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, tk0 = 0, tk1 = 0, accN = 0, accB = 0, accS = 0, cntN = 0, cntB = 0, cntS = 0
#define WEDM_STAMP_ACC() do { acc0 += st1 - st0; acc1 += st2 - st1; acc2 += st3 - st2; acc3 += st4 - st3; } while (0)
#define WEDM_STAMP_OUT()                                                                         \
    do {                                                                                         \
        if (k.dbg && (threadIdx.x & 63) == 0) {                                                  \
            unsigned long long* o = k.dbg + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;   \
            o[0] = acc0; o[1] = acc1; o[2] = acc2; o[3] = acc3;                                  \
            unsigned long long* o2 = k.dbg + (size_t)gridDim.x * 16 + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 6; \
            o2[0] = accN; o2[1] = accB; o2[2] = accS; o2[3] = cntN; o2[4] = cntB; o2[5] = cntS;  \
        }                                                                                        \
    } while (0)
#else
#define WEDM_STAMP(var) do { } while (0)
#define WEDM_STAMP_DECL do { } while (0)
#define WEDM_STAMP_ACC() do { } while (0)
#define WEDM_STAMP_OUT() do { } while (0)
#endif

