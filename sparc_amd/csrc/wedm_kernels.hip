// wedm_kernels.hip — gfx950 kernels + the C-ABI of include/wedm_hip.h.
//
// Kernels (DESIGN.md section 4 has the table with what binds each of them)
//   wedm_step_global : one lane per environment, wire temperature walked in place in global memory; the float64-stencil
//                      and variate-injection modes, and wires no LDS kernel fits.
// The wire block is quad-interleaved, T[seg >> 2][env][seg & 3] (include/wedm_hip.h, ABI v4): a lane that owns a run of
// segments of one environment moves it with global_load / store_dwordx4, a wavefront still touches contiguous 1-KB runs.
//   wedm_step_split  : single microseconds where the stream kernel does not fit: the wire cut over the four waves of
//                      a block, in place in global memory.
//   wedm_step_stream<L>: single microseconds (the reference's step() cadence), uniform geometry: the whole chunk of a
//                      lane requested up front into registers, one tile walk, no barrier.  Launches of exactly one
//                      microsecond have their own instantiation: the walk runs out of those registers (packed pairs of
//                      adjacent cells), every tile stored where it is computed; otherwise the walk is the LDS one.
//   wedm_step_lanes<L>: any geometry (one (h, d) pair per environment: BASELINE config 5).  L lanes per environment,
//                      wire chunks in LDS; interior formula stage-major with per-cell coefficients from the lane's own
//                      indices, boundary / plasma cells patched (the per-cell predicated walk remains as fallback).
//   wedm_step_fused<L>: uniform geometry.  L lanes share one environment: the wire is cut into L chunks, chunk c of
//                      environment el lives in LDS column (el*L + c) as [cell j][256 lanes] (lane-linear ->
//                      conflict-free), halos are read from the neighbour lane's column before any store of the step
//                      (wave lock-step, no barrier).  The scalar physics runs redundantly in the L lanes (bit-identical
//                      inputs -> bit-identical results).  The walk follows a host-built, wave-uniform TILE TABLE
//                      (build_walk): regular tiles of 8 cells run stage-major without a per-cell predicate; boundary,
//                      plasma and tail cells are patched from values computed before the walk.
//   wedm_step_packed<L>: the same with two chunks per lane advanced together in float2 registers.
//   wedm_step_regs<CELLS, L>: wires of at most 128 segments, uniform geometry: the wire lives in the registers of the L
//                      (1 or 2) lanes of its environment for the whole launch, as packed pairs of two virtual chunks; no
//                      LDS, halos between the two lanes by DPP, one wave-uniform mask per microsecond picks the tiles that
//                      need more than 88 packed operations.  The headline kernel (65 536 x 128: two lanes per environment).
//   wedm_reset_kernel: WireEDMEnv.reset for a masked subset.
// The packed / fused kernels exist in several instantiations (signal trace point, FROZEN_OK for autoreset handles,
// N1 / EXTRA for tile tables with one-change tiles or short tails): code that costs the other launches 1-2 % by its
// mere presence lives in its own instantiation, chosen per handle in plan_launch().
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math (see
// __graft_entry__.build()).  -ffp-contract=off is part of the numerics contract.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstddef>
#include <type_traits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>
#include <string>

#include "wedm_device.h"

using namespace wedm;

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

template <bool TRACE, bool F64, bool REPLAY>
__global__ void __launch_bounds__(256) wedm_step_global(const KArgs k) {
    const ColdRef cold = kernarg_cold();
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k.num_envs) return;
    Env s;
    load_env(cold, e, s);
    const bool reinit = s.done && WEDM_AUTORESET(cold);
    const bool frozen = s.done && !reinit && k.hot.done_value;  // terminated and not reset: nothing to step
    if (frozen && !WEDM_TRACING(k)) {
        if (WEDM_REWARD_ON(cold)) cold->s.reward[e] = 0.0f;  // a frozen environment earns nothing (not the previous launch's reward)
        return;
    }
    const GlobalT T = global_wire(cold->s.T, cold->s.stride, e);
    if (reinit) {  // next-step autoreset: wedm_reset for this environment, inside the launch
        reinit_env(cold, e, s, true);
        for (int q = 0; q < WEDM_T_QUADS(k.n_seg_max); ++q) T.stq(q, f4v{k.hot.spool, k.hot.spool, k.hot.spool, k.hot.spool});
    }
    unfreeze_wire(k.hot, s);  // keep_stepping_terminated: the DONE row is `terminated` of the last step and freezes nothing
    s.ipk = s.done ? 0.0 : peak_current(cold, s.mode, e);
    Geom g;
    load_geom(k.hot, cold, e, g);
    run_substeps<TRACE, F64, REPLAY>(k, cold, g, e, k.hot.env_id_offset + (uint32_t)e, s, T);
    if (WEDM_REWARD_ON(cold)) {
        if (!frozen) write_reward(cold, e, s);
        else cold->s.reward[e] = 0.0f;
    }
    store_time_hi(cold, e, s, (uint32_t)k.n_substeps * (uint32_t)k.hot.dt_us);
        store_env(cold, e, s);
}

// np.max over finite temperatures; maps to v_max_f32 / v_max3_f32
__device__ __forceinline__ float fmax_gt(float a, float b) { return __builtin_fmaxf(a, b); }

// phase stamps of the split kernel (diagnostic build -DWEDM_STAMPS only): raw s_memtime at
// [kernel entry, loop top, prelude done, barrier 1, walk done, barrier 2, loop exit, stored]
#ifdef WEDM_STAMPS
#define WEDM_SPLIT_STAMP_DECL unsigned long long sst[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sst[7])::"memory")
#define WEDM_SPLIT_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sst[i])::"memory"); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#define WEDM_SPLIT_STAMP_OUT() do { if (k.dbg && (threadIdx.x & 63) == 0) { \
    unsigned long long* o = k.dbg + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; \
    for (int q = 0; q < 8; ++q) o[q] = sst[q]; } } while (0)
#else
#define WEDM_SPLIT_STAMP_DECL do { } while (0)
#define WEDM_SPLIT_STAMP(i) do { } while (0)
#define WEDM_SPLIT_STAMP_OUT() do { } while (0)
#endif

// ===================================================== split global-memory kernel (1 us / launch)
// The reference's step() is ONE microsecond: every byte of T has to cross HBM once per launch.
// With one lane per environment (wedm_step_global) a lane walks the whole wire through a chain
// of dependent memory round trips (25 us even for a single block).  Here the wire is cut into
// QL = 4 chunks walked by four WAVES of a block (chunk-major thread layout: a wave = one chunk of
// 64 consecutive environments, every row access still a 256-B coalesced transaction).  Wave 0
// runs the scalar physics once per environment and publishes the stencil coefficients through
// LDS; the chunk maxima come back the same way.  T is updated in place: halos (OLD neighbour
// values) are read before the barrier that precedes the first store.  Any geometry (predicated
// cell).  Three barriers per microsecond.
#define WEDM_QL 4
#ifndef WEDM_SPLIT_RB
#define WEDM_SPLIT_RB 16
#endif
template <bool TRACE>
__global__ void __launch_bounds__(256) wedm_step_split(const KArgs k) {
    const ColdRef cold = kernarg_cold();
    __shared__ float sh_f[5][64];    // jf, q, conv_base, conv_zone, adv
    __shared__ int32_t sh_i[4][64];  // joule_on, pidx, adv_on, skip (environment frozen)
    __shared__ float sh_max[WEDM_QL][64];
    const int tid = threadIdx.x;
    const int c = tid >> 6, el = tid & 63;
    const int64_t e = (int64_t)blockIdx.x * 64 + el;
    const bool live = e < k.num_envs;
    const int64_t stride = cold->s.stride;
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    const uint32_t gid = k.hot.env_id_offset + (uint32_t)e;

    Geom g;
    load_geom(k.hot, cold, live ? e : 0, g);
    const int n = g.n_seg;
    // cells per wave: a multiple of 4, so that every wave's chunk starts on a 16-byte word of the quad-interleaved block
    const int C = ((k.n_seg_max + 4 * WEDM_QL - 1) / (4 * WEDM_QL)) * 4;
    const int i0 = c * C, i1 = (i0 + C < n) ? i0 + C : n;  // this lane's cells [i0, i1) (may be empty)
    const GlobalT T = global_wire(cold->s.T, stride, live ? e : 0);

    // next-step autoreset: every wave of the block sees the environment's DONE flag
    const bool reinit = live && WEDM_AUTORESET(cold) && cold->s.i8[(int64_t)WEDM_B_DONE * stride + e] != 0;
    if (reinit) {  // this lane's words of the wire (all of the block's rows, as wedm_reset does)
        const int qe = (i0 + C) >> 2, qn = WEDM_T_QUADS(k.n_seg_max);
        for (int q = i0 >> 2; q < (qe < qn ? qe : qn); ++q) T.stq(q, f4v{spool, spool, spool, spool});
    }
    Env s;
    Persist ps{0.0f, 0.0f, 0.0f, 0};
    bool frozen0 = true;
    if (c == 0) {
        if (live) load_env(cold, e, s);
        else s.done = WEDM_DEAD_LANE;
        if (reinit) reinit_env(cold, e, s, true);
        unfreeze_wire(k.hot, s);  // keep_stepping_terminated: nothing is frozen
        frozen0 = s.done;
        if (!s.done) {
            s.ipk = peak_current(cold, s.mode, e);
            init_persist(k.hot, cold, e, s, ps);
        }
    }
    int trace_next = k.trace_next, trace_slot = k.trace_slot;
    (void)trace_next; (void)trace_slot;
    WEDM_SPLIT_STAMP_DECL;

    for (int it = 0; it < k.n_substeps; ++it) {
        WEDM_SPLIT_STAMP(0);
        if (c == 0) {
            Coef cf{0.0f, 0.0f, 0, -1};
            if (!s.done) cf = scalar_prelude(k.hot, cold, g, e, gid, s, ps, true);
            sh_f[0][el] = cf.jf; sh_f[1][el] = cf.q; sh_f[2][el] = ps.conv_base; sh_f[3][el] = ps.conv_zone;
            sh_f[4][el] = ps.adv;
            sh_i[0][el] = cf.joule_on; sh_i[1][el] = cf.pidx; sh_i[2][el] = ps.adv_on;
            sh_i[3][el] = s.done | s.broken;  // (keep_stepping_terminated: a broken wire stays as it is, wire.py:260-261)
        }
        WEDM_SPLIT_STAMP(1);
        // OLD neighbour values, read before the barrier that precedes every store of this step
        float halo_l = spool, halo_r = 0.0f;
        if (live && i0 < i1) {
            if (i0 > 0) halo_l = T.ld(i0 - 1);
            if (i1 < n) halo_r = T.ld(i1);
            if (reinit && it == 0) { halo_l = spool; halo_r = spool; }  // the neighbour wave's fill may not have landed
        }
        __syncthreads();
        WEDM_SPLIT_STAMP(2);
        const Coef cf{sh_f[0][el], sh_f[1][el], sh_i[0][el], sh_i[1][el]};
        const Persist pw{sh_f[4][el], sh_f[2][el], sh_f[3][el], sh_i[2][el]};
        const bool skip = sh_i[3][el] != 0;
        float tmax = spool;
        if (live && !skip && i0 < i1) {
            // RB cells = RB / 4 sixteen-byte words per batch of loads, unconditional from a clamped word index (no branch
            // between them, all in flight together), plus the first cell after them (right neighbour of the batch's last).
            // Stamps show the walk phase itself moving ~7.7 TB/s chip-wide: what is left is the lock-step of the blocks
            // (all in the scalar phase, then all walking).
            constexpr int RB = WEDM_SPLIT_RB;
            static_assert(RB % 4 == 0, "a batch is a whole number of 16-byte words");
            float tm1 = halo_l;
            const int qlast = (i1 - 1) >> 2;
            for (int ib = i0; ib < i1; ib += RB) {
                float buf[RB + 1], tn[RB];
#pragma unroll
                for (int h = 0; h < RB / 4; ++h) {
                    int q = (ib >> 2) + h;
                    q = q < qlast ? q : qlast;  // past the chunk: any valid word, the values are not used
                    const f4v v = T.ldq(q);
#pragma unroll
                    for (int w = 0; w < 4; ++w) buf[4 * h + w] = v[w];
                }
                {
                    int idx = ib + RB;
                    idx = idx < i1 ? idx : i1 - 1;
                    buf[RB] = T.ld(idx);
                }
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int i = ib + u;
                    tn[u] = buf[u];
                    if (i < i1) {
                        const float tp1 = (i + 1 < i1) ? buf[u + 1] : halo_r;
                        tn[u] = (i >= 1) ? stencil_cell(i, n, (i == 1) ? spool : tm1, buf[u], tp1, g, cf, pw, tref, alpha, tdiel)
                                         : spool;
                        tmax = tn[u] > tmax ? tn[u] : tmax;
                        tm1 = buf[u];
                    }
                }
#pragma unroll
                for (int h = 0; h < RB / 4; ++h) {
                    const int iq = ib + 4 * h;
                    if (iq + 3 < i1) {
                        T.stq(iq >> 2, f4v{tn[4 * h], tn[4 * h + 1], tn[4 * h + 2], tn[4 * h + 3]});
                    } else {  // the wire's last, partial word: the cells past the end are padding and keep their value
#pragma unroll
                        for (int w = 0; w < 4; ++w)
                            if (iq + w < i1) T.st(iq + w, tn[4 * h + w]);
                    }
                }
            }
        }
        WEDM_SPLIT_STAMP(3);
        sh_max[c][el] = tmax;
        __syncthreads();
        WEDM_SPLIT_STAMP(4);
        if (c == 0 && !s.done) {
            float m = sh_max[0][el];
#pragma unroll
            for (int q = 1; q < WEDM_QL; ++q) m = fmax_gt(m, sh_max[q][el]);
            scalar_epilogue(k.hot, s, m);
            if (s.ctrl) control_step_outputs(cold, e, s, true);
        }
        if (TRACE && it == trace_next) {  // wave-uniform schedule; T rows of the step just finished
            const wedm_trace_desc& tr = k.trace;
            const int64_t tcol = live ? trace_column(tr, e) : -1;
            if (tcol >= 0) {
                if (c == 0) trace_scalars(tr, tcol, s, trace_slot, k.hot.done_value == 0);
                if (tr.T) {
                    const int64_t tcnt = tr.env_count;
                    float* tT = tr.T + (int64_t)trace_slot * k.n_seg_max * tcnt + tcol;
                    for (int i = i0; i < i1; ++i) tT[(int64_t)i * tcnt] = T.ld(i);
                }
            }
            trace_next += tr.every;
            trace_slot = (trace_slot + 1 == tr.capacity) ? 0 : trace_slot + 1;
        }
        if (it + 1 < k.n_substeps) __syncthreads();  // the next step's halo reads follow this step's stores
    }
    WEDM_SPLIT_STAMP(5);
    if (c == 0 && live) {
        if (WEDM_REWARD_ON(cold)) {
            if (!frozen0) write_reward(cold, e, s);
            else cold->s.reward[e] = 0.0f;  // a frozen environment earns nothing (not the previous launch's reward)
        }
        store_time_hi(cold, e, s, (uint32_t)k.n_substeps * (uint32_t)k.hot.dt_us);
        store_env(cold, e, s);
    }
    WEDM_SPLIT_STAMP(6);
    WEDM_SPLIT_STAMP_OUT();
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

// A wave that starts with a terminated (frozen) environment in a kernel instantiation without the frozen-lane tile
// code tells the host (Cold::frozen_seen, host-visible): the next launches of the handle take the FROZEN_OK instantiation.
#define WEDM_REPORT_FROZEN(cond)                                                      \
    do {                                                                              \
        if (!kFrozenOk && __any(cond)) {                                              \
            int32_t* const seen = cold->frozen_seen;                                  \
            if (seen && (threadIdx.x & 63) == 0) *seen = 1;                           \
        }                                                                             \
    } while (0)

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
template <int L, bool TRACE, int CMAX, bool ONE = false>
__global__ void __launch_bounds__(256, 2) wedm_step_stream(const KArgs k) {
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
                tpl = stencil_cell(cf.pidx, n, tm, tcc, tp, g, cf, ps, tref, alpha, tdiel);
            }
        }
        if (owns_last && !s.done) {
            const int jl = n - 1 - cbase;
            float tm = jl > 0 ? col[(jl - 1) * 256] : halo_l;
            if (n - 1 == 1) tm = spool;
            tlast = stencil_cell(n - 1, n, tm, col[jl * 256], 0.0f, g, cf, ps, tref, alpha, tdiel);
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
                } else if (!((slow_now >> t) & 1u)) {
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
                        float tn = interior_cell<true>(tm1, tc, tp1, g.k, g.tuf, conv, tdiel, ps.adv, jfe, alpha, tref);
                        if (!inter && valid) {  // boundary cells and irregular waves: predicated formula
                            const int i = cbase + jj;
                            tn = (i >= 1) ? stencil_cell(i, n, (i == 1) ? spool : tm1, tc, tp1, g, cf, ps, tref, alpha, tdiel)
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
        const uint32_t n_now = kind_n | kind_ne | (__any(cf.joule_on && cf.jf != 0.0f) ? 0u : kind_nj);
        // ---- patched cells: full predicated formula from OLD values, stored after the walk
        const bool owns_pl = cf.pidx >= 1 && cf.pidx >= cbase && cf.pidx < cbase + C;
        float tpl = 0.0f, tlast = 0.0f;
        if (__any(owns_pl)) {
            if (owns_pl) {
                const int jp = cf.pidx - cbase;
                float tm = jp > 0 ? old_at(jp - 1, 0) : halo_l;
                if (cf.pidx == 1) tm = spool;
                const float tp = jp < C - 1 ? old_at(jp + 1, 0) : halo_r;
                tpl = stencil_cell(cf.pidx, n, tm, old_at(jp, 0), tp, g, cf, ps, tref, alpha, tdiel);
            }
        }
        if (owns_last) {
            const int jl = n - 1 - cbase;
            float tm = jl > 0 ? old_at(jl - 1, 0) : halo_l;
            if (n - 1 == 1) tm = spool;
            tlast = stencil_cell(n - 1, n, tm, old_at(jl, 0), 0.0f, g, cf, ps, tref, alpha, tdiel);
        }
        float tmax = spool;
        const float jf_lane = cf.joule_on ? cf.jf : 0.0f;
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
                const float conv_lo = ((zone_lo >> t) & 1u) ? ps.conv_zone : ps.conv_base;
                const float jfe_lo = ((joule_lo >> t) & 1u) ? jf_lane : 0.0f;
                const uint32_t off = offc + (uint32_t)(j >> 2) * rowb;
                if ((n_now >> t) & 1u) {
                    cv[0] = f2{conv_lo, conv_lo}; jv[0] = f2{jfe_lo, jfe_lo};
                    if (joule_wave && __any(jfe_lo != 0.0f))
                        quad_staged<true, false>(tm, tc, tp, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    else
                        quad_staged<false, false>(tm, tc, tp, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
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
                    const float conv_hi = ((zone_hi >> t) & 1u) ? ps.conv_zone : ps.conv_base;
                    const float jfe_hi = ((joule_hi >> t) & 1u) ? jf_lane : 0.0f;
                    const uint32_t im1 = (uint32_t)(cbase + j - 1);  // (i - 1) of the tile's first cell
                    const uint32_t span = (uint32_t)(n - 3);         // interior <=> (i - 1) <= n - 3 (unsigned)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        cv[m] = f2{2 * m < split ? conv_lo : conv_hi, 2 * m + 1 < split ? conv_lo : conv_hi};
                        jv[m] = f2{2 * m < split ? jfe_lo : jfe_hi, 2 * m + 1 < split ? jfe_lo : jfe_hi};
                    }
                    if (joule_wave) quad_staged<true, true>(tm, tc, tp, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    else quad_staged<false, true>(tm, tc, tp, tn, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
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


// ============================================ register kernel: one environment per lane, the whole wire in VGPRs
// Wires of at most CELLS (128) segments, uniform geometry, float32 stencil, launches without a trace sample.
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
template <int CELLS, int L, bool TRACE = false>
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
    // the lane's coefficients (the quiet one never does)
    f2 convp[H / 8];
    auto build_conv = [&]() {
#pragma unroll
        for (int t = 0; t < H / 8; ++t)
            convp[t] = f2{((zoneA >> t) & 1u) ? ps.conv_zone : ps.conv_base, ((zoneB >> t) & 1u) ? ps.conv_zone : ps.conv_base};
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
        if (act) {  // (the lanes of terminated environments sit the walk out: their registers stay)
            // a wave with a negative plasma heat walks every cell on the predicated formula (identical results, slower)
            const bool all_slow = __any(cf.q < 0.0f);
            // regular tiles of THIS microsecond: a contact-flag change inside a tile only matters while current flows
            const float jf_lane = cf.joule_on ? cf.jf : 0.0f;
            const bool joule_wave = __any(jf_lane != 0.0f);
            const uint32_t n_now = all_slow ? 0u : (kind_n | kind_ne | (joule_wave ? 0u : kind_nj));
            // the tiles that hold some lane's plasma cell (a lane's own cells only)
            const int pcell = (cf.pidx >= 1 && cf.pidx >= base && cf.pidx < base + 2 * H) ? cf.pidx - base : -1;  // lane-local
            uint32_t ptiles = 0u;
            if (__any(pcell >= 0)) {
                const int pt = pcell >= 0 ? ((pcell & (H - 1)) >> 3) : -1;
#pragma unroll
                for (int t = 0; t < H / 8; ++t) ptiles |= __any(pt == t) ? (1u << t) : 0u;
            }
            // tiles that need more than the regular code without a Joule term
            const uint32_t general = ~n_now | (joule_wave ? joule_any : 0u) | ptiles | last_tile;
            f2 leftp = f2{halo_l, a_last};  // OLD pair before the tile
#pragma unroll
            for (int t = 0; t < H / 8; ++t) {
                if (t < n_tiles) {
                    const int j = 8 * t;
                    f2 tm[8], tc[8], tp[8], pn[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        tc[u] = P[j + u];
                        tm[u] = u == 0 ? leftp : P[j + u - 1];
                        tp[u] = (j + u + 1 < H) ? P[j + u + 1 < H ? j + u + 1 : 0] : f2{b_first, halo_r};
                    }
                    leftp = tc[7];
                    f2 cv[4], jv[4];
                    cv[0] = convp[t];
                    f2 tmA[4], tcA[4], tpA[4], pnA[4], tmB[4], tcB[4], tpB[4], pnB[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { tmA[u] = tm[u]; tcA[u] = tc[u]; tpA[u] = tp[u]; tmB[u] = tm[4 + u]; tcB[u] = tc[4 + u]; tpB[u] = tp[4 + u]; }
                    if (!((general >> t) & 1u)) {
                        jv[0] = f2{0.0f, 0.0f};
                        quad_staged<false, false, SW>(tmA, tcA, tpA, pnA, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                        quad_staged<false, false, SW>(tmB, tcB, tpB, pnB, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
#pragma unroll
                        for (int u = 0; u < 4; ++u) { pn[u] = pnA[u]; pn[4 + u] = pnB[u]; }
                        if (t == 0) pn[0].x = (c == 0) ? spool : pn[0].x;  // wire cell 0
                        float m0 = max3_raw(tmax, pn[0].x, pn[0].y), m1 = max3_raw(pn[1].x, pn[1].y, pn[2].x);
                        m0 = max3_raw(m0, pn[2].y, pn[3].x); m1 = max3_raw(m1, pn[3].y, pn[4].x);
                        m0 = max3_raw(m0, pn[4].y, pn[5].x); m1 = max3_raw(m1, pn[5].y, pn[6].x);
                        m0 = max3_raw(m0, pn[6].y, pn[7].x);
                        tmax = max3_raw(m0, m1, pn[7].y);
                    } else if (((n_now | (all_slow ? 0u : kind_nj)) >> t) & 1u) {
                        // regular, with odd cells: a Joule term, the wire's last cell, plasma cells
                        jv[0] = f2{((jouleA >> t) & 1u) ? jf_lane : 0.0f, ((jouleB >> t) & 1u) ? jf_lane : 0.0f};
                        if (!((n_now >> t) & 1u)) {
                            // a contact index inside the tile while current flows (kind_nj; the zone flag is uniform): the Joule
                            // coefficient cell by cell from the table
                            f2 jq[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const uint32_t zj = wt->zj[j + u];
                                jq[u] = f2{((zj >> (16 + 2 * c)) & 1u) ? jf_lane : 0.0f, ((zj >> (17 + 2 * c)) & 1u) ? jf_lane : 0.0f};
                            }
                            f2 jvA[4], jvB[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) { jvA[u] = jq[u]; jvB[u] = jq[4 + u]; }
                            quad_staged<true, true, SW>(tmA, tcA, tpA, pnA, g.k, g.tuf, cv4(cv[0]).v, tdiel, ps.adv, jvA, alpha, tref);
                            quad_staged<true, true, SW>(tmB, tcB, tpB, pnB, g.k, g.tuf, cv4(cv[0]).v, tdiel, ps.adv, jvB, alpha, tref);
                        } else if (joule_wave && ((joule_any >> t) & 1u)) {
                            quad_staged<true, false, SW>(tmA, tcA, tpA, pnA, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                            quad_staged<true, false, SW>(tmB, tcB, tpB, pnB, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                        } else {
                            quad_staged<false, false, SW>(tmA, tcA, tpA, pnA, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                            quad_staged<false, false, SW>(tmB, tcB, tpB, pnB, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { pn[u] = pnA[u]; pn[4 + u] = pnB[u]; }
                        if (t == 0) pn[0].x = (c == 0) ? spool : pn[0].x;  // wire cell 0
                        // the last cell (last position of the last chunk's tile): out of the regular maximum, predicated formula
                        const bool has_last = ((last_tile >> t) & 1u) && owns_last;
                        float m0 = max3_raw(tmax, pn[0].x, pn[0].y), m1 = max3_raw(pn[1].x, pn[1].y, pn[2].x);
                        m0 = max3_raw(m0, pn[2].y, pn[3].x); m1 = max3_raw(m1, pn[3].y, pn[4].x);
                        m0 = max3_raw(m0, pn[4].y, pn[5].x); m1 = max3_raw(m1, pn[5].y, pn[6].x);
                        m0 = max3_raw(m0, pn[6].y, pn[7].x);
                        tmax = max3_raw(m0, m1, has_last ? spool : pn[7].y);
                        if ((last_tile >> t) & 1u) {
                            const float x = stencil_cell(base + H + j + 7, nw, tm[7].y, tc[7].y, 0.0f, gw, cf, ps, tref, alpha, tdiel);
                            pn[7].y = has_last ? x : pn[7].y;
                            tmax = has_last ? fmax_gt(tmax, x) : tmax;
                        }
                        // plasma cells of the lanes that have one in this tile: the predicated formula from the same OLD values
                        // (the regular value stays in the maximum, as where the LDS kernels patch the cell after the walk)
                        if ((ptiles >> t) & 1u) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                if (__any(pcell == j + u)) {
                                    const float x = stencil_cell(base + j + u, nw, (base + j + u == 1) ? spool : tm[u].x, tc[u].x, tp[u].x, gw, cf, ps, tref, alpha, tdiel);
                                    pn[u].x = (pcell == j + u) ? x : pn[u].x;
                                    tmax = (pcell == j + u) ? fmax_gt(tmax, x) : tmax;
                                }
                                if (__any(pcell == H + j + u)) {
                                    const float x = stencil_cell(base + H + j + u, nw, tm[u].y, tc[u].y, tp[u].y, gw, cf, ps, tref, alpha, tdiel);
                                    pn[u].y = (pcell == H + j + u) ? x : pn[u].y;
                                    tmax = (pcell == H + j + u) ? fmax_gt(tmax, x) : tmax;
                                }
                            }
                        }
                    } else {
                        // not regular in this microsecond: every cell that exists on the predicated formula
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            pn[u] = tc[u];
                            const int ia = base + j + u, ib = base + H + j + u;
                            {
                                const float x = (ia >= 1) ? stencil_cell(ia, nw, (ia == 1) ? spool : tm[u].x, tc[u].x, tp[u].x, gw, cf, ps, tref, alpha, tdiel) : spool;
                                pn[u].x = ia < nw ? x : pn[u].x;
                                tmax = ia < nw ? fmax_gt(tmax, x) : tmax;
                            }
                            {
                                const float x = stencil_cell(ib, nw, tm[u].y, tc[u].y, tp[u].y, gw, cf, ps, tref, alpha, tdiel);
                                pn[u].y = ib < nw ? x : pn[u].y;
                                tmax = ib < nw ? fmax_gt(tmax, x) : tmax;
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) P[j + u] = pn[u];
                }
            }
        }
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
// Wires of up to 2 H L (512) segments, uniform geometry, float32 stencil.
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
template <int H, int L, bool CUT, bool TRACE = false>
__global__ void __launch_bounds__(256, WEDM_WIDE_MIN_BLOCKS) wedm_step_regs_wide(const KArgs k) {
    static_assert(H % 8 == 0 && H <= 32, "whole tiles");
    static_assert(L == 4 || L == 8 || L == 16, "the lanes of an environment lie in one DPP row");
    constexpr int EPB = 256 / L;
    constexpr int SW = WEDM_WIDE_SW;
    const ColdRef cold = kernarg_cold();
    Hot hv = k.hot;
    pin_hot_in_vgprs(hv);
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
    auto build_conv = [&]() {
#pragma unroll
        for (int m = 0; m < H; ++m)
            convc[m] = f2{((zoneA >> m) & 1u) ? ps.conv_zone : ps.conv_base, ((zoneB >> m) & 1u) ? ps.conv_zone : ps.conv_base};
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
            const float jf_lane = (!PLAIN && cf.joule_on) ? cf.jf : 0.0f;
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
                        quad_staged<true, true, SW>(tmA, tcA, tpA, pnA, g.k, g.tuf, cvA, tdiel, ps.adv, jvA, alpha, tref);
                        quad_staged<true, true, SW>(tmB, tcB, tpB, pnB, g.k, g.tuf, cvB, tdiel, ps.adv, jvB, alpha, tref);
                    } else {
                        quad_staged<false, true, SW>(tmA, tcA, tpA, pnA, g.k, g.tuf, cvA, tdiel, ps.adv, cvA, alpha, tref);
                        quad_staged<false, true, SW>(tmB, tcB, tpB, pnB, g.k, g.tuf, cvB, tdiel, ps.adv, cvB, alpha, tref);
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
                            const float x = stencil_cell(base + (last_in_b ? H : 0) + j + 7, nw, last_in_b ? tm[7].y : tm[7].x,
                                                         last_in_b ? tc[7].y : tc[7].x, 0.0f, gw, cw, ps, tref, alpha, tdiel);
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
                                    const float x = stencil_cell(base + j + u, nw, tm[u].x, tc[u].x, 0.0f, gw, cw, ps, tref, alpha, tdiel);
                                    pn[u].x = owns_last ? x : pn[u].x;
                                }
                                if (lw == H + j + u) {
                                    const float x = stencil_cell(base + H + j + u, nw, tm[u].y, tc[u].y, 0.0f, gw, cw, ps, tref, alpha, tdiel);
                                    pn[u].y = owns_last ? x : pn[u].y;
                                }
                            }
                        }
                        // plasma cells of the lanes that have one in this tile
                        if (!PLAIN && ((ptiles >> t) & 1u)) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                if (__any(pcell == j + u)) {
                                    const float x = stencil_cell(base + j + u, nw, (base + j + u == 1) ? spool : tm[u].x, tc[u].x, tp[u].x, gw, cf, ps, tref, alpha, tdiel);
                                    pn[u].x = (pcell == j + u) ? x : pn[u].x;
                                }
                                if (__any(pcell == H + j + u)) {
                                    const float x = stencil_cell(base + H + j + u, nw, tm[u].y, tc[u].y, tp[u].y, gw, cf, ps, tref, alpha, tdiel);
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


#include "wedm_served.h"
#include "wedm_lanes2.h"

// ------------------------------------------------------------ translation-unit parts (build time only)
// The packed and fused kernels exist in 32 and 40 instantiations and take hipcc two minutes in one translation unit.
// __graft_entry__.build_hip() compiles this file three times in parallel: -DWEDM_PART=1 emits the packed instantiations
// only, -DWEDM_PART=2 the fused ones, -DWEDM_PART=0 everything else (host code, the other kernels) with the two families
// declared `extern template`; the three objects link into the one shared library.  Without -DWEDM_PART the file is one
// self-contained translation unit (the diagnostic builds of tools/ use it that way).
#define WEDM_BOOLS3(X, L) X(L, false, false, false) X(L, false, false, true) X(L, false, true, false) X(L, false, true, true) \
                          X(L, true, false, false) X(L, true, false, true) X(L, true, true, false) X(L, true, true, true)
#define WEDM_PACKED_LIST(X) WEDM_BOOLS3(X, 1) WEDM_BOOLS3(X, 2) WEDM_BOOLS3(X, 4) WEDM_BOOLS3(X, 8)
#define WEDM_FUSED_LIST(X) WEDM_BOOLS3(X, 1) WEDM_BOOLS3(X, 2) WEDM_BOOLS3(X, 4) WEDM_BOOLS3(X, 8) WEDM_BOOLS3(X, 16)
#define WEDM_INST_PACKED(L, a, b, c) template __global__ void wedm_step_packed<L, a, b, c>(const KArgs);
#define WEDM_INST_FUSED(L, a, b, c) template __global__ void wedm_step_fused<L, a, b, c>(const KArgs);
// stencil_mode 1 on the tile walk: <L, TRACE, FROZEN_OK = true, N1 = false, F64 = true>
#define WEDM_FUSED_F64_LIST(X) X(1, false) X(1, true) X(2, false) X(2, true) X(4, false) X(4, true) X(8, false) X(8, true) X(16, false) X(16, true)
#define WEDM_INST_FUSED_F64(L, tr) template __global__ void wedm_step_fused<L, tr, true, false, true>(const KArgs);
#define WEDM_EXT_FUSED_F64(L, tr) extern template __global__ void wedm_step_fused<L, tr, true, false, true>(const KArgs);
#define WEDM_EXT_PACKED(L, a, b, c) extern template __global__ void wedm_step_packed<L, a, b, c>(const KArgs);
#define WEDM_EXT_FUSED(L, a, b, c) extern template __global__ void wedm_step_fused<L, a, b, c>(const KArgs);
// the served kernels (wedm_served.h): <L, EXTRA>
#define WEDM_SERVED_LIST(X) X(4, false) X(4, true) X(8, false) X(8, true)
#define WEDM_LANES_PK_LIST(X) X(1, false) X(1, true) X(2, false) X(2, true) X(4, false) X(4, true) X(8, false) X(8, true) X(16, false) X(16, true)
#define WEDM_INST_LANES_PK(L, tr) template __global__ void wedm_step_lanes_pk<L, tr>(const KArgs);
#define WEDM_EXT_LANES_PK(L, tr) extern template __global__ void wedm_step_lanes_pk<L, tr>(const KArgs);
#define WEDM_LANES_SERVED_LIST(X) X(4) X(8) X(16)
#define WEDM_INST_LANES_SERVED(L) template __global__ void wedm_step_lanes_served<L>(const KArgs);
#define WEDM_EXT_LANES_SERVED(L) extern template __global__ void wedm_step_lanes_served<L>(const KArgs);
#define WEDM_INST_SERVED(L, ex) template __global__ void wedm_step_served<L, ex>(const KArgs);
#define WEDM_EXT_SERVED(L, ex) extern template __global__ void wedm_step_served<L, ex>(const KArgs);
#if defined(WEDM_PART) && WEDM_PART == 1
WEDM_PACKED_LIST(WEDM_INST_PACKED)
#elif defined(WEDM_PART) && WEDM_PART == 2
WEDM_FUSED_LIST(WEDM_INST_FUSED)
WEDM_FUSED_F64_LIST(WEDM_INST_FUSED_F64)
#elif defined(WEDM_PART) && WEDM_PART == 3
WEDM_SERVED_LIST(WEDM_INST_SERVED)
WEDM_LANES_PK_LIST(WEDM_INST_LANES_PK)
WEDM_LANES_SERVED_LIST(WEDM_INST_LANES_SERVED)
#else
#if defined(WEDM_PART)
WEDM_PACKED_LIST(WEDM_EXT_PACKED)
WEDM_FUSED_LIST(WEDM_EXT_FUSED)
WEDM_FUSED_F64_LIST(WEDM_EXT_FUSED_F64)
WEDM_SERVED_LIST(WEDM_EXT_SERVED)
WEDM_LANES_PK_LIST(WEDM_EXT_LANES_PK)
WEDM_LANES_SERVED_LIST(WEDM_EXT_LANES_SERVED)
#endif

__global__ void __launch_bounds__(256)
wedm_reset_kernel(const wedm_params p, const wedm_state_ptrs s, int32_t num_envs, int32_t n_seg_max,
                  const uint8_t* mask, uint32_t key_lo, uint32_t key_hi, int32_t reseed) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= num_envs) return;
    if (mask && !mask[e]) return;
    const int64_t stride = s.stride;
    int32_t episode = *WEDM_ROW(s.i32, WEDM_I_EPISODE);
    int32_t klo = *WEDM_ROW(s.i32, WEDM_I_KEY_LO), khi = *WEDM_ROW(s.i32, WEDM_I_KEY_HI);
    // reset_semantics 1 = the reference's own reset(): a new EDMState only (wire_edm.py:106-114).  The rows that mirror what
    // its MODULE objects hold survive: `prev_accel` (mechanics.py:60), the debris volume and the flow / density caches
    // (dielectric.py:69-80), the convection cache and coefficients (wire.py:205,224), the short timers and the current
    // cache (ignition.py:75-81), the crater list and its statistics (material.py:133).
    const bool keep_modules = p.reset_semantics != 0 && !(reseed & WEDM_RESET_FRESH);
    constexpr uint32_t module_f64 = (1u << WEDM_F_PREV_ACCEL) | (1u << WEDM_F_DEBRIS_VOLUME) | (1u << WEDM_F_FLOW) |
                                    (1u << WEDM_F_LAST_GAP) | (1u << WEDM_F_LAST_DENSITY) | (1u << WEDM_F_WIRE_LAST_FLOW) |
                                    (1u << WEDM_F_H_BASE) | (1u << WEDM_F_H_ZONE);
    constexpr uint32_t module_i32 = (1u << WEDM_I_RANDOM_SHORT_REM) | (1u << WEDM_I_DEBRIS_SHORT_REM) | (1u << WEDM_I_SPARK_COUNT);
    constexpr uint32_t module_i8 = 1u << WEDM_B_MODE_CACHED;
    for (int f = 0; f < WEDM_F64_COUNT; ++f)
        if (!(keep_modules && ((module_f64 >> f) & 1u))) *WEDM_ROW(s.f64, f) = 0.0;
    for (int f = 0; f < WEDM_I32_COUNT; ++f)
        if (!(keep_modules && ((module_i32 >> f) & 1u))) *WEDM_ROW(s.i32, f) = 0;
    for (int f = 0; f < WEDM_I8_COUNT; ++f)
        if (!(keep_modules && ((module_i8 >> f) & 1u))) *WEDM_ROW(s.i8, f) = 0;
    if (s.stats && !keep_modules) {
        *WEDM_ROW(s.stats, WEDM_S_CRATER_SUM) = 0.0; *WEDM_ROW(s.stats, WEDM_S_CRATER_SUMSQ) = 0.0;
        *WEDM_ROW(s.stats, WEDM_S_CRATER_MIN) = __builtin_inf(); *WEDM_ROW(s.stats, WEDM_S_CRATER_MAX) = -__builtin_inf();
    }
    if (reseed & WEDM_RESET_RESEED) {
        *WEDM_ROW(s.i32, WEDM_I_EPISODE) = 0;
        *WEDM_ROW(s.i32, WEDM_I_KEY_LO) = (int32_t)key_lo;
        *WEDM_ROW(s.i32, WEDM_I_KEY_HI) = (int32_t)key_hi;
    } else {
        *WEDM_ROW(s.i32, WEDM_I_EPISODE) = episode + 1;
        *WEDM_ROW(s.i32, WEDM_I_KEY_LO) = klo;
        *WEDM_ROW(s.i32, WEDM_I_KEY_HI) = khi;
    }
    // state.current_mode = None: 0, or -1 where the surviving module's current cache names a mode (ignition.py:98-113:
    // None then resolves through default_current_mode instead of the fresh cache's 60 A)
    if (keep_modules && *WEDM_ROW(s.i8, WEDM_B_MODE_CACHED)) *WEDM_ROW(s.i32, WEDM_I_CURRENT_MODE) = -1;
    *WEDM_ROW(s.f64, WEDM_F_WORKPIECE_POS) = p.initial_gap;            // wire_edm.py:111
    *WEDM_ROW(s.f64, WEDM_F_TARGET_POS) = p.target_cutting_distance;   // wire_edm.py:112
    *WEDM_ROW(s.f64, WEDM_F_UNWIND_VEL) = 0.2;                         // state.py:55
    *WEDM_ROW(s.f64, WEDM_F_SPARK_Y) = __builtin_nan("");              // [0, None, 0]
    if (!keep_modules) {
        *WEDM_ROW(s.f64, WEDM_F_LAST_GAP) = -1.0;                      // dielectric.py:78
        *WEDM_ROW(s.f64, WEDM_F_LAST_DENSITY) = -1.0;                  // dielectric.py:79
    }
    const float spool = (float)p.spool_T;
    *WEDM_ROW(s.f64, WEDM_F_TMAX) = (double)spool;
    for (int q = 0; q < WEDM_T_QUADS(n_seg_max); ++q)  // wire.py:264-269 (whole 16-byte words: padding cells included)
        *(f4v*)(s.T + (((int64_t)q * stride + e) << 2)) = f4v{spool, spool, spool, spool};
    if (s.obs)
        for (int c = 0; c < p.obs_dim; ++c) s.obs[(int64_t)c * stride + e] = 0.0f;
    if (s.reward) s.reward[e] = 0.0f;
}

// Probe of the device math the physics relies on (test hook; see wedm_debug_math).
__global__ void wedm_debug_math_kernel(int32_t kind, const double* a, const double* b, double* out, int32_t n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b ? b[i] : 0.0;
    double r = 0.0;
    switch (kind) {
        case 0: r = portable_exp(x); break;
        case 1: r = portable_log(x); break;
        case 2: r = cube_cr(x); break;
        case 3: r = sqrt(x); break;
        case 4: r = py_floordiv(x, y); break;
        case 5: r = x / y; break;
        case 6: {  // x = time, y = env id, key fixed; all four step uniforms observable
            W4 w = philox4(0x12345678u, 0x9abcdef0u, (uint32_t)x, 3u, (uint32_t)y, 0u);
            r = u32_to_unit(w.x) + 2.0 * u32_to_unit(w.y) + 4.0 * u32_to_unit(w.z) + 8.0 * u32_to_unit(w.w);
            break;
        }
        case 7: r = philox_std_normal(0x12345678u, 0x9abcdef0u, (uint32_t)x, 3u, (uint32_t)y); break;
        case 8: r = (double)spark_cell_offset(x, y); break;
        default: break;
    }
    out[i] = r;
}

// Fills every CU's LDS with `value` (test hook; see wedm_debug_poison_lds): rows of the LDS image that a kernel never
// stages (cells past a wire's end) then hold a conspicuous value instead of whatever the previous kernel left there.
__global__ void __launch_bounds__(256) wedm_debug_poison_lds_kernel(float value, int32_t n_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < n_floats; i += 256) lds[i] = value;
    __syncthreads();
    if (lds[(threadIdx.x * 97) % n_floats] != value) __builtin_trap();  // keeps the stores observable
}

// =================================================================== C-ABI
struct LaunchPlan {
    bool valid = false;
    const void* fn = nullptr;
    int grid = 0;
    int block = 256;
    size_t lds = 0;
    const WalkTable* walk = nullptr;
    char name[160] = {0};
};

struct wedm_ctx {
    wedm_params p;
    int32_t num_envs = 0, n_seg_max = 0;
    int device = -1;
    bool bound = false, geom_bound = false;
    wedm_state_ptrs s{};
    wedm_geom_ptrs g{};
    void* tables_dev = nullptr;
    wedm_params* params_dev = nullptr;  // "cold" parameters, read through rare branches only
    Tables tb{};
    int32_t variant = 0;
    int32_t lanes = 0;                 // lanes per environment for the fused kernel (0 = auto)
    bool auto_prefers_packed = true;
    unsigned long long* dbg = nullptr; // diagnostic builds: phase stamp buffer
    int lds_limit = 0;
    WalkTable* walk_dev = nullptr;     // [11] tables for L = 1, 2, 4, 8, 16; the same with chunks of whole 16-byte words (stream kernel); two chunks of 64 cells (register kernel)
    bool walk_ok[5] = {false, false, false, false, false};
    int32_t walk_C[5] = {0, 0, 0, 0, 0};
    bool walk4_ok[5] = {false, false, false, false, false};
    bool walk_regs_ok = false;
    int32_t walk4_C[5] = {0, 0, 0, 0, 0};
    uint32_t walk_n1z = 0;             // bit i: table i has a one-change tile with a zone change (see WalkTable::kind_n1_mask)
    // signal trace (wedm_bind_trace): descriptor, microseconds stepped and samples written since the bind
    const double* replay = nullptr;    // wedm_bind_rng_replay
    int64_t replay_steps = 0;
    bool trace_on = false;
    wedm_trace_desc trace{};
    int64_t trace_us = 0, trace_count = 0;
    std::string err;
    std::string last_kernel;
    LaunchPlan plans[2][2][2];         // [single microsecond][trace point][frozen-lane tile code]: cached launch decisions
    int32_t* frozen_seen = nullptr;    // pinned host word the kernels set (Cold::frozen_seen), and its device alias
    int32_t* frozen_seen_dev = nullptr;
    const LaunchPlan* last_plan = nullptr;
    int32_t last_n_sub = 0;
    void invalidate_plans() { for (auto& a : plans) for (auto& b : a) for (auto& pl : b) pl.valid = false; }
};

static int32_t fail(wedm_ctx* ctx, int32_t code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}
static int32_t hip_fail(wedm_ctx* ctx, hipError_t e, const char* what) {
    return fail(ctx, WEDM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}


// Walk table for L lanes per environment (uniform geometry): for every chunk-local cell j
// which chunks have that cell inside the zone / between the contacts / interior / valid, and
// per 8-cell tile whether it needs the per-cell (SPECIAL) path.
// `align`: the chunk length is rounded up to a multiple of it (4 for the stream kernel, whose lanes load their chunk in
// 16-byte words of the quad-interleaved block: every chunk then starts on a word; chunks may end up partly or wholly
// past the wire's end, which the valid / interior masks express like any ragged tail).
static bool build_walk(const wedm_params& p, int L, WalkTable& t, int align = 1) {
    std::memset(&t, 0, sizeof(t));
    const int n = p.n_seg;
    const int C = ((n + L - 1) / L + align - 1) / align * align;
    if (C + 1 > WEDM_MAX_C) return false;  // +1: the halo row
    const int cb = p.contact_bottom, ct = p.contact_top, zs = p.az_start, ze = p.az_end;
    t.C = C;
    t.n_tiles = (C + 7) / 8;
    const uint16_t all = (uint16_t)((1u << L) - 1u);
    for (int j = 0; j < t.n_tiles * 8; ++j) {
        uint16_t zone = 0, joule = 0, inter = 0, valid = 0;
        for (int c = 0; c < L && j < C; ++c) {
            const int i = c * C + j;
            if (zs < ze && i >= zs && i < ze) zone |= (uint16_t)(1u << c);
            if (i >= cb && i <= ct) joule |= (uint16_t)(1u << c);
            if (i >= 1 && i <= n - 2) inter |= (uint16_t)(1u << c);
            if (i < n) valid |= (uint16_t)(1u << c);
        }
        t.zj[j] = (uint32_t)zone | ((uint32_t)joule << 16);
        t.iv[j] = (uint32_t)inter | ((uint32_t)valid << 16);
    }
    for (int tile = 0; tile < t.n_tiles; ++tile) {
        const int j0 = 8 * tile, j1 = std::min(j0 + 8, C);
        bool all_interior = (j1 - j0 == 8);
        int changes = 0, split = 8;
        for (int j = j0; j < j1; ++j) {
            if ((t.iv[j] & 0xffffu) != all) all_interior = false;
            if (j > j0 && t.zj[j] != t.zj[j - 1]) { ++changes; split = j - j0; }
        }
        // regular apart from the wire's two end cells / apart from the contact flag?
        bool ends_only = (j1 - j0 == 8);
        int zone_changes = 0;
        for (int j = j0; j < j1; ++j) {
            uint16_t ends = 0;
            for (int c = 0; c < L; ++c) {
                const int i = c * C + j;
                if (i == 0 || i == n - 1) ends |= (uint16_t)(1u << c);
            }
            if ((t.iv[j] >> 16) != all) ends_only = false;                          // a cell past the wire's end
            if ((uint16_t)((t.iv[j] & 0xffffu) | ends) != all) ends_only = false;   // non-interior and not an end cell
            if (j > j0 && (t.zj[j] & 0xffffu) != (t.zj[j - 1] & 0xffffu)) ++zone_changes;
        }
        // (an end cell inside a full tile sits at its first / last position: cell 0 is j = 0 of chunk 0, and cell n-1
        // can only be followed by cells past the wire's end, which a tile with ends_only does not have)
        if (n < 2) ends_only = false;
        if (ends_only && !all_interior && changes == 0) t.kind_ne_mask |= 1u << tile;
        if (ends_only && zone_changes == 0 && changes > 0) t.kind_nj_mask |= 1u << tile;
        if (ends_only && changes == 1) t.kind_n1_mask |= (1u << tile) | (zone_changes == 1 ? 0x80000000u : 0u);
        // cells past the chunk keep the last real cell's flags so that zj[8t+7] is the tile's "hi" set
        for (int j = j1; j < j0 + 8; ++j) t.zj[j] = t.zj[j1 - 1];
        t.split[tile] = (uint32_t)split;
        if (changes > 1) t.kind[tile] = TILE_S;
        else if (all_interior && changes == 0) t.kind[tile] = TILE_N;
        else t.kind[tile] = TILE_B;
    }
    for (int tile = 0; tile < t.n_tiles; ++tile) {
        const uint32_t lo = t.zj[8 * tile], hi = t.zj[8 * tile + 7];
        for (int c = 0; c < 16; ++c) {
            t.chunk_flags[c][0] |= ((lo >> c) & 1u) << tile;
            t.chunk_flags[c][1] |= ((lo >> (16 + c)) & 1u) << tile;
            t.chunk_flags[c][2] |= ((hi >> c) & 1u) << tile;
            t.chunk_flags[c][3] |= ((hi >> (16 + c)) & 1u) << tile;
        }
        t.kind_n_mask |= (t.kind[tile] == TILE_N ? 1u : 0u) << tile;
        t.kind_s_mask |= (t.kind[tile] == TILE_S ? 1u : 0u) << tile;
        t.split_pack[tile >> 3] |= (t.split[tile] & 15u) << ((tile & 7) * 4);
    }
    return true;
}

template <bool TR, bool F64> static const void* pick_lanes(int L) {
    switch (L) {
        case 1: return (const void*)wedm_step_lanes<1, TR, F64>;
        case 2: return (const void*)wedm_step_lanes<2, TR, F64>;
        case 4: return (const void*)wedm_step_lanes<4, TR, F64>;
        case 8: return (const void*)wedm_step_lanes<8, TR, F64>;
        default: return (const void*)wedm_step_lanes<16, TR, F64>;
    }
}
template <bool TR, bool FZ, bool N1> static const void* pick_fused(int L) {
    switch (L) {
        case 1: return (const void*)wedm_step_fused<1, TR, FZ, N1>;
        case 2: return (const void*)wedm_step_fused<2, TR, FZ, N1>;
        case 4: return (const void*)wedm_step_fused<4, TR, FZ, N1>;
        case 8: return (const void*)wedm_step_fused<8, TR, FZ, N1>;
        default: return (const void*)wedm_step_fused<16, TR, FZ, N1>;
    }
}
template <bool TR, bool FZ> static const void* pick_fused(int L, bool n1) {
    return n1 ? pick_fused<TR, FZ, true>(L) : pick_fused<TR, FZ, false>(L);
}
template <bool TR> static const void* pick_fused_f64(int L) {
    switch (L) {
        case 1: return (const void*)wedm_step_fused<1, TR, true, false, true>;
        case 2: return (const void*)wedm_step_fused<2, TR, true, false, true>;
        case 4: return (const void*)wedm_step_fused<4, TR, true, false, true>;
        case 8: return (const void*)wedm_step_fused<8, TR, true, false, true>;
        default: return (const void*)wedm_step_fused<16, TR, true, false, true>;
    }
}
// rows a lane of the stream kernel holds in registers: 64 (128 segments over 2 lanes, 400 over 8) or 104 (400 over 4)
template <bool TR, int CMAX, bool ONE = false> static const void* pick_stream(int L) {
    switch (L) {
        case 1: return (const void*)wedm_step_stream<1, TR, CMAX, ONE>;
        case 2: return (const void*)wedm_step_stream<2, TR, CMAX, ONE>;
        case 4: return (const void*)wedm_step_stream<4, TR, CMAX, ONE>;
        case 8: return (const void*)wedm_step_stream<8, TR, CMAX, ONE>;
        default: return (const void*)wedm_step_stream<16, TR, CMAX, ONE>;
    }
}
template <bool TR, bool FZ, bool EX> static const void* pick_packed(int L) {
    switch (L) {
        case 1: return (const void*)wedm_step_packed<1, TR, FZ, EX>;
        case 2: return (const void*)wedm_step_packed<2, TR, FZ, EX>;
        case 4: return (const void*)wedm_step_packed<4, TR, FZ, EX>;
        default: return (const void*)wedm_step_packed<8, TR, FZ, EX>;
    }
}
template <bool TR, bool FZ> static const void* pick_packed(int L, bool extra) {
    return extra ? pick_packed<TR, FZ, true>(L) : pick_packed<TR, FZ, false>(L);
}

template <bool TR> static const void* pick_lanes_pk(int L) {
    switch (L) {
        case 1: return (const void*)wedm_step_lanes_pk<1, TR>;
        case 2: return (const void*)wedm_step_lanes_pk<2, TR>;
        case 4: return (const void*)wedm_step_lanes_pk<4, TR>;
        case 8: return (const void*)wedm_step_lanes_pk<8, TR>;
        default: return (const void*)wedm_step_lanes_pk<16, TR>;
    }
}
static const void* pick_served(int L, bool extra) {
    switch (L) {
        case 4: return extra ? (const void*)wedm_step_served<4, true> : (const void*)wedm_step_served<4, false>;
        default: return extra ? (const void*)wedm_step_served<8, true> : (const void*)wedm_step_served<8, false>;
    }
}

// A handle belongs to the device that was current in wedm_create: its parameter / table / walk buffers
// live there and its launches must go to a stream of that device.  Launching with another device
// current would hand hipLaunchKernel a foreign stream (hipErrorInvalidResourceHandle at best).
static int32_t check_device(wedm_ctx* ctx, const char* who) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipGetDevice");
    if (dev != ctx->device)
        return fail(ctx, WEDM_ERR_BAD_ARG, std::string(who) + ": handle was created on device " + std::to_string(ctx->device) +
                                               " but device " + std::to_string(dev) + " is current (hipSetDevice first)");
    return WEDM_OK;
}

static int lanes_index(int L) { return L == 1 ? 0 : L == 2 ? 1 : L == 4 ? 2 : L == 8 ? 3 : L == 16 ? 4 : -1; }

// What wedm_step launches for (single microsecond?, trace point?) under the handle's current settings: decided once
// and cached (the decision walks a cost model over five lane counts; on the one-launch-per-microsecond path that and
// a hipFuncSetAttribute per call were a measurable part of the host time per launch).
static int32_t plan_launch(wedm_ctx* ctx, bool single, bool tr, bool frozen_ok, LaunchPlan& out) {
    const wedm_params& P = ctx->p;
    // kernel 3 (one chunk per lane) and kernel 4 (two packed chunks per lane, table of 2L chunks).
    // Auto-selection by a small cost model fitted to measurements:
    //   cycles per step ~ rounds * (4500 + tiles_per_lane * 8 * cell_cost),  tiles_per_lane: see eff_tiles below,
    //   rounds = ceil(blocks / (256 CUs * resident blocks per CU)), resident = min(2 [VGPRs], LDS fit),
    //   cell_cost = 90 per cell, a packed pair = 2 * 90 * 0.93.
    const bool uniform = !ctx->p.per_env_geometry && ctx->walk_dev;
    int lanes = ctx->lanes, planes = ctx->lanes;
    // tiles a chunk of C cells costs: its full tiles, a whole tile for a partial one, a quarter for a 1- / 2-cell tail
    // (computed with the patched cells) -- 32 768 x 400: fused<8> 3.60e9, packed<8> 3.75e9 measured
    auto eff_tiles = [](int C) -> double {
        const int rest = C & 7;
        return (double)(C / 8) + (rest == 0 ? 0.0 : (C > 8 && rest <= 2) ? 0.25 : 1.0);
    };
    double best_lds_cost = 1e300;  // cycles per microsecond of the whole batch on the better of kernels 3 / 4, by the model above
    {
        double best3 = 1e300, best4 = 1e300;
        int l3 = 0, l4 = 0;
        const int Ls[5] = {1, 2, 4, 8, 16};
        for (int i = 0; i < 5 && uniform; ++i) {
            const int Lc = Ls[i];
            const long blocks = (ctx->num_envs + (256 / Lc) - 1) / (256 / Lc);
            if (ctx->walk_ok[i]) {  // kernel 3 with Lc lanes: table i
                const size_t lds = ((size_t)ctx->walk_C[i] + 1) * 1024;
                if (lds <= (size_t)ctx->lds_limit) {
                    const long rb = std::min<long>(2, (long)(160 * 1024 / lds));
                    const long rounds = (blocks + 256 * rb - 1) / (256 * rb);
                    const double cost = rounds * (4500.0 + eff_tiles(ctx->walk_C[i]) * 8 * 90.0);
                    if (cost < best3) { best3 = cost; l3 = Lc; }
                }
            }
            if (Lc <= 8 && ctx->walk_ok[lanes_index(2 * Lc)]) {  // kernel 4 with Lc lanes: table of 2*Lc chunks
                const int ti = lanes_index(2 * Lc);
                const size_t lds = (2 * (size_t)ctx->walk_C[ti] + 2) * 1024;
                if (lds <= (size_t)ctx->lds_limit) {
                    const long rb = std::min<long>(2, (long)(160 * 1024 / lds));
                    const long rounds = (blocks + 256 * rb - 1) / (256 * rb);
                    const double cost = rounds * (4500.0 + eff_tiles(ctx->walk_C[ti]) * 8 * 2 * 90.0 * 0.93);
                    if (cost < best4) { best4 = cost; l4 = Lc; }
                }
            }
        }
        if (!lanes) lanes = l3;
        if (!planes) planes = l4;
        ctx->auto_prefers_packed = best4 <= best3;
        best_lds_cost = std::min(best3, best4);
    }
    const int li = lanes_index(lanes);
    const bool fused_ok = uniform && li >= 0 && ctx->walk_ok[li] &&
                          ((size_t)ctx->walk_C[li] + 1) * 1024 <= (size_t)ctx->lds_limit;
    const int pli = (planes >= 1 && planes <= 8) ? lanes_index(2 * planes) : -1;
    const bool packed_ok = uniform && pli >= 0 && ctx->walk_ok[pli] &&
                           (2 * (size_t)ctx->walk_C[pli] + 2) * 1024 <= (size_t)ctx->lds_limit;
    // kernel 2 (any geometry): lanes per environment = the caller's choice, else the smallest L whose
    // chunk fits in LDS, raised until the launch has ~2 waves per SIMD
    int glanes = 0;
    {
        const int Ls[5] = {1, 2, 4, 8, 16};
        for (int i = 0; i < 5; ++i) {
            const size_t b = (size_t)((ctx->n_seg_max + Ls[i] - 1) / Ls[i]) * 1024;
            if (b > (size_t)ctx->lds_limit) continue;
            if (ctx->lanes) { if (Ls[i] == ctx->lanes) glanes = Ls[i]; continue; }
            glanes = Ls[i];
            const long waves = (long)((ctx->num_envs + (256 / Ls[i]) - 1) / (256 / Ls[i])) * 4;
            if (waves >= 2048) break;
        }
    }
    const bool lanes_ok = glanes > 0;
    // ... and its packed form (wedm_step_lanes_pk, float32 stencil): two virtual chunks of ceil(n_seg_max / 2L) cells per lane
    int pklanes = 0;
    {
        const int Ls[5] = {1, 2, 4, 8, 16};
        for (int i = 0; i < 5; ++i) {
            const size_t b = (2 * (size_t)((ctx->n_seg_max + 2 * Ls[i] - 1) / (2 * Ls[i])) + 2) * 1024;
            if (b > (size_t)ctx->lds_limit) continue;
            if (ctx->lanes) { if (Ls[i] == ctx->lanes) pklanes = Ls[i]; continue; }
            pklanes = Ls[i];
            const long waves = (long)((ctx->num_envs + (256 / Ls[i]) - 1) / (256 / Ls[i])) * 4;
            if (waves >= 2048) break;
        }
    }
    const bool lanes_pk_ok = pklanes > 0;
    // ... and the served form of that (wedm_step_lanes_served: 4, 8 or 16 lanes per environment, three walker waves + the scalar
    // wave per block, three blocks per CU where the LDS image allows): the caller's lane count, else the fewest lanes whose
    // blocks fill the chip at three per CU
    int svgl = 0;
    {
        const int Ls[3] = {4, 8, 16};
        for (int i = 0; i < 3; ++i) {
            const size_t b = (2 * (size_t)((ctx->n_seg_max + 2 * Ls[i] - 1) / (2 * Ls[i])) + 2) * 768 + sizeof(ServedBox<48>);
            if (b > (size_t)ctx->lds_limit) continue;
            if (ctx->lanes) { if (Ls[i] == ctx->lanes) svgl = Ls[i]; continue; }
            svgl = Ls[i];
            const long blocks = (ctx->num_envs + (192 / Ls[i]) - 1) / (192 / Ls[i]);
            if (blocks >= 768 && 3 * b <= 160 * 1024) break;
        }
    }
    const bool lanes_sv_ok = svgl > 0 && !ctx->replay && P.stencil_mode == 0 && !P.keep_stepping_terminated;
    // kernel 6 (stream, single microseconds, uniform geometry): the caller's lane count, else -- among the L whose chunk
    // has at most 64 cells (the registers a lane holds its chunk in) -- the largest one whose blocks are all resident at
    // once (2 048 waves): a launch of one microsecond is one dependent chain per wave, and a shorter chunk is a shorter
    // chain (4 096 x 400: 26.3 / 18.8 / 14.4 us with 4 / 8 / 16 lanes); a batch too large for one round takes the
    // smallest such L (65 536 x 128: 2 lanes); failing all that a chunk of at most 104 cells
    int slanes = 0;
    if (uniform && (uint64_t)WEDM_T_QUADS(ctx->n_seg_max) * (uint64_t)ctx->s.stride * 16ull < (1ull << 32)) {
        const int Ls[5] = {1, 2, 4, 8, 16};
        for (int pass = 0; pass < 2 && !slanes; ++pass)
            for (int i = 0; i < 5; ++i) {
                if (!ctx->walk4_ok[i] || ctx->walk4_C[i] > (pass ? 104 : 64) ||
                    ((size_t)ctx->walk4_C[i] + 1) * 1024 > (size_t)ctx->lds_limit) continue;
                if (ctx->lanes && Ls[i] != ctx->lanes) continue;
                const long waves = (long)((ctx->num_envs + (256 / Ls[i]) - 1) / (256 / Ls[i])) * 4;
                if (slanes && (pass || waves > 2048)) break;
                slanes = Ls[i];
            }
    }
    const bool stream_ok = slanes > 0;
    const bool stream_auto = stream_ok && ctx->walk4_C[lanes_index(slanes)] <= 64 &&
                             (long)((ctx->num_envs + (256 / slanes) - 1) / (256 / slanes)) * 4 <= 2048;
    int variant = ctx->variant;
    if (ctx->replay) {
        if (variant != 0 && variant != 1)
            return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: injected variates (wedm_bind_rng_replay) run on kernel 1 only");
        if (P.stencil_mode != 0)
            return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: injected variates and stencil_mode 1 cannot be combined");
        variant = 1;
    }
    const bool f64 = P.stencil_mode != 0;
    if (f64) {
        // Numba's typing of the stencil: the fused tile walk (uniform geometry), the predicated LDS kernel (any geometry),
        // or in place in global memory; no packed form, no single-microsecond kernels
        if (variant != 0 && variant != 1 && variant != 2 && variant != 3 && variant != 10)
            return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: stencil_mode 1 (float64 stencil expressions) runs on kernels 1, 2 (10) and 3 only");
        if (variant == 0) variant = (!single && fused_ok) ? 3 : (lanes_ok ? 2 : 1);
    }
    // kernel 2 is the packed form where it applies (float32 stencil, no injected variates); kernel 10 names the cell-by-cell
    // form explicitly (A/B timing, tests), which also serves stencil_mode 1
    const bool use_pk = !f64 && !ctx->replay && lanes_pk_ok;
    // kernel 8 (wide register kernel): 4, 8 or 16 lanes per environment (the fewest that hold the wire), 32 cells each in
    // registers; uniform geometry, float32 stencil, at most 512 segments.  Chosen by itself for a batch
    // that one round of blocks covers at one wave per
    // SIMD: such a launch is one wave's dependent chain per microsecond whatever the kernel, and this one's is the
    // shortest (measured, 4 096 x 400 and 16 384 x 128: DESIGN.md 4.1b)
    const int wl_min = P.n_seg <= 128 ? 4 : P.n_seg <= 256 ? 8 : 16;
    const int wl = (ctx->lanes == 4 || ctx->lanes == 8 || ctx->lanes == 16) ? ctx->lanes : wl_min;
    const bool wide_ok = uniform && P.n_seg >= 9 && P.n_seg <= 512 && !f64 && !ctx->replay &&
                         (ctx->lanes == 0 || (wl == ctx->lanes && wl >= wl_min));
    if (variant == 0 && !single && wide_ok && ctx->lanes == 0 &&
        (int64_t)ctx->num_envs * wl <= (int64_t)WEDM_WIDE_AUTO_MAX_LANES)
        variant = 8;
    if (variant == 8 && !wide_ok)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: wide register kernel needs uniform geometry, 9 to 512 segments, the float32 stencil and lanes 0, 4, 8 or 16 with 32 cells per lane covering the wire");
    // kernel 7 (register kernel): one or two lanes per environment with the wire in their registers; wires of at most 128
    // segments, uniform geometry, float32 stencil; a launch with a trace sample runs its TRACE instantiation
    const bool regs_ok = uniform && ctx->walk_regs_ok && ctx->n_seg_max <= 128 && !f64 && !ctx->replay;
    if (variant == 0) {
        // fused launches of a batch that gives most CUs a block of the register kernel (measured, 128 segments, two lanes
        // per environment against the best LDS kernel: 8 192 environments 2.8e9 vs 3.5e9, 16 384: 5.5e9 vs 6.1e9,
        // 24 576: 8.3e9 vs 7.4e9, 32 768: 1.10e10 vs 9.9e9, 65 536: 1.67e10 vs 1.44e10, 131 072: 1.76e10 vs 1.50e10;
        // up to 16 384 environments the wide register kernel above has taken the launch: 8.1e9 there)
        if (!single && regs_ok && ctx->lanes == 0 && ctx->num_envs >= 20480) variant = 7;
    }
    // kernel 9 (served packed kernel, wedm_served.h): the packed walk on three waves of a block, the scalar physics on the fourth;
    // 4 or 8 lanes per environment; no trace point and no keep_stepping_terminated (such launches stay on kernel 4).
    // Cost model (cycles per microsecond of the whole batch, same unit as the model of kernels 3 / 4; fitted to
    // profiles/r4/plan_sweep.txt): a block's chain c = 1300 + 950 x tiles per lane; j blocks resident together on a CU take
    // c x f(j), f = 1, 1.49, 1.80 (three fit the 168-register budget, fewer where the LDS image is large); blocks are
    // dispatched as CUs free up, so the busiest CU runs b = ceil(blocks / 256) of them in groups of at most `rb`.
    auto served_cost = [&](int L) -> double {
        const int ti = lanes_index(2 * L);
        if (ti < 0 || !ctx->walk_ok[ti]) return 1e300;
        const size_t lds = (2 * (size_t)ctx->walk_C[ti] + 2) * 768 + (L == 8 ? sizeof(ServedBox<24>) : sizeof(ServedBox<48>));
        if (lds > (size_t)ctx->lds_limit) return 1e300;
        const long rb = std::min<long>(3, (long)(160 * 1024 / lds));
        const long blocks = (ctx->num_envs + (192 / L) - 1) / (192 / L);
        const long b = (blocks + 255) / 256;
        static const double f[4] = {0.0, 1.0, 1.49, 1.80};
        // (a partial tile of 3 ... 7 cells runs the boundary-tile code for every lane: two tiles' worth -- 16 384 x 200 over 8 lanes,
        // chunks of 13 cells: 5.3 ms against 3.4 ms over 4 lanes)
        const int Cv = ctx->walk_C[ti], rest = Cv & 7;
        const double tiles = eff_tiles(Cv) + ((rest >= 3 || (rest && Cv < 8)) ? 1.0 : 0.0);
        const double c = 1300.0 + 950.0 * tiles;
        return (double)(b / rb) * c * f[rb] + ((b % rb) ? c * f[b % rb] : 0.0);
    };
    const double sv_cost4 = served_cost(4), sv_cost8 = served_cost(8);
    const int svl = (ctx->lanes == 4 || ctx->lanes == 8) ? ctx->lanes : (sv_cost4 < sv_cost8 ? 4 : 8);
    const int svi = lanes_index(2 * svl);
    const size_t sv_box = svl == 8 ? sizeof(ServedBox<24>) : sizeof(ServedBox<48>);  // three walker waves: 24 / 48 environments per block
    const bool served_ok = uniform && !f64 && !ctx->replay && !P.keep_stepping_terminated && (ctx->lanes == 0 || ctx->lanes == svl) &&
                           svi >= 0 && ctx->walk_ok[svi] &&
                           (2 * (size_t)ctx->walk_C[svi] + 2) * 768 + sv_box <= (size_t)ctx->lds_limit;
    // the served kernel where its model beats what the choice so far would take (measured over 2 048 ... 131 072 environments x
    // 128 ... 512 segments, profiles/r4/plan_sweep.txt: blocks of 24 / 48 environments, three to a CU, fill the chip where
    // blocks of 32 ... 128 leave a ragged second round, and a sixth fewer instructions)
    if (!single && !tr && served_ok && ctx->lanes == 0 && P.n_seg <= 512 /* the range the model was fitted on */ &&
        (variant == 0 || (variant == 7 && ctx->variant == 0))) {
        const double sv = std::min(sv_cost4, sv_cost8);
        double other = best_lds_cost;
        if (variant == 7) {  // the two-lane register kernel: 128 environments per block, two blocks per CU (6 050 / 7 800 cycles)
            const long b = ((ctx->num_envs + 127) / 128 + 255) / 256;
            other = (double)(b / 2) * 7800.0 + (double)(b % 2) * 6050.0;
        }
        if (sv < other) variant = 9;
    }
    if (variant == 0) {
        // single-microsecond launches: the stream kernel where one round of blocks covers the batch with chunks of
        // at most 64 cells (measured: 27.5 vs 30.3 us at 65 536 x 128, 20.5 vs 24.9 us at 4 096 x 400), else the
        // split global-memory kernel (32.7 vs 48.9 us at 32 768 x 400, where the stream kernel needs two rounds)
        if (single) variant = (stream_ok && stream_auto) ? 6 : 5;
        else if (packed_ok && (ctx->auto_prefers_packed || !fused_ok)) variant = 4;
        else if (fused_ok) variant = 3;
        else variant = (lanes_ok || use_pk) ? 2 : 1;
    }
    if (variant == 9 && !served_ok)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: served kernel needs uniform geometry, the float32 stencil, lanes 4 or 8, two chunks that fit in LDS and freeze_terminated");
    if (variant == 9 && tr) variant = packed_ok ? 4 : fused_ok ? 3 : (lanes_ok || use_pk) ? 2 : 1;
    if (variant == 7 && !regs_ok)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: register kernel needs uniform geometry, at most 128 segments and the float32 stencil");
    if (variant == 3 && !fused_ok)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: fused kernel needs uniform geometry and a chunk that fits in LDS");
    if (variant == 4 && !packed_ok)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: packed kernel needs uniform geometry, lanes in {1,2,4,8} and two chunks that fit in LDS");
    if (variant == 11 && (!lanes_sv_ok || tr)) variant = 2;  // (a trace sample, stencil_mode 1, keep-stepping: the unserved forms)
    if ((variant == 2 && !use_pk && !lanes_ok) || (variant == 10 && !lanes_ok))
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: no lane count puts a chunk of the wire in LDS");
    if (variant == 6 && !stream_ok)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: stream kernel needs uniform geometry and lanes in {1,2,4,8,16} with a chunk of at most 104 cells");

    const void* fn = nullptr;
    int grid = 0;
    size_t fl = 0;
    out.walk = nullptr;
    if (variant == 1) {
        grid = (ctx->num_envs + 255) / 256;
        fn = ctx->replay ? (tr ? (const void*)wedm_step_global<true, false, true> : (const void*)wedm_step_global<false, false, true>)
           : f64 ? (tr ? (const void*)wedm_step_global<true, true, false> : (const void*)wedm_step_global<false, true, false>)
                 : (tr ? (const void*)wedm_step_global<true, false, false> : (const void*)wedm_step_global<false, false, false>);
        std::snprintf(out.name, sizeof(out.name), "wedm_step_global%s<<<%d,256>>>", ctx->replay ? "[injected variates]" : f64 ? "[f64 stencil]" : "", grid);
    } else if (variant == 7) {
        const int rl = ctx->lanes == 1 ? 1 : 2;  // lanes per environment (default 2: two waves per SIMD)
        grid = (ctx->num_envs + 256 / rl - 1) / (256 / rl);
        out.walk = ctx->walk_dev + (rl == 1 ? 10 : 11);  // two chunks of 64 cells / four of 32
        fn = tr ? (rl == 1 ? (const void*)wedm_step_regs<128, 1, true> : (const void*)wedm_step_regs<128, 2, true>)
                : (rl == 1 ? (const void*)wedm_step_regs<128, 1> : (const void*)wedm_step_regs<128, 2>);
        std::snprintf(out.name, sizeof(out.name), "wedm_step_regs<%d><<<%d,256>>>", rl, grid);
    } else if (variant == 8) {
        grid = (ctx->num_envs + 256 / wl - 1) / (256 / wl);
        fn = tr ? (wl == 4 ? (const void*)wedm_step_regs_wide<16, 4, true, true> : wl == 8 ? (const void*)wedm_step_regs_wide<16, 8, true, true>
                                                                                       : (const void*)wedm_step_regs_wide<16, 16, true, true>)
           : (P.n_seg & 7) ? (wl == 4 ? (const void*)wedm_step_regs_wide<16, 4, true> : wl == 8 ? (const void*)wedm_step_regs_wide<16, 8, true>
                                                                                             : (const void*)wedm_step_regs_wide<16, 16, true>)
                           : (wl == 4 ? (const void*)wedm_step_regs_wide<16, 4, false> : wl == 8 ? (const void*)wedm_step_regs_wide<16, 8, false>
                                                                                              : (const void*)wedm_step_regs_wide<16, 16, false>);
        std::snprintf(out.name, sizeof(out.name), "wedm_step_regs_wide<%d><<<%d,256>>>", wl, grid);
    } else if (variant == 5) {
        grid = (ctx->num_envs + 63) / 64;
        fn = tr ? (const void*)wedm_step_split<true> : (const void*)wedm_step_split<false>;
        std::snprintf(out.name, sizeof(out.name), "wedm_step_split<<<%d,256>>>", grid);
    } else if (variant == 6) {
        const int sli = lanes_index(slanes);
        grid = (ctx->num_envs + 256 / slanes - 1) / (256 / slanes);
        fl = ((size_t)ctx->walk4_C[sli] + 1) * 1024;
        out.walk = ctx->walk_dev + 5 + sli;
        // launches of one microsecond without a trace sample, chunks of at most 64 cells: the instantiation without the loop
        const bool one = WEDM_STREAM_REGWALK && single && !tr && ctx->walk4_C[sli] <= 64;
        fn = one ? pick_stream<false, 64, true>(slanes)
           : ctx->walk4_C[sli] <= 64 ? (tr ? pick_stream<true, 64>(slanes) : pick_stream<false, 64>(slanes))
                                     : (tr ? pick_stream<true, 104>(slanes) : pick_stream<false, 104>(slanes));
        std::snprintf(out.name, sizeof(out.name), "wedm_step_stream<%d><<<%d,256,%zuB>>>", slanes, grid, fl);
    } else if (lanes_sv_ok && variant == 11) {  // (by name only: at 16 384 environments x <= 450 segments it measures 2.39e9 against the packed form's 2.48e9 - 2.62e9)
        grid = (ctx->num_envs + 192 / svgl - 1) / (192 / svgl);
        fl = (2 * (size_t)((ctx->n_seg_max + 2 * svgl - 1) / (2 * svgl)) + 2) * 768 + (svgl == 4 ? sizeof(ServedBox<48>) : svgl == 8 ? sizeof(ServedBox<24>) : sizeof(ServedBox<12>));
        fn = svgl == 4 ? (const void*)wedm_step_lanes_served<4> : svgl == 8 ? (const void*)wedm_step_lanes_served<8> : (const void*)wedm_step_lanes_served<16>;
        std::snprintf(out.name, sizeof(out.name), "wedm_step_lanes_served<%d><<<%d,256,%zuB>>>", svgl, grid, fl);
    } else if ((variant == 2 || variant == 11) && use_pk) {
        grid = (ctx->num_envs + 256 / pklanes - 1) / (256 / pklanes);
        fl = (2 * (size_t)((ctx->n_seg_max + 2 * pklanes - 1) / (2 * pklanes)) + 2) * 1024;
        fn = tr ? pick_lanes_pk<true>(pklanes) : pick_lanes_pk<false>(pklanes);
        std::snprintf(out.name, sizeof(out.name), "wedm_step_lanes_pk<%d><<<%d,256,%zuB>>>", pklanes, grid, fl);
    } else if (variant == 2 || variant == 10) {
        grid = (ctx->num_envs + 256 / glanes - 1) / (256 / glanes);
        fl = (size_t)((ctx->n_seg_max + glanes - 1) / glanes) * 1024;
        fn = f64 ? (tr ? pick_lanes<true, true>(glanes) : pick_lanes<false, true>(glanes))
                 : (tr ? pick_lanes<true, false>(glanes) : pick_lanes<false, false>(glanes));
        std::snprintf(out.name, sizeof(out.name), "wedm_step_lanes<%d>%s<<<%d,256,%zuB>>>", glanes, f64 ? "[f64 stencil]" : "", grid, fl);
    } else if (variant == 9) {
        grid = (ctx->num_envs + 192 / svl - 1) / (192 / svl);
        fl = (2 * (size_t)ctx->walk_C[svi] + 2) * 768 + sv_box;
        out.walk = ctx->walk_dev + svi;
        out.block = 256;  // three walker waves + the scalar wave
        const bool extra = ((ctx->walk_n1z >> svi) & 1u) || ((ctx->walk_C[svi] > 8) && (ctx->walk_C[svi] & 7) >= 1 && (ctx->walk_C[svi] & 7) <= 2);
        fn = pick_served(svl, extra);
        std::snprintf(out.name, sizeof(out.name), "wedm_step_served<%d><<<%d,256,%zuB>>>", svl, grid, fl);
    } else if (variant == 4) {
        grid = (ctx->num_envs + 256 / planes - 1) / (256 / planes);
        fl = (2 * (size_t)ctx->walk_C[pli] + 2) * 1024;
        out.walk = ctx->walk_dev + pli;
        // handles with in-launch autoreset expect terminations, and so do handles whose kernels have reported a frozen
        // environment (wedm_ctx::frozen_seen): the instantiation that tolerates frozen lanes
        // tables with a one-change boundary tile or a 1- / 2-cell tail: the instantiation that handles them
        const bool extra = ((ctx->walk_n1z >> pli) & 1u) || ((ctx->walk_C[pli] > 8) && (ctx->walk_C[pli] & 7) >= 1 && (ctx->walk_C[pli] & 7) <= 2);
        fn = frozen_ok ? (tr ? pick_packed<true, true>(planes, extra) : pick_packed<false, true>(planes, extra))
                       : (tr ? pick_packed<true, false>(planes, extra) : pick_packed<false, false>(planes, extra));
        std::snprintf(out.name, sizeof(out.name), "wedm_step_packed<%d>%s<<<%d,256,%zuB>>>", planes, frozen_ok ? "[frozen lanes ok]" : "", grid, fl);
    } else {
        grid = (ctx->num_envs + 256 / lanes - 1) / (256 / lanes);
        fl = ((size_t)ctx->walk_C[li] + 1) * 1024;
        out.walk = ctx->walk_dev + li;
        const bool n1 = (ctx->walk_n1z >> li) & 1u;  // the table has a one-change tile that is a boundary tile in every microsecond
        fn = f64 ? (tr ? pick_fused_f64<true>(lanes) : pick_fused_f64<false>(lanes))
           : frozen_ok ? (tr ? pick_fused<true, true>(lanes, n1) : pick_fused<false, true>(lanes, n1))
                       : (tr ? pick_fused<true, false>(lanes, n1) : pick_fused<false, false>(lanes, n1));
        std::snprintf(out.name, sizeof(out.name), "wedm_step_fused<%d>%s<<<%d,256,%zuB>>>", lanes,
                      f64 ? "[f64 stencil]" : frozen_ok ? "[frozen lanes ok]" : "", grid, fl);
    }
    if (fl) {
        // The attribute belongs to the kernel FUNCTION, not to this handle or plan: two live handles with different wire
        // lengths can resolve to the same instantiation, and a later plan with a smaller image must not lower the limit
        // under an earlier plan that is still cached.  Every function is therefore opened up to the device's limit.
        hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
        if (ea != hipSuccess) return hip_fail(ctx, ea, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    out.fn = fn;
    out.grid = grid;
    if (variant != 9) out.block = 256;
    out.lds = fl;
    out.valid = true;
    return WEDM_OK;
}

static thread_local std::string g_create_error;

extern "C" {

int32_t wedm_abi_version(void) { return WEDM_ABI_VERSION; }
#ifndef WEDM_BUILD_ID
#define WEDM_BUILD_ID "unknown"
#endif
const char* wedm_build_id(void) { return WEDM_BUILD_ID; }
int64_t wedm_sizeof_params(void) { return (int64_t)sizeof(wedm_params); }

const char* wedm_last_error(wedm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
const char* wedm_last_kernel(wedm_ctx* ctx) {
    if (!ctx) return "";
    if (ctx->last_plan) ctx->last_kernel = std::string(ctx->last_plan->name) + " n_sub=" + std::to_string(ctx->last_n_sub);
    return ctx->last_kernel.c_str();
}

int32_t wedm_last_occupancy(wedm_ctx* ctx) {
    if (!ctx || !ctx->last_plan) return WEDM_ERR_BAD_ARG;
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ctx->last_plan->fn, ctx->last_plan->block, ctx->last_plan->lds);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
    return (int32_t)n;
}

int32_t wedm_create(const wedm_params* params, int32_t num_envs, int32_t n_seg_max, wedm_ctx** out) {
    if (!params || !out || num_envs <= 0 || n_seg_max <= 0) {
        g_create_error = "wedm_create: null pointer or non-positive size";
        return WEDM_ERR_BAD_ARG;
    }
    if (!params->per_env_geometry && (params->n_seg < 1 || params->n_seg > n_seg_max)) {
        g_create_error = "wedm_create: params.n_seg outside [1, n_seg_max]";
        return WEDM_ERR_BAD_ARG;
    }
    if (params->servo_interval <= 0 || params->dt_us <= 0 || (params->control_mode != 0 && params->control_mode != 1)) {
        g_create_error = "wedm_create: servo_interval/dt must be positive, control_mode 0 or 1";
        return WEDM_ERR_BAD_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("wedm_create: no HIP device visible (") + hipGetErrorString(e) + ")";
        return WEDM_ERR_NO_DEVICE;
    }
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) {
        g_create_error = std::string("hipGetDevice: ") + hipGetErrorString(e);
        return WEDM_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return WEDM_ERR_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("wedm_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return WEDM_ERR_NO_DEVICE;
    }
    wedm_ctx* ctx = new (std::nothrow) wedm_ctx();
    if (!ctx) return WEDM_ERR_BAD_ARG;
    ctx->p = *params;
    ctx->num_envs = num_envs;
    ctx->n_seg_max = n_seg_max;
    ctx->device = dev;
    ctx->lds_limit = (int)prop.sharedMemPerBlock;
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && optin > ctx->lds_limit)
        ctx->lds_limit = optin;
    // per-mode tables -> one small device buffer (per-lane indexed loads)
    const size_t n = WEDM_MAX_MODE + 1;
    const size_t bytes = 4 * n * sizeof(double) + n * sizeof(int32_t);
    if ((e = hipMalloc(&ctx->tables_dev, bytes)) != hipSuccess) {
        g_create_error = std::string("hipMalloc(tables): ") + hipGetErrorString(e);
        delete ctx;
        return WEDM_ERR_HIP;
    }
    char host[4 * 20 * 8 + 20 * 4];
    std::memcpy(host + 0 * n * 8, params->mode_current, n * 8);
    std::memcpy(host + 1 * n * 8, params->crater_mean, n * 8);
    std::memcpy(host + 2 * n * 8, params->crater_std, n * 8);
    std::memcpy(host + 3 * n * 8, params->crater_depth, n * 8);
    std::memcpy(host + 4 * n * 8, params->crater_valid, n * 4);
    if ((e = hipMemcpy(ctx->tables_dev, host, bytes, hipMemcpyHostToDevice)) != hipSuccess) {
        g_create_error = std::string("hipMemcpy(tables): ") + hipGetErrorString(e);
        (void)hipFree(ctx->tables_dev);
        delete ctx;
        return WEDM_ERR_HIP;
    }
    const double* d = (const double*)ctx->tables_dev;
    ctx->tb.mode_current = d;
    ctx->tb.crater_mean = d + n;
    ctx->tb.crater_std = d + 2 * n;
    ctx->tb.crater_depth = d + 3 * n;
    ctx->tb.crater_valid = (const int32_t*)(d + 4 * n);
    if ((e = hipMalloc((void**)&ctx->params_dev, sizeof(wedm_params))) != hipSuccess ||
        (e = hipMemcpy(ctx->params_dev, params, sizeof(wedm_params), hipMemcpyHostToDevice)) != hipSuccess) {
        g_create_error = std::string("params copy: ") + hipGetErrorString(e);
        if (ctx->params_dev) (void)hipFree(ctx->params_dev);
        (void)hipFree(ctx->tables_dev);
        delete ctx;
        return WEDM_ERR_HIP;
    }
    if (!params->per_env_geometry) {
        std::vector<WalkTable> host_tabs(12);  // [10], [11]: two chunks of exactly 64 cells, four of 32 (register kernel)
        const int Ls[5] = {1, 2, 4, 8, 16};
        for (int i = 0; i < 5; ++i) {
            ctx->walk_ok[i] = build_walk(*params, Ls[i], host_tabs[i]);
            ctx->walk_C[i] = host_tabs[i].C;
            if (ctx->walk_ok[i] && (host_tabs[i].kind_n1_mask & 0x80000000u)) ctx->walk_n1z |= 1u << i;
            ctx->walk4_ok[i] = build_walk(*params, Ls[i], host_tabs[5 + i], 4);
            ctx->walk4_C[i] = host_tabs[5 + i].C;
            if (i == 0) ctx->walk_regs_ok = params->n_seg <= 128 && build_walk(*params, 2, host_tabs[10], 64) && host_tabs[10].C == 64 &&
                                            build_walk(*params, 4, host_tabs[11], 32) && host_tabs[11].C == 32;
        }
        const size_t tab_bytes = host_tabs.size() * sizeof(WalkTable);
        if ((e = hipMalloc((void**)&ctx->walk_dev, tab_bytes)) != hipSuccess ||
            (e = hipMemcpy(ctx->walk_dev, host_tabs.data(), tab_bytes, hipMemcpyHostToDevice)) != hipSuccess) {
            g_create_error = std::string("walk tables: ") + hipGetErrorString(e);
            if (ctx->walk_dev) (void)hipFree(ctx->walk_dev);
            (void)hipFree(ctx->params_dev);
            (void)hipFree(ctx->tables_dev);
            delete ctx;
            return WEDM_ERR_HIP;
        }
    }
    // (optional: without it every handle without autoreset simply keeps the instantiation without the frozen-lane code)
    if (hipHostMalloc((void**)&ctx->frozen_seen, sizeof(int32_t), hipHostMallocMapped) == hipSuccess) {
        *ctx->frozen_seen = 0;
        if (hipHostGetDevicePointer((void**)&ctx->frozen_seen_dev, ctx->frozen_seen, 0) != hipSuccess) {
            (void)hipHostFree(ctx->frozen_seen);
            ctx->frozen_seen = ctx->frozen_seen_dev = nullptr;
        }
    } else {
        (void)hipGetLastError();
        ctx->frozen_seen = nullptr;
    }
    *out = ctx;
    return WEDM_OK;
}

int32_t wedm_destroy(wedm_ctx* ctx) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (ctx->frozen_seen) (void)hipHostFree(ctx->frozen_seen);
    if (ctx->tables_dev) (void)hipFree(ctx->tables_dev);
    if (ctx->walk_dev) (void)hipFree(ctx->walk_dev);
    if (ctx->params_dev) (void)hipFree(ctx->params_dev);
    delete ctx;
    return WEDM_OK;
}

int32_t wedm_bind_state(wedm_ctx* ctx, const wedm_state_ptrs* state) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (!state || !state->f64 || !state->i32 || !state->i8 || !state->T)
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_state: null state block");
    if (state->stride < ctx->num_envs) return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_state: stride < num_envs");
    if (ctx->p.obs_dim > 0 && !state->obs) return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_state: obs_dim > 0 but obs is null");
    ctx->s = *state;
    ctx->bound = true;
    ctx->invalidate_plans();
    return WEDM_OK;
}

int32_t wedm_bind_geometry(wedm_ctx* ctx, const wedm_geom_ptrs* geom) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (!geom || !geom->f64 || !geom->i32) return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_geometry: null geometry block");
    ctx->g = *geom;
    ctx->geom_bound = true;
    ctx->invalidate_plans();
    return WEDM_OK;
}

int32_t wedm_bind_trace(wedm_ctx* ctx, const wedm_trace_desc* desc) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    ctx->trace_on = false;
    ctx->trace_us = ctx->trace_count = 0;
    if (!desc) return WEDM_OK;
    const uint32_t f64_all = (1u << WEDM_F64_COUNT) - 1u, i32_all = (1u << WEDM_I32_COUNT) - 1u,
                   i8_all = (1u << WEDM_I8_COUNT) - 1u;
    if ((desc->f64_mask & ~f64_all) || (desc->i32_mask & ~i32_all) || (desc->i8_mask & ~i8_all))
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_trace: mask names a row that does not exist");
    if (desc->i32_mask & (1u << WEDM_I_TIME_HI))
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_trace: TIME_HI is maintained at the end of a launch only (read it from the state block)");
    if (desc->f64_mask & (1u << WEDM_F_VOLT_SUM))
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_trace: VOLT_SUM is published at control steps only (read it from the state block)");
    if ((desc->f64_mask != 0) != (desc->f64 != nullptr) || (desc->i32_mask != 0) != (desc->i32 != nullptr) ||
        (desc->i8_mask != 0) != (desc->i8 != nullptr))
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_trace: a buffer must be given exactly for the non-empty masks");
    if (!desc->f64_mask && !desc->i32_mask && !desc->i8_mask && !desc->T)
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_trace: nothing selected");
    if (desc->every < 1 || desc->capacity < 1)
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_trace: every and capacity must be >= 1");
    if (desc->env_lo < 0 || desc->env_count < 1 || (int64_t)desc->env_lo + desc->env_count > ctx->num_envs)
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_trace: environment range outside [0, num_envs)");
    ctx->trace = *desc;  // travels by value with every launch: nothing to copy to the device here
    ctx->trace_on = true;
    return WEDM_OK;
}

int64_t wedm_trace_samples(wedm_ctx* ctx) { return ctx ? ctx->trace_count : 0; }

int32_t wedm_bind_rng_replay(wedm_ctx* ctx, const double* table, int64_t n_steps) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (table && n_steps < 1) return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_rng_replay: n_steps must be >= 1");
    ctx->replay = table;
    ctx->replay_steps = table ? n_steps : 0;
    ctx->invalidate_plans();
    return WEDM_OK;
}

int32_t wedm_set_kernel(wedm_ctx* ctx, int32_t variant) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (variant < 0 || variant > 11) return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_set_kernel: variant must be 0..11");
    ctx->variant = variant;
    ctx->invalidate_plans();
    return WEDM_OK;
}

#ifdef WEDM_STAMPS
// diagnostic builds only (-DWEDM_STAMPS, tools/stamps*.py): device buffer receiving the phase
// cycle stamps of every wave.  Not part of the shipped library, not declared in the header.
int32_t wedm_debug_set_stamp_buffer(wedm_ctx* ctx, void* buf) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    ctx->dbg = (unsigned long long*)buf;
    return WEDM_OK;
}
#endif

int32_t wedm_set_lanes(wedm_ctx* ctx, int32_t lanes) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (lanes != 0 && lanes_index(lanes) < 0)
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_set_lanes: lanes must be 0 (auto), 1, 2, 4, 8 or 16");
    ctx->lanes = lanes;
    ctx->invalidate_plans();
    return WEDM_OK;
}

int32_t wedm_reset(wedm_ctx* ctx, const uint8_t* mask, uint64_t seed, int32_t reseed, void* stream) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (!ctx->bound) return fail(ctx, WEDM_ERR_NOT_BOUND, "wedm_reset: call wedm_bind_state first");
    if (int32_t rc = check_device(ctx, "wedm_reset")) return rc;
    const int block = 256;
    const int grid = (ctx->num_envs + block - 1) / block;
    hipLaunchKernelGGL(wedm_reset_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, ctx->p, ctx->s,
                       ctx->num_envs, ctx->n_seg_max, mask, (uint32_t)seed, (uint32_t)(seed >> 32), reseed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(ctx, e, "wedm_reset launch");
    if (!mask && ctx->frozen_seen) *(volatile int32_t*)ctx->frozen_seen = 0;  // every environment reset: none is frozen
    return WEDM_OK;
}

int32_t wedm_step(wedm_ctx* ctx, int32_t n_substeps, const wedm_action_ptrs* action, void* stream) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (!ctx->bound) return fail(ctx, WEDM_ERR_NOT_BOUND, "wedm_step: call wedm_bind_state first");
    if (!action || !action->servo || !action->target_voltage || !action->on_time || !action->off_time ||
        !action->current_mode)
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_step: null action leaf");
    if (n_substeps < 0) return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_step: n_substeps < 0");
    if (ctx->p.per_env_geometry && !ctx->geom_bound)
        return fail(ctx, WEDM_ERR_NOT_BOUND, "wedm_step: per_env_geometry set but wedm_bind_geometry not called");
    if (n_substeps == 0) return WEDM_OK;
    if ((uint64_t)n_substeps * (uint64_t)ctx->p.dt_us >= (1ull << 31))
        return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_step: n_substeps * dt_us must stay below 2^31 us per launch (the clock's high word is carried per launch)");
    if (int32_t rc = check_device(ctx, "wedm_step")) return rc;

    const wedm_params& P = ctx->p;
    KArgs k;
    Hot& h = k.hot;
    h.hard_short_gap = P.hard_short_gap; h.base_critical_density = P.base_critical_density;
    h.gap_coefficient = P.gap_coefficient; h.max_critical_density = P.max_critical_density;
    h.sigmoid_steepness = P.sigmoid_steepness;
    h.ignition_a = P.ignition_a; h.ignition_b = P.ignition_b; h.ignition_c = P.ignition_c; h.ln2 = P.ln2;
    h.default_target_voltage = P.default_target_voltage; h.default_on_time = P.default_on_time;
    h.default_off_time = P.default_off_time; h.spark_voltage_factor = P.spark_voltage_factor;
    h.debris_removal_per_us = P.debris_removal_per_us;
    h.dt_s = P.dt_s; h.damping_coeff = P.damping_coeff; h.stiffness_coeff = P.stiffness_coeff;
    h.omega_n = P.omega_n; h.max_acceleration = P.max_acceleration; h.max_jerk_dt = P.max_jerk_dt;
    h.max_speed = P.max_speed;
    h.spool = (float)P.spool_T; h.tref = (float)P.temp_ref; h.alpha = (float)P.alpha_rho;
    h.tdiel = (float)P.dielectric_temperature;
    h.tcrit = (float)P.critical_temperature; h.tbreak = (float)P.breaking_temperature;
    h.servo_interval = P.servo_interval; h.dt_us = P.dt_us; h.control_mode = P.control_mode;
    h.disable_ignition = P.disable_ignition;
    h.has_random_short = P.random_short_max_probability != 0.0 ? 1 : 0;
    h.per_env_geometry = P.per_env_geometry; h.env_id_offset = P.env_id_offset; h.n_seg = P.n_seg;
    h.done_value = P.keep_stepping_terminated ? 0 : 1;
    k.cold.p = ctx->params_dev;
    k.cold.g = ctx->g;
    k.cold.a = *action;
    k.cold.s = ctx->s;
    k.cold.tb = ctx->tb;
    k.cold.replay = ctx->replay;
    k.cold.replay_steps = ctx->replay_steps;
    k.cold.frozen_seen = ctx->frozen_seen_dev;
    k.num_envs = ctx->num_envs;
    k.n_substeps = n_substeps;
    k.n_seg_max = ctx->n_seg_max;
    k.walk = nullptr;
    k.dbg = ctx->dbg;
    k.trace = ctx->trace;
    k.trace_next = INT32_MAX;
    k.trace_slot = 0;
    if (ctx->trace_on) {
        const int64_t every = ctx->trace.every;
        k.trace_next = (int32_t)(every - ctx->trace_us % every - 1);  // 0-based substep of the next sample
        k.trace_slot = (int32_t)(ctx->trace_count % ctx->trace.capacity);
    }

    const bool tr = ctx->trace_on && k.trace_next < n_substeps;  // a sample falls into this launch
    // frozen-lane tile code: handles with in-launch autoreset, and any handle one of whose launches has found a terminated
    // environment.  The kernels set the host-visible word; the host reads it without synchronising, so the switch comes as
    // late as the host runs ahead of the device: every launch ENQUEUED before the first kernel that sets the word has run
    // still takes the instantiation without the frozen-lane code (whose waves with a frozen lane walk cell by cell: slower,
    // same results).  A reset of every environment clears the word on the host while kernels queued earlier may still
    // set it again, and masked resets never clear it: both only keep the FROZEN_OK instantiation (2 % slower on a batch
    // without frozen environments) longer than needed.  Speed only; no result depends on the word.
    const bool frozen_ok = P.autoreset || (ctx->frozen_seen && *(volatile int32_t*)ctx->frozen_seen != 0);
    LaunchPlan& plan = ctx->plans[n_substeps <= 1 ? 1 : 0][tr ? 1 : 0][frozen_ok ? 1 : 0];
    if (!plan.valid) {
        if (int32_t rc = plan_launch(ctx, n_substeps <= 1, tr, frozen_ok, plan)) return rc;
    }
    k.walk = plan.walk;
    void* kargs[] = {(void*)&k};
    hipError_t el = hipLaunchKernel(plan.fn, dim3(plan.grid), dim3(plan.block), kargs, plan.lds, (hipStream_t)stream);
    if (el != hipSuccess) return hip_fail(ctx, el, "wedm_step launch");
    ctx->last_plan = &plan;
    ctx->last_n_sub = n_substeps;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(ctx, e, "wedm_step launch");
    if (ctx->trace_on) {
        const int64_t every = ctx->trace.every;
        ctx->trace_count += (ctx->trace_us % every + n_substeps) / every;
        ctx->trace_us += n_substeps;
    }
    return WEDM_OK;
}

int32_t wedm_debug_math(int32_t kind, const double* a, const double* b, double* out, int32_t n, void* stream) {
    if (!a || !out || n <= 0 || kind < 0 || kind > 8) return WEDM_ERR_BAD_ARG;
    hipLaunchKernelGGL(wedm_debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, kind, a, b,
                       out, n);
    return hipGetLastError() == hipSuccess ? WEDM_OK : WEDM_ERR_HIP;
}

int32_t wedm_debug_poison_lds(float value, void* stream) {
    int dev = 0, lds = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return WEDM_ERR_HIP;
    (void)hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (lds <= 0 || cus <= 0) return WEDM_ERR_HIP;
    if (hipFuncSetAttribute((const void*)wedm_debug_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return WEDM_ERR_HIP;
    // one block per CU holds the whole LDS; a few rounds so that every CU gets one whatever the dispatch order
    hipLaunchKernelGGL(wedm_debug_poison_lds_kernel, dim3(4 * cus), dim3(256), (size_t)lds, (hipStream_t)stream, value, lds / 4);
    return hipGetLastError() == hipSuccess ? WEDM_OK : WEDM_ERR_HIP;
}

}  // extern "C"

#endif  // WEDM_PART
