// wedm_k_global_split.h — wedm_step_global (one lane per environment, the wire walked in place in global memory) and
// wedm_step_split (single microseconds: the wire cut over the four waves of a block, in place in global memory).
//
// Included by wedm_kernels.hip (one translation unit per WEDM_PART; see the bottom of that file).
#pragma once

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


