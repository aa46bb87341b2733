// wedm_served.h — "served" kernels: the float64 scalar physics of a block's environments on a wave of its own.
//
// Included by wedm_kernels.hip (it uses that file's KArgs, WalkTable, copy_wire, tile8_staged, interior_cell ...).
//
// Why.  In every kernel with L >= 4 lanes per environment the scalar physics of a microsecond (wire_edm.py:116-157 without
// the stencil: ~340 wave-level instructions, most of them float64) is executed by every lane of the environment -- once per
// WAVE, i.e. per 64 / L environments: 43 of the 103 wave-instructions per env-step of wedm_step_packed<8> at 32 768 x 400, 85 of
// the 214 of wedm_step_lanes<8> (profiles/valu.json, round 3).  The reference runs that chain once per environment-step.  Here
// a block's last wave is the SCALAR wave: lane i of it owns environment i of the block (its whole Env in registers) and
// runs prelude and epilogue once per environment and microsecond; the other WW (three) waves are WALKER waves: they own the
// wire (in LDS, as in wedm_step_packed) and nothing else -- no Env, no float64 -- and get the six stencil coefficients of a
// microsecond through a mailbox in LDS.  Blocks of FOUR waves (3 walkers + 1 scalar, 24 environments at 8 lanes each), three
// of them per CU at a 168-register budget: a block of 4 + 1 waves is admitted only ONCE per CU at that budget although the
// occupancy API answers 2 (census by HW_ID, tools/stamps_served.py: the dispatcher wants room for ceil(5 / 4) = 2 waves on
// EVERY SIMD per block), and one such block per CU is as fast as wedm_step_packed and no faster.
//
// Protocol (LDS operations of one wave are processed in order; all waves of a block are resident together):
//   scalar wave, step k : prelude(k) -> cf[k & 1][env] (jf, q, plasma cell, flags, convection pair) -> cf_seq = k + 1
//   walker wave, step k : spin until cf_seq > k -> walk(k) -> tmax[k & 1][env] -> tm_seq[wave] = k + 1
//   scalar wave         : reads tmax(k) when tm_seq[every walker wave] > k
// The scalar wave runs ONE STEP AHEAD: the only thing the epilogue of step k needs from the walk is max(T) (wire.py:376-388:
// time above the critical temperature; the break test, after which the reference returns before mechanics and clocks).  Where
// the scalar wave can PROVE from max(T) of step k - 1 and the coefficients of step k that step k cannot break the wire, it runs
// the rest of the epilogue (mechanics, clocks, termination) and the prelude of step k + 1 while the walkers are still in step
// k, and applies the temperature monitor of step k when its maximum arrives.  The proof is a maximum principle of the explicit
// scheme: with every coefficient non-negative and tuf (2 k + conv + adv) <= 1,
//       max(T_new) <= M + tuf (jf (1 + alpha (M - T_ref)) + max(q, 0)),   M = max(max(T_old), T_dielectric)
// (each new cell is a convex combination of old cells and T_dielectric plus the two source terms; float32 rounding of the ~10
// operations of a cell is below 1e-2 K, the test keeps 1 K).  Where the bound does not stay below the breaking temperature,
// at control steps (the observation carries max(T)) and in the first step of a launch (max(T) of a wire the caller may have
// assigned is unknown) the scalar wave simply waits for the walkers, as an unserved kernel does every step.  Nothing is ever
// rolled back; a break in a step that was proven safe would set the environment's sticky ERROR flag (it cannot happen).
//
// Results are bit-identical to every other kernel: the same prelude / epilogue functions on the same values in the same order
// per environment, the same tile code on the same coefficients.
#pragma once

namespace wedm {

// flags word of a published coefficient set
enum { SV_JOULE = 1, SV_DONE = 2, SV_STOP = 4, SV_ADV = 8 };

template <int EPB>
struct ServedBox {   // lives in LDS behind the wire image
    uint32_t cf_seq;          // steps whose coefficients are published
    uint32_t tm_seq[4];       // per walker wave (at most four): steps whose maxima are published
    uint32_t pad_[3];
    float jf[2][EPB], q[2][EPB], conv_base[2][EPB], conv_zone[2][EPB];
    int32_t pidx[2][EPB], flags[2][EPB];
    float tmax[2][EPB];
    float adv[EPB];
};

// diagnostic build -DWEDM_STAMPS (tools/stamps_served.py): per wave {HW_ID, XCC_ID, start, end (100 MHz clock all XCDs
// share), shader-clock cycles spent spinning, shader-clock cycles in the loop, steps speculated, three phase sums}
#ifdef WEDM_STAMPS
struct SvStamps {
    unsigned long long wait_acc = 0, t0 = 0, t1 = 0, c0 = 0, c1 = 0, spec = 0, pa = 0, pb = 0, pc = 0, q0 = 0, q1 = 0, w0 = 0;
    uint32_t hwid = 0, xcc = 0;
};
#define WEDM_SV_CLOCK(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define WEDM_SV_WAIT_BEGIN() WEDM_SV_CLOCK(svs.w0)
#define WEDM_SV_WAIT_END() do { unsigned long long sv_w1_; WEDM_SV_CLOCK(sv_w1_); svs.wait_acc += sv_w1_ - svs.w0; } while (0)
#define WEDM_SV_PHASE(acc) do { __builtin_amdgcn_sched_barrier(0); WEDM_SV_CLOCK(svs.q1); svs.acc += svs.q1 - svs.q0; svs.q0 = svs.q1; __builtin_amdgcn_sched_barrier(0); } while (0)
#define WEDM_SV_PHASE_START() do { __builtin_amdgcn_sched_barrier(0); WEDM_SV_CLOCK(svs.q0); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WEDM_SV_LOOP_START() WEDM_SV_CLOCK(svs.c0)
#define WEDM_SV_LOOP_END() WEDM_SV_CLOCK(svs.c1)
#define WEDM_SV_COUNT_SPEC() (++svs.spec)
__device__ __forceinline__ void sv_stamps_begin(SvStamps& svs) {
    asm volatile("s_getreg_b32 %0, hwreg(4)\n\ts_getreg_b32 %1, hwreg(20)" : "=s"(svs.hwid), "=s"(svs.xcc));
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(svs.t0)::"memory");
}
// `row`: this wave's 12 words of the stamp buffer (lane 0 writes), or null
__device__ __forceinline__ void sv_stamps_out(SvStamps& svs, unsigned long long* row) {
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(svs.t1)::"memory");
    if (row && (threadIdx.x & 63) == 0) {
        row[0] = svs.hwid; row[1] = svs.xcc; row[2] = svs.t0; row[3] = svs.t1; row[4] = svs.wait_acc; row[5] = svs.c1 - svs.c0;
        row[6] = svs.spec; row[7] = svs.pa; row[8] = svs.pb; row[9] = svs.pc;
    }
}
#else
struct SvStamps { };
#define WEDM_SV_WAIT_BEGIN() do { } while (0)
#define WEDM_SV_WAIT_END() do { } while (0)
#define WEDM_SV_PHASE(acc) do { } while (0)
#define WEDM_SV_PHASE_START() do { } while (0)
#define WEDM_SV_LOOP_START() do { } while (0)
#define WEDM_SV_LOOP_END() do { } while (0)
#define WEDM_SV_COUNT_SPEC() do { } while (0)
__device__ __forceinline__ void sv_stamps_begin(SvStamps&) { }
__device__ __forceinline__ void sv_stamps_out(SvStamps&, unsigned long long*) { }
#endif

__device__ __forceinline__ void sv_wait(const volatile uint32_t* p, uint32_t want) {
    // wave-uniform spin on an LDS word another wave of the block advances (monotone counters: signed distance)
    for (;;) {
        const uint32_t v = __builtin_amdgcn_readfirstlane(*p);
        if ((int32_t)(v - want) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// the same with one word per lane (the scalar wave: every lane waits for the walker wave of ITS environment)
__device__ __forceinline__ void sv_wait_lanes(const volatile uint32_t* p, uint32_t want) {
    for (;;) {
        const uint32_t v = *p;
        if (__all((int32_t)(v - want) >= 0)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

}  // namespace wedm

// -DWEDM_SV_CALL_GENERAL (tried, slower, kept as a switch): the GENERAL prelude (a lane ignites, a short, a control-step
// latch: a few per cent of a batch's microseconds) as a real function call.  Inlined, its ~100 temporaries on top of the Env
// cost the scalar wave 46-62 spilled registers at the 168 registers three blocks per CU leave a wave.  Called, what it takes
// by reference lives in the stack frame around the call only -- but every other value that lives across the call is spilled
// around it, in the hot loop too, and the chain of the scalar wave got longer, not shorter.
// (Arguments of a function are vector registers: what is uniform -- the every-step parameters, the pointer block, the
// geometry -- is read again from the kernel-argument segment inside, into scalar registers, as the kernels read it.  The
// segment's address is handed over by the caller: in a function __builtin_amdgcn_kernarg_segment_ptr() is a NULL pointer.)
__device__ __attribute__((noinline)) void sv_general_prelude(uint64_t kernarg_addr, int64_t e, uint32_t gid, Env& s, Persist& ps,
                                                             const QuietTry& qt, Coef& out) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc cannot copy a struct out of the constant address space)
    const uint32_t ka_lo = __builtin_amdgcn_readfirstlane((uint32_t)kernarg_addr), ka_hi = __builtin_amdgcn_readfirstlane((uint32_t)(kernarg_addr >> 32));
    const WEDM_AS4 KArgs* const ka = (const WEDM_AS4 KArgs*)(((uint64_t)ka_hi << 32) | ka_lo);
    const Hot hot = ka->hot;
    const ColdRef cold{(ColdPtr)((const WEDM_AS4 char*)ka + offsetof(KArgs, cold))};
    Geom g;
    load_geom(hot, cold, e, g);
    out = scalar_prelude(hot, cold, g, e, gid, s, ps, true, qt);
#endif
}

// copy_wire() for a block whose first NT threads (NT = L x environments per block) move the wire; LDS rows of NT floats
template <int L, int NT, bool TO_LDS, class Slot>
__device__ __forceinline__ void copy_wire_nt(float* T, int64_t stride, int64_t e0, int num_envs, int n, int tid, float* lds, Slot slot) {
    constexpr int EPB = NT / L;
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

#ifndef WEDM_SERVED_STAGE_W
#define WEDM_SERVED_STAGE_W 2  // pairs per stage of the packed walk
#endif
#ifndef WEDM_SERVED_DENSE
#define WEDM_SERVED_DENSE WEDM_PACKED_DENSE  // the quiet line also carries sparks that keep burning or end
#endif
#ifndef WEDM_SERVED_WAVES_PER_EU
#define WEDM_SERVED_WAVES_PER_EU 3
#endif

// ------------------------------------------------------------------------------------------------ the scalar wave
// Lane `sl` owns environment e0 + sl of the block (EPB of them); the walker waves hold LPE lanes per environment, so the
// maxima of environment sl come from walker wave (sl * LPE) / 64.  Runs from the launch's first barrier to its last store.
template <int EPB, int LPE>
__device__ __forceinline__ void served_scalar_wave(const KArgs& k, const ColdRef cold, volatile ServedBox<EPB>* box, int64_t e0, int sl,
                                                   SvStamps& svs, unsigned long long* stamp_row) {
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
#ifdef WEDM_SV_NO_SCALAR  // (register-pressure probes of the two roles: tools/kernel_resources.py -DWEDM_SV_NO_...)
    return;
#endif
#ifndef WEDM_SV_NO_PRIO
    __builtin_amdgcn_s_setprio(3);  // its chain is on the critical path of four walker waves
#endif
    Hot hv = k.hot;
#ifdef WEDM_SV_PIN  // (no pinned constants: the scalar wave's 168 registers hold an Env and a prelude's temporaries, nothing to spare)
    pin_mechanics_in_vgprs(hv);
    pin_quiet_in_vgprs(hv);
#endif
    const int64_t e = e0 + (sl < EPB ? sl : 0);
    const bool live = sl < EPB && e0 + sl < k.num_envs;
    Env s;
    Geom g;
    Persist ps{0.0f, 0.0f, 0.0f, 0};
    load_geom(k.hot, cold, live ? e : 0, g);
    if (live) load_env(cold, e, s);
    else { s.done = WEDM_DEAD_LANE; s.unwind = 0.0; s.h_base = 0.0f; s.h_zone = 0.0f; s.broken = 0; s.ctrl = 0; s.tmax = spool; s.tcrit = 0; }
    const bool reinit = live && s.done && WEDM_AUTORESET(cold);  // next-step autoreset
    if (reinit) reinit_env(cold, e, s, true);
    const bool frozen0 = s.done;
    if (!s.done) {
        s.ipk = peak_current(cold, s.mode, e);
        init_persist(k.hot, cold, e, s, ps);
    }
    const uint32_t gid = k.hot.env_id_offset + (uint32_t)e;
    if (sl < EPB) box->adv[sl] = ps.adv;
    __syncthreads();  // (A) the walkers have staged the wire; the mailbox is initialised

    // the maximum principle needs non-negative coefficients and the explicit scheme inside its stability limit
    const float kf = g.k, tuf = g.tuf;
    bool pending = false;      // the previous step's temperature monitor is still to be applied (wave-uniform)
    bool pend_live = false;    // ... for this lane
    bool have_m = false;       // max(T) of the wire as it is now is known (wave-uniform): not in a launch's first step
    float M = spool;
    int it = 0;
    WEDM_SV_LOOP_START();
    for (; it < k.n_substeps; ++it) {
        const int slot = it & 1;
        if (__all(s.done != 0)) {  // every environment of the block is terminated: the walkers stop too
            if (sl < EPB) box->flags[slot][sl] = SV_STOP | SV_DONE;
            asm volatile("" ::: "memory");
            if (sl == 0) box->cf_seq = (uint32_t)it + 1u;
            break;
        }
        WEDM_SV_PHASE_START();
        Coef cf{0.0f, 0.0f, 0, -1};
        QuietTry qt;
#ifdef WEDM_SV_STUB_SCALAR  // (instruction-count probe: the scalar wave publishes a quiet step and computes nothing; results are garbage)
        const bool was_quiet = true;
        qt.have_w = false;
#else
        const bool was_quiet = quiet_prelude_t<WEDM_SERVED_DENSE>(hv, cold, g, e, gid, s, qt, cf);
#endif
#ifndef WEDM_SV_NO_GENERAL
#ifndef WEDM_SV_CALL_GENERAL
        if (__builtin_expect(!was_quiet && !s.done, 0)) cf = scalar_prelude(hv, cold, g, e, gid, s, ps, true, qt);
#else
        // (measured: 32 768 x 400 9.8 ms with the call against 8.3 ms inlined -- the values that live across the call are
        // spilled around it in the hot loop too; kept as a switch)
        if (__builtin_expect(!was_quiet && !s.done, 0)) {
            // through COPIES: an object whose address goes to a call lives in memory for its whole life -- handing `s` itself
            // over put every access of the hot loop into scratch (no "spill" in the statistics, 10 000 cycles per step)
            Env s_mem = s;
            Persist ps_mem = ps;
            QuietTry qt_mem = qt;
            Coef cf_mem{0.0f, 0.0f, 0, -1};
            sv_general_prelude((uint64_t)(const WEDM_AS4 char*)__builtin_amdgcn_kernarg_segment_ptr(), e, gid, s_mem, ps_mem, qt_mem, cf_mem);
            s = s_mem;
            ps = ps_mem;
            cf = cf_mem;
        }
#endif
#endif
        const float jf_eff = (cf.joule_on && !s.done) ? cf.jf : 0.0f;
        if (sl < EPB) {
            box->jf[slot][sl] = cf.jf; box->q[slot][sl] = cf.q; box->pidx[slot][sl] = cf.pidx;
            box->conv_base[slot][sl] = ps.conv_base; box->conv_zone[slot][sl] = ps.conv_zone;
            box->flags[slot][sl] = (cf.joule_on ? SV_JOULE : 0) | (s.done ? SV_DONE : 0) | (ps.adv_on ? SV_ADV : 0);
        }
        asm volatile("" ::: "memory");
        if (sl == 0) box->cf_seq = (uint32_t)it + 1u;
        WEDM_SV_PHASE(pa);  // prelude and publication
        // ---- the previous step's temperature monitor, now that its maximum is there (the walkers had a prelude's time)
        if (pending) {
            { WEDM_SV_WAIT_BEGIN(); sv_wait_lanes(&box->tm_seq[(sl < EPB ? sl * LPE : 0) >> 6], (uint32_t)it); WEDM_SV_WAIT_END(); }
            const float tm = sl < EPB ? box->tmax[slot ^ 1][sl] : spool;
            if (pend_live) {
                s.tcrit = tm > hv.tcrit ? s.tcrit + 1 : 0;
                s.tmax = tm;
                if (tm > hv.tbreak) { s.err = 1; s.broken = 1; s.done = hv.done_value; }  // (proven impossible; loud if it ever is not)
                M = tm;
            }
            pending = false;
            have_m = true;
        }
        WEDM_SV_PHASE(pb);  // wait for and apply the previous step's monitor
        // ---- can this step break the wire?
        const float Mb = fmaxf(M, tdiel);
        const float conv_max = fmaxf(ps.conv_base, ps.conv_zone);
        const bool scheme_ok = kf >= 0.0f && tuf > 0.0f && alpha >= 0.0f && ps.conv_base >= 0.0f && ps.conv_zone >= 0.0f &&
                               ps.adv >= 0.0f && jf_eff >= 0.0f && tuf * (2.0f * kf + conv_max + ps.adv) <= 1.0f;
        const float rise = tuf * (jf_eff * (1.0f + alpha * (Mb - tref)) + fmaxf(cf.q, 0.0f));
        const bool safe = s.done || (scheme_ok && !s.ctrl && Mb + rise + 1.0f < hv.tbreak);  // (a NaN anywhere: not safe)
#ifdef WEDM_SV_NO_SPEC  // (ablation: never ahead -- every step waits for its maximum, as an unserved kernel does)
        if (false) {
#else
        if (have_m && __all(safe)) {
#endif
            // proven: no lane's wire breaks in this step -> the rest of the epilogue now, the monitor when the maximum arrives
            pend_live = !s.done;
            WEDM_SV_COUNT_SPEC();
#ifndef WEDM_SV_STUB_SCALAR
            if (!s.done) {
                epilogue_voltage_sum(s);
                epilogue_motion(hv, s);
            }
#endif
            pending = true;
        } else {
            { WEDM_SV_WAIT_BEGIN(); sv_wait_lanes(&box->tm_seq[(sl < EPB ? sl * LPE : 0) >> 6], (uint32_t)it + 1u); WEDM_SV_WAIT_END(); }
            const float tm = sl < EPB ? box->tmax[slot][sl] : spool;
            if (!s.done) {
                scalar_epilogue(hv, s, tm);
                if (s.ctrl) control_step_outputs(cold, e, s, true);
                M = tm;
            }
            have_m = true;
        }
        WEDM_SV_PHASE(pc);  // the proof and the rest of the epilogue
    }
    if (pending) {  // the last step's monitor
        sv_wait_lanes(&box->tm_seq[(sl < EPB ? sl * LPE : 0) >> 6], (uint32_t)it);
        const float tm = sl < EPB ? box->tmax[(it - 1) & 1][sl] : spool;
        if (pend_live) {
            s.tcrit = tm > hv.tcrit ? s.tcrit + 1 : 0;
            s.tmax = tm;
            if (tm > hv.tbreak) { s.err = 1; s.broken = 1; s.done = hv.done_value; }
        }
    }
    WEDM_SV_LOOP_END();
    sv_stamps_out(svs, stamp_row);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();  // (B) the walkers' last step is in LDS
    if (live) {
        if (WEDM_REWARD_ON(cold)) {
            if (!frozen0) write_reward(cold, e, s);
            else cold->s.reward[e] = 0.0f;  // a frozen environment earns nothing (not the previous launch's reward)
        }
        store_time_hi(cold, e, s, (uint32_t)k.n_substeps * (uint32_t)k.hot.dt_us);
        store_env(cold, e, s);
    }
}

// ============================================ served packed kernel: L lanes / env, 2 cells / op, scalar physics on a wave of its own
// Walk, LDS image and tile table are wedm_step_packed's (two virtual chunks per lane in float2 registers; the table built for
// 2 L chunks; one-change tiles and 1- / 2-cell tails with EXTRA).  A wave with a frozen (terminated) environment keeps the
// tile code, its lanes do not store (wedm_step_packed's FROZEN_OK, always on here).
// Not here (the launch plan keeps such launches on wedm_step_packed): a trace sample inside the launch, keep_stepping_terminated.
template <int L, bool EXTRA, int WW = 3>
__global__ void __launch_bounds__((WW + 1) * 64, WEDM_SERVED_WAVES_PER_EU) wedm_step_served(const KArgs k) {
    constexpr int NT = WW * 64;   // walker threads = columns of the LDS image
    constexpr int EPB = NT / L;   // environments per block
    static_assert(EPB <= 64 && WW <= 4, "one lane of the scalar wave per environment of the block");
    typedef ServedBox<EPB> Box;
    const ColdRef cold = kernarg_cold();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const WalkTable* __restrict__ wt = k.walk;  // built for 2L virtual chunks
    const int Cv = wt->C;
    const int R = 2 * Cv;  // data rows per lane; rows R and R+1 are the halo pair
    const int n = k.hot.n_seg;
    const int64_t stride = cold->s.stride;
    volatile Box* const box = (volatile Box*)(lds + (size_t)(R + 2) * NT);
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    const bool scalar_wave = tid >= NT;
    SvStamps svs;
    sv_stamps_begin(svs);
    unsigned long long* const stamp_row = k.dbg ? k.dbg + ((size_t)blockIdx.x * (WW + 1) + (tid >> 6)) * 12 : nullptr;

    if (tid == 0) { box->cf_seq = 0u; box->tm_seq[0] = 0u; box->tm_seq[1] = 0u; box->tm_seq[2] = 0u; box->tm_seq[3] = 0u; }

    if (scalar_wave) {
        served_scalar_wave<EPB, L>(k, cold, box, e0, tid - NT, svs, stamp_row);
        return;
    }

    // ---------------------------------------------------------------------------------------------------- the walker waves
#ifdef WEDM_SV_NO_WALK
    return;
#endif
    const int el = tid / L, c = tid % L;
    const int64_t e = e0 + el;
    const bool live = e < k.num_envs;
    const int wave = tid >> 6;
    // ---- stage: wire cell i -> virtual chunk vc = i / Cv, cell r = i % Cv -> lane vc/2, row 2r + vc%2
    const auto wire_slot = [Cv](int i) { const int vc = i / Cv; return (2 * (i - vc * Cv) + (vc & 1)) * NT + (vc >> 1); };
    copy_wire_nt<L, NT, true>(cold->s.T, stride, e0, k.num_envs, n, tid, lds, wire_slot);
    Geom g;
    load_geom(k.hot, cold, 0, g);  // uniform geometry
    float* col = lds + tid;
    // next-step autoreset: the environment's wire starts at the spool temperature (the scalar wave re-initialises the state)
    const bool reinit = live && WEDM_AUTORESET(cold) && cold->s.i8[(int64_t)WEDM_B_DONE * stride + e] != 0;
    __syncthreads();  // (A)
    if (reinit) {
        for (int row = 0; row < R; ++row) col[row * NT] = spool;
    }
    Persist ps{box->adv[el], 0.0f, 0.0f, 0};

    const int baseA = 2 * c * Cv, baseB = baseA + Cv;  // first wire cell of each virtual chunk
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
    const int t_last = (n - 1 - baseB) >> 3;
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

    WEDM_SV_LOOP_START();
    for (int it = 0; it < k.n_substeps; ++it) {
        const int slot = it & 1;
        { WEDM_SV_WAIT_BEGIN(); sv_wait(&box->cf_seq, (uint32_t)it + 1u); WEDM_SV_WAIT_END(); }
        WEDM_SV_PHASE_START();
        const int32_t fl = box->flags[slot][el];
        if (fl & SV_STOP) break;  // (block-wide: every lane reads it)
        Coef cf{box->jf[slot][el], box->q[slot][el], (fl & SV_JOULE) ? 1 : 0, box->pidx[slot][el]};
        ps.conv_base = box->conv_base[slot][el];
        ps.conv_zone = box->conv_zone[slot][el];
        ps.adv_on = (fl & SV_ADV) ? 1 : 0;
        const bool done = !live || (fl & SV_DONE);

        // ---- halos (OLD values, read before any store of this step)
        const float halo_l = (c > 0) ? col[(R - 1) * NT - 1] : spool;  // left neighbour lane's B[Cv-1]
        const float halo_r = (c < L - 1) ? col[1] : 0.0f;               // right neighbour lane's A[0]
        const float a_last = col[(R - 2) * NT];                        // own A[Cv-1]: left halo of B
        const float b_first = col[NT];                                  // own B[0]: right halo of A
        col[R * NT] = b_first;
        col[(R + 1) * NT] = halo_r;

        const bool frozen_wave = __any(done);
        const bool all_slow = __any(cf.q < 0.0f);  // a negative plasma heat: every cell on the predicated path (same results)
        const uint32_t slow_now = all_slow ? 0xffffffffu : kind_s;
        // regular tiles of THIS microsecond: a contact-flag change inside a tile only matters while current flows
        const uint32_t n_now = (kind_n | kind_ne | (__any(cf.joule_on && !done && cf.jf != 0.0f) ? 0u : kind_nj)) & ~(all_slow ? 0xffffffffu : 0u);

        // full predicated formula for one owned cell, from OLD values (patched cells)
        auto patch_value = [&](int i, int own) -> float {
            const int v = own - 1, r = i - (v ? baseB : baseA), row = 2 * r + v;
            const float left = col[(r > 0 ? row - 2 : row) * NT];
            float tm = r > 0 ? left : (v ? a_last : halo_l);
            if (i == 1) tm = spool;
            const float tp = col[(row + 2) * NT];
            return stencil_cell(i, n, tm, col[row * NT], tp, g, cf, ps, tref, alpha, tdiel);
        };
        const int own_pl = (!done && cf.pidx >= 1) ? owner(cf.pidx) : 0;
        float tpl = 0.0f, tlast = 0.0f;
        if (__any(own_pl != 0)) {
            if (own_pl) tpl = patch_value(cf.pidx, own_pl);
        }
        if (own_last && !done) tlast = patch_value(n - 1, own_last);

        // ---- tail cells: new values from OLD ones, now (not on the predicated path, whose last tile covers them)
        const bool use_tail = EXTRA && tail != 0 && !all_slow;
        // (as ONE packed pair per tail position -- both virtual chunks together, the Joule term only where some lane of the
        // wave carries current, exactly as the tiles do it -- instead of two scalar cells with the Joule term always)
        f2 ttp[2] = {f2{0.0f, 0.0f}, f2{0.0f, 0.0f}};  // [q] = (chunk A, chunk B)
        if (use_tail) {
            const float jfl = (cf.joule_on && !done) ? cf.jf : 0.0f;
            const bool joule_tail = __any(jfl != 0.0f);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q < tail) {
                    const int r = Cv - tail + q;
                    const uint32_t ba = tail_bits >> (4 * (2 * q)), bb = tail_bits >> (4 * (2 * q + 1));
                    const f2 tm = {col[(2 * (r - 1)) * NT], col[(2 * (r - 1) + 1) * NT]};
                    const f2 tcc = {col[(2 * r) * NT], col[(2 * r + 1) * NT]};
                    const f2 tp = {col[(2 * (r + 1)) * NT], col[(2 * (r + 1) + 1) * NT]};
                    const f2 conv = {(ba & 1u) ? ps.conv_zone : ps.conv_base, (bb & 1u) ? ps.conv_zone : ps.conv_base};
                    const f2 jfe = {(ba & 2u) ? jfl : 0.0f, (bb & 2u) ? jfl : 0.0f};
                    ttp[q] = joule_tail ? interior2<true>(tm, tcc, tp, g.k, g.tuf, conv, tdiel, ps.adv, jfe, alpha, tref)
                                        : interior2<false>(tm, tcc, tp, g.k, g.tuf, conv, tdiel, ps.adv, jfe, alpha, tref);
                }
            }
        }
        const int n_walk = use_tail ? n_tiles - 1 : n_tiles;

        float tmax = spool;
        f2 tm1 = {halo_l, a_last};
        f2 tc = {col[0], col[NT]};
        WEDM_SV_PHASE(pa);  // mailbox, halos, patched cells and tails from old values
        {
            const float jf_lane = (cf.joule_on && !done) ? cf.jf : 0.0f;
            const bool joule_wave = __any(jf_lane != 0.0f);
            const float cz = ps.conv_zone, cb = ps.conv_base;

            auto load8 = [&](auto clamp, f2 (&dst)[8], int r0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int p = r0 + 1 + u;
                    if (decltype(clamp)::value) p = p < Cv ? p : Cv;  // pair Cv is the halo pair; later pairs are never used
                    dst[u].x = col[(2 * p) * NT];
                    dst[u].y = col[(2 * p + 1) * NT];
                }
            };
            auto store2 = [&](int r, f2 v) {
                col[(2 * r) * NT] = v.x;
                col[(2 * r + 1) * NT] = v.y;
            };
            // a tile with one coefficient change at `split`: the pairs' coefficients are selected group by group, right before
            // the stage-major group that uses them (all eight up front are 32 registers the walker does not have)
            auto percell_tile = [&](const f2 (&old)[10], f2 (&tn)[8], int split, f2 conv_lo, f2 conv_hi, f2 jfe_lo, f2 jfe_hi) {
                constexpr int W = WEDM_SERVED_STAGE_W;
#pragma unroll
                for (int o = 0; o < 8; o += W) {
                    f2 cv[8], jv[8];  // (only entries o .. o + W - 1 are set and read)
#pragma unroll
                    for (int u = 0; u < W; ++u) {
                        cv[o + u] = (o + u) < split ? conv_lo : conv_hi;
                        jv[o + u] = (o + u) < split ? jfe_lo : jfe_hi;
                    }
                    if (joule_wave) tile_staged<f2, true, true, W>(old, tn, o, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    else tile_staged<f2, false, true, W>(old, tn, o, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                }
            };
            auto tile = [&](auto frozen, int t, f2 (&cur)[8]) {
                constexpr bool FROZEN = decltype(frozen)::value;  // the copy for a wave with frozen lanes: they do not store
                const int r0 = 8 * t;
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
                    if (joule_wave && __any(jfe_lo.x != 0.0f || jfe_lo.y != 0.0f)) {
#pragma unroll
                        for (int o = 0; o < 8; o += WEDM_SERVED_STAGE_W)
                            tile_staged<f2, true, false, WEDM_SERVED_STAGE_W>(old, tn, o, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    } else {
#pragma unroll
                        for (int o = 0; o < 8; o += WEDM_SERVED_STAGE_W)
                            tile_staged<f2, false, false, WEDM_SERVED_STAGE_W>(old, tn, o, g.k, g.tuf, cv, tdiel, ps.adv, jv, alpha, tref);
                    }
                    tn[0].x = (c == 0 && t == 0) ? spool : tn[0].x;
                    const float last_y = (own_last == 2 && t == t_last) ? spool : tn[7].y;
                    float m0 = fmax_gt(tn[0].x, tn[0].y), m1 = fmax_gt(tn[1].x, tn[1].y);
                    if (!FROZEN || !done) {
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
                    const int split = (int)((split_pack[t >> 3] >> ((t & 7) * 4)) & 15u);
                    const f2 conv_hi = {((zhA >> t) & 1u) ? cz : cb, ((zhB >> t) & 1u) ? cz : cb};
                    const f2 jfe_hi = {((jhA >> t) & 1u) ? jf_lane : 0.0f, ((jhB >> t) & 1u) ? jf_lane : 0.0f};
                    f2 old[10], tn[8];
                    old[0] = tm1; old[1] = tc;
#pragma unroll
                    for (int u = 0; u < 8; ++u) old[u + 2] = cur[u];
                    percell_tile(old, tn, split, conv_lo, conv_hi, jfe_lo, jfe_hi);
                    tn[0].x = (c == 0 && t == 0) ? spool : tn[0].x;
                    const float last_y = (own_last == 2 && t == t_last) ? spool : tn[7].y;
                    if (!FROZEN || !done) {
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
                    f2 old[10], tn[8];
                    old[0] = tm1; old[1] = tc;
#pragma unroll
                    for (int u = 0; u < 8; ++u) old[u + 2] = cur[u];
                    percell_tile(old, tn, split, conv_lo, conv_hi, jfe_lo, jfe_hi);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u < cnt) {
                            if (!FROZEN || !done) store2(r0 + u, tn[u]);
                            const bool inA = (n >= 3) && (imA + (uint32_t)u <= span);
                            const bool inB = (n >= 3) && (imB + (uint32_t)u <= span);
                            tmax = inA ? fmax_gt(tmax, tn[u].x) : tmax;
                            tmax = inB ? fmax_gt(tmax, tn[u].y) : tmax;
                        }
                    }
                    tm1 = cur[6];
                    tc = cur[7];
                } else {
                    // TILE_S: per-cell predicated fallback for both components (rare)
#pragma unroll 1
                    for (int u = 0; u < 8; ++u) {
                        const int r = r0 + u;
                        const uint32_t zj = wt->zj[r], iv = wt->iv[r];
                        const f2 tp1 = cur[0];
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            const int vcid = 2 * c + v;
                            const bool zbit = (zj >> vcid) & 1u, jbit = (zj >> (16 + vcid)) & 1u;
                            const bool inter = ((iv >> vcid) & 1u) && !all_slow;
                            const bool valid = ((iv >> (16 + vcid)) & 1u) && !done;
                            const float conv = zbit ? cz : cb, jfe = jbit ? jf_lane : 0.0f;
                            const float m = v ? tm1.y : tm1.x, cc = v ? tc.y : tc.x, pp = v ? tp1.y : tp1.x;
                            float x = interior_cell<true>(m, cc, pp, g.k, g.tuf, conv, tdiel, ps.adv, jfe, alpha, tref);
                            if (!inter && valid) {
                                const int i = (v ? baseB : baseA) + r;
                                x = (i >= 1) ? stencil_cell(i, n, (i == 1) ? spool : m, cc, pp, g, cf, ps, tref, alpha, tdiel) : spool;
                            }
                            if (valid) {
                                col[(2 * r + v) * NT] = x;
                                tmax = fmax_gt(tmax, x);
                            }
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
            if (!frozen_wave) {
                for (int t = 0; t < n_walk; ++t) tile(std::false_type{}, t, bufA);
            } else {
                for (int t = 0; t < n_walk; ++t) tile(std::true_type{}, t, bufA);
            }
        }
        WEDM_SV_PHASE(pb);  // the tiles
        // ---- patches (after every store of the walk): tail cells, then boundary condition, last cell, plasma cell
        if (use_tail && !done) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q < tail) {
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        if ((tail_bits >> (4 * (2 * q + v))) & 4u) {  // interior: exists, counts, and is not the wire's last cell
                            const float x = v ? ttp[q].y : ttp[q].x;
                            col[(2 * (Cv - tail + q) + v) * NT] = x;
                            tmax = fmax_gt(tmax, x);
                        }
                    }
                }
            }
        }
        if (c == 0 && !done) col[0] = spool;
        if (own_last && !done) {
            const int v = own_last - 1;
            col[(2 * (n - 1 - (v ? baseB : baseA)) + v) * NT] = tlast;
            tmax = fmax_gt(tmax, tlast);
        }
        if (own_pl) {
            const int v = own_pl - 1;
            col[(2 * (cf.pidx - (v ? baseB : baseA)) + v) * NT] = tpl;
            tmax = fmax_gt(tmax, tpl);
        }
        tmax = max_over_env_lanes<L>(tmax);
        if (c == 0) box->tmax[slot][el] = tmax;
        asm volatile("" ::: "memory");
        if ((tid & 63) == 0) box->tm_seq[wave] = (uint32_t)it + 1u;
        WEDM_SV_PHASE(pc);  // patches, reduction, publication
    }

    WEDM_SV_LOOP_END();
    sv_stamps_out(svs, stamp_row);
    __syncthreads();  // (B)
    copy_wire_nt<L, NT, false>(cold->s.T, stride, e0, k.num_envs, n, tid, lds, wire_slot);
}

// ============================================ served register kernel: the wire in the walkers' registers, no LDS image
// wedm_step_regs<128, 2>'s walk (wedm_regs_walk.inc) on TWO walker waves of a block -- two lanes per environment, 32 packed
// pairs each --, the scalar physics of the block's 64 environments on a third wave, all 64 of its lanes busy.  Blocks of three
// waves, four to a CU at 168 registers: 256 environments per CU, 65 536 in ONE round.  LDS holds the mailbox only.
// By name only (kernel 12), measured at 65 536 x 128 (round 4): fused launches 1.666e10 env-steps/s against wedm_step_regs<128, 2>'s
// 1.674e10 -- a sixth fewer instructions, given back by spills: the walk with 64 wire registers per lane wants 239 registers
// and spills 53 at the 168 that three waves per SIMD leave, the scalar wave 113 -- and launches of ONE microsecond 41.7 us
// against wedm_step_stream's 20.4 us (every state row loaded and stored, scratch traffic cold at every launch).
template <int CELLS>
__global__ void __launch_bounds__(192, WEDM_SERVED_WAVES_PER_EU) wedm_step_regs_served(const KArgs k) {
    constexpr int L = 2, H = CELLS / (2 * L), EPB = 64, NT = 128;
    static_assert(H % 8 == 0 && H / 8 <= 16, "whole tiles");
    constexpr int SW = WEDM_REGS_SW2;
    typedef ServedBox<EPB> Box;
    const ColdRef cold = kernarg_cold();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    volatile Box* const box = (volatile Box*)lds;
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    SvStamps svs;
    sv_stamps_begin(svs);
    unsigned long long* const stamp_row = k.dbg ? k.dbg + ((size_t)blockIdx.x * 3 + (tid >> 6)) * 12 : nullptr;
    if (tid == 0) { box->cf_seq = 0u; box->tm_seq[0] = 0u; box->tm_seq[1] = 0u; box->tm_seq[2] = 0u; box->tm_seq[3] = 0u; }
    if (tid >= NT) {
        served_scalar_wave<EPB, L>(k, cold, box, e0, tid - NT, svs, stamp_row);
        return;
    }

    // ---------------------------------------------------------------------------------------------------- the walker waves
    const int c = tid % L, el = tid / L, wave = tid >> 6;
    const int64_t e = e0 + el;
    const bool live = e < k.num_envs;
    const WalkTable* __restrict__ wt = k.walk;  // 2 L chunks of H cells
    const int n = k.hot.n_seg;
    const int64_t stride = cold->s.stride;
    const int base = c * 2 * H;  // this lane's first cell
    const float spool = k.hot.spool, tref = k.hot.tref, alpha = k.hot.alpha, tdiel = k.hot.tdiel;
    // the wire first: word q = cells 4 q .. 4 q + 3 of this environment, 16 bytes per lane
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
    Geom g;
    load_geom(k.hot, cold, 0, g);  // uniform geometry
    // next-step autoreset: the environment's wire starts at the spool temperature (the scalar wave re-initialises the state)
    const bool reinit = live && WEDM_AUTORESET(cold) && cold->s.i8[(int64_t)WEDM_B_DONE * stride + e] != 0;
    if (__any(reinit)) {
#pragma unroll
        for (int m = 0; m < H; ++m) P[m] = reinit ? f2{spool, spool} : P[m];
    }
    if (c == 0) P[0].x = spool;  // wire cell 0 is held at the spool temperature (wire.py:83)
    // tile flags of this lane's two chunks (bit t: the tile's first cell lies in the workpiece zone / between the contacts)
    const int n_tiles = wt->n_tiles;
    uint32_t zoneA = 0u, zoneB = 0u, jouleA = 0u, jouleB = 0u, joule_any = 0u;
    for (int t = 0; t < n_tiles; ++t) {
        const uint32_t lo = wt->zj[8 * t];
        zoneA |= ((lo >> (2 * c)) & 1u) << t;       zoneB |= ((lo >> (2 * c + 1)) & 1u) << t;
        jouleA |= ((lo >> (16 + 2 * c)) & 1u) << t; jouleB |= ((lo >> (17 + 2 * c)) & 1u) << t;
        joule_any |= ((lo >> 16) != 0u ? 1u : 0u) << t;
    }
    joule_any = __builtin_amdgcn_readfirstlane(joule_any);
    const uint32_t kind_n = __builtin_amdgcn_readfirstlane(wt->kind_n_mask);
    const uint32_t kind_ne = __builtin_amdgcn_readfirstlane(wt->kind_ne_mask), kind_nj = __builtin_amdgcn_readfirstlane(wt->kind_nj_mask);
    const int last_base = (2 * L - 1) * H;
    const uint32_t last_tile = (n > last_base) ? (1u << ((n - 1 - last_base) >> 3)) : 0u;
    const bool owns_last = c == L - 1;
    __syncthreads();  // (A) the mailbox is initialised, the scalar wave has published the advection coefficients
    Persist ps{box->adv[el], 0.0f, 0.0f, 0};
    f2 convp[H / 8];  // the convection coefficient pair (chunk A, chunk B) of every tile, from the published coefficients

    WEDM_SV_LOOP_START();
    for (int it = 0; it < k.n_substeps; ++it) {
        const int slot = it & 1;
        { WEDM_SV_WAIT_BEGIN(); sv_wait(&box->cf_seq, (uint32_t)it + 1u); WEDM_SV_WAIT_END(); }
        WEDM_SV_PHASE_START();
        const int32_t fl = box->flags[slot][el];
        if (fl & SV_STOP) break;  // (block-wide: every lane reads it)
        Coef cf{box->jf[slot][el], box->q[slot][el], (fl & SV_JOULE) ? 1 : 0, box->pidx[slot][el]};
        ps.conv_base = box->conv_base[slot][el];
        ps.conv_zone = box->conv_zone[slot][el];
        ps.adv_on = (fl & SV_ADV) ? 1 : 0;
        const bool act = live && !(fl & SV_DONE);
#pragma unroll
        for (int t = 0; t < H / 8; ++t)
            convp[t] = f2{((zoneA >> t) & 1u) ? ps.conv_zone : ps.conv_base, ((zoneB >> t) & 1u) ? ps.conv_zone : ps.conv_base};
        float tmax = spool;
        asm volatile("" : "+v"(zoneA), "+v"(zoneB), "+v"(jouleA), "+v"(jouleB));  // (see wedm_step_regs: the rare code's predicates stay where they are used)
        Geom gw = g;
        gw.n_seg = __builtin_amdgcn_readfirstlane(g.n_seg); gw.az_start = __builtin_amdgcn_readfirstlane(g.az_start);
        gw.az_end = __builtin_amdgcn_readfirstlane(g.az_end); gw.cb = __builtin_amdgcn_readfirstlane(g.cb);
        gw.ct = __builtin_amdgcn_readfirstlane(g.ct);
        asm volatile("" : "+s"(gw.n_seg), "+s"(gw.az_start), "+s"(gw.az_end), "+s"(gw.cb), "+s"(gw.ct));
        int nw = __builtin_amdgcn_readfirstlane(n);
        asm volatile("" : "+s"(nw));
        // halos, OLD values (every lane takes part in the exchange, frozen environments included)
        const float a_last = P[H - 1].x, b_first = P[0].y;
        const float give = c == 0 ? P[H - 1].y : P[0].x;
        const float got = __int_as_float(swap_with_neighbour(__float_as_int(give)));
        const float halo_l = c == 0 ? spool : got, halo_r = c == 0 ? got : 0.0f;
        WEDM_SV_PHASE(pa);
        constexpr bool kF64 = false;  // (float32 stencil only)
        const StencilF64 f64c{0.0, 0.0, 0.0};
        const float rw_h_base = 0.0f, rw_h_zone = 0.0f;
#define WEDM_REGS_WALK_TILE_FENCE 1
#include "wedm_regs_walk.inc"
#undef WEDM_REGS_WALK_TILE_FENCE
        WEDM_SV_PHASE(pb);
        tmax = fmax_gt(tmax, __int_as_float(swap_with_neighbour(__float_as_int(tmax))));
        if (c == 0) box->tmax[slot][el] = tmax;
        asm volatile("" ::: "memory");
        if ((tid & 63) == 0) box->tm_seq[wave] = (uint32_t)it + 1u;
        WEDM_SV_PHASE(pc);
    }
    WEDM_SV_LOOP_END();
    sv_stamps_out(svs, stamp_row);

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
    __syncthreads();  // (B)
}
