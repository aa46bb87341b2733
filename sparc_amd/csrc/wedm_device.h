// wedm_device.h — per-lane physics of one Wire-EDM microsecond on gfx950 (CDNA4).
//
// One lane owns one environment's scalar state in registers (float64, exactly the
// reference's Python-float arithmetic) and walks its wire temperature (float32)
// through an accessor (global memory or LDS).  The five reference modules
// (ignition -> material -> dielectric -> wire -> mechanics, wire_edm.py:123-132)
// are fused: everything before the stencil produces six float32 coefficients, the
// stencil makes ONE pass over the wire, everything after consumes max(T).
//
// Bit-exactness rules (compile with -ffp-contract=off -fno-fast-math):
//   * float64 expressions keep CPython's left-to-right order, no FMA contraction;
//   * `x**2` is x*x, `x**3` a correctly rounded cube (two explicit fma's), exp/log
//     are the table-free IEEE-basic-ops versions below — the same expression trees
//     the CPU oracle evaluates in its PORTABLE math mode;
//   * the stencil is float32 op-for-op as the reference evaluates wire.py:58-123
//     under NumPy-2 scalar promotion (every Python float operand cast to float32).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wedm_hip.h"

namespace wedm {

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ double bits2d(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint64_t d2bits(double d) { return (uint64_t)__double_as_longlong(d); }

// exp(x) from IEEE basic operations (ln2 hi/lo reduction + degree-5 minimax)
__device__ __forceinline__ double portable_exp(double x) {
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03;
    const double P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06;
    const double P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.0) return __builtin_huge_val();
    if (x < -708.0) return 0.0;
    double hi = x, lo = 0.0;
    int k = 0;
    double ax = __builtin_fabs(x);
    if (ax > 0.34657359027997264) {
        k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
        hi = x - (double)k * ln2hi;
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (ax < 3.725290298461914e-09) {
        return 1.0 + x;
    }
    double t = x * x;
    double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    return y * bits2d((uint64_t)(k + 1023) << 52);
}

// log(x) for positive normal x (only use: the polar method's s in (0,1))
__device__ __forceinline__ double portable_log(double x) {
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01;
    const double Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01;
    const double Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01;
    const double Lg7 = 1.479819860511658591e-01;
    uint64_t ux = d2bits(x);
    uint32_t hx = (uint32_t)(ux >> 32);
    int k = (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    uint32_t i = (hx + 0x95f64u) & 0x100000u;
    ux = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffu);
    k += (int)(i >> 20);
    double f = bits2d(ux) - 1.0;
    double dk = (double)k;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    return dk * ln2hi - ((hfsq - (s * (hfsq + R) + dk * ln2lo)) - f);
}

// correctly rounded x^3 (double-double product; `(g/g_ref)**3`, dielectric.py:118)
__device__ __forceinline__ double cube_cr(double x) {
    double p = x * x, e = __builtin_fma(x, x, -p);
    double q = p * x, eq = __builtin_fma(p, x, -q);
    return q + (eq + e * x);
}

// CPython float `//` (float_divmod); wire.py:290-294 `int(y_spark // segment_len)`
__device__ __forceinline__ double py_floordiv(double vx, double wx) {
    double mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod != 0.0) {
        if ((wx < 0) != (mod < 0)) {
            mod += wx;
            div -= 1.0;
        }
    }
    double fd;
    if (div != 0.0) {
        fd = floor(div);
        if (div - fd > 0.5) fd += 1.0;
    } else {
        fd = __builtin_copysign(0.0, vx / wx);
    }
    return fd;
}

// int(y // seg) for the spark's cell (wire.py:290-294).  For y >= 0, seg > 0 CPython's float_divmod returns the exact
// floor of the real quotient (fmod is exact, (y - mod) / seg is within half a unit of the integer it then rounds to),
// and that integer is also what one division, a floor and ONE fused remainder give: r = fma(-n, seg, y) is the
// correctly rounded y - n seg, so its sign and its comparison with seg are those of the exact remainder, which
// corrects a floor(y / seg) that the division's rounding put one off.  ~20 instructions instead of ~105 (fmod is a
// software loop); anything else (negative, NaN, huge) takes CPython's own sequence.
__device__ __forceinline__ int spark_cell_offset(double y, double seg) {
    const double q = y / seg;
    const bool fast = y >= 0.0 && seg > 0.0 && q < 1.0e9;
    double n = floor(q);
    const double r = __builtin_fma(-n, seg, y);
    n = r < 0.0 ? n - 1.0 : (r >= seg ? n + 1.0 : n);
    if (!fast) n = py_floordiv(y, seg);
    return (int)n;
}

// ------------------------------------------------------------------------- RNG
// Philox4x32-10, counter {time, episode, global env id, stream}, key = reset seed.
// Counter-based: a variate is a pure function of (seed, env, episode, time, slot),
// so results do not depend on launch geometry, fusion depth or sharding.
//   stream 0   : the four uniforms one microsecond can need (debris-short roll,
//                random-short roll, ignition roll, spark location), u = (w+0.5)*2^-32;
//   stream 1+j : 53-bit pair j of the polar method (crater normal).
struct W4 { uint32_t x, y, z, w; };

__device__ __forceinline__ W4 philox4(uint32_t key0, uint32_t key1, uint32_t time, uint32_t episode,
                                      uint32_t gid, uint32_t stream) {
    uint32_t c0 = time, c1 = episode, c2 = gid, c3 = stream, k0 = key0, k1 = key1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // 64-bit products: one v_mad_u64_u32 each instead of a mul_lo + mul_hi pair
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return W4{c0, c1, c2, c3};
}

__device__ __forceinline__ double u32_to_unit(uint32_t w) { return ((double)w + 0.5) * 2.3283064365386963e-10; }

struct U2 { double a, b; };
__device__ __forceinline__ U2 philox_pair(uint32_t key0, uint32_t key1, uint32_t time, uint32_t episode,
                                           uint32_t gid, uint32_t stream) {
    W4 w = philox4(key0, key1, time, episode, gid, stream);
    U2 u;
    u.a = ((double)(w.x >> 5) * 67108864.0 + (double)(w.y >> 6)) / 9007199254740992.0;
    u.b = ((double)(w.z >> 5) * 67108864.0 + (double)(w.w >> 6)) / 9007199254740992.0;
    return u;
}

// Marsaglia polar method on Philox streams 1, 2, ... (material.py:127's N(0,1))
__device__ __forceinline__ double philox_std_normal(uint32_t key0, uint32_t key1, uint32_t time,
                                                    uint32_t episode, uint32_t gid) {
    for (uint32_t j = 0; j < 64; ++j) {
        U2 u = philox_pair(key0, key1, time, episode, gid, 1u + j);
        double v1 = 2.0 * u.a - 1.0, v2 = 2.0 * u.b - 1.0;
        double s = v1 * v1 + v2 * v2;
        if (s < 1.0 && s != 0.0) return v1 * sqrt(-2.0 * portable_log(s) / s);
    }
    return 0.0;
}

// ------------------------------------------------------------- per-lane state
struct Env {
    double wp, x, v, prev_a;
    double debris, rho, flow, last_gap, last_rho, wire_last_flow;
    double V, I, y, last_crater, cavity;
    double tdelta, tvolt, on, off, tpos, unwind;
    double ipk;  // peak current of the latched mode (ignition.py:98-113); derived, not stored
    double vacc; // running voltage sum since the last control step (row WEDM_F_VOLT_ACC)
    float h_base, h_zone, tmax;
    int32_t time, tss, tsov, tsi, tse, dur, rnd_rem, deb_rem, tcrit, mode, episode, sparks;
    uint32_t key0, key1;
    int32_t state;
    int32_t is_short, broken, reached, done, ctrl, err;
};

// What an ORDINARY microsecond reads.  Passed by value (kernarg -> SGPRs).  Everything
// else stays in the device copy of wedm_params ("cold") and is fetched through an
// opaque pointer inside the rare branch that needs it, so it never occupies SGPRs in
// the substep loop (the full struct is ~1 KB: by value it spilled >150 SGPRs to VGPR
// lanes and cost ~500 v_readlane per step).
struct Hot {
    double hard_short_gap, base_critical_density, gap_coefficient, max_critical_density, sigmoid_steepness;
    double ignition_a, ignition_b, ignition_c, ln2;
    double default_target_voltage, default_on_time, default_off_time, spark_voltage_factor;
    double debris_removal_per_us;
    double dt_s, damping_coeff, stiffness_coeff, omega_n, max_acceleration, max_jerk_dt, max_speed;
    float spool, tref, alpha, tdiel, tcrit, tbreak;
    int32_t servo_interval, dt_us, control_mode, disable_ignition, has_random_short, per_env_geometry;
    uint32_t env_id_offset;
    int32_t n_seg;  // uniform geometry only
    // what a terminating step writes into the lane's `done` (= "frozen") register: 1, or 0 with
    // wedm_params.keep_stepping_terminated (nothing is ever frozen then; store_env derives the DONE row from the flags)
    int32_t done_value;
};

// Pin the float64 every-step constants in VGPRs (same value in every lane).  gfx9 VALU
// instructions read at most ONE scalar operand, so an f64 op on two kernel constants needs a
// v_mov of one of them anyway, and the ~45 SGPRs they occupy are what pushes the substep loop
// into v_readlane / v_writelane SGPR spills.  Only for kernels with VGPRs to spare.
__device__ __forceinline__ void pin_hot_in_vgprs(Hot& h) {
#define WEDM_PIN(x) asm volatile("" : "+v"(h.x))
    WEDM_PIN(hard_short_gap); WEDM_PIN(base_critical_density); WEDM_PIN(gap_coefficient);
    WEDM_PIN(max_critical_density); WEDM_PIN(sigmoid_steepness);
    WEDM_PIN(ignition_a); WEDM_PIN(ignition_b); WEDM_PIN(ignition_c); WEDM_PIN(ln2);
    WEDM_PIN(default_target_voltage); WEDM_PIN(default_on_time); WEDM_PIN(default_off_time);
    WEDM_PIN(spark_voltage_factor); WEDM_PIN(debris_removal_per_us);
    WEDM_PIN(dt_s); WEDM_PIN(damping_coeff); WEDM_PIN(stiffness_coeff); WEDM_PIN(omega_n);
    WEDM_PIN(max_acceleration); WEDM_PIN(max_jerk_dt); WEDM_PIN(max_speed);
#undef WEDM_PIN
}

__device__ __forceinline__ void pin_mechanics_in_vgprs(Hot& h) {  // the epilogue's constants only
#define WEDM_PIN(x) asm volatile("" : "+v"(h.x))
    WEDM_PIN(dt_s); WEDM_PIN(damping_coeff); WEDM_PIN(stiffness_coeff); WEDM_PIN(omega_n);
    WEDM_PIN(max_acceleration); WEDM_PIN(max_jerk_dt); WEDM_PIN(max_speed);
#undef WEDM_PIN
}

__device__ __forceinline__ void pin_quiet_in_vgprs(Hot& h) {  // the quiet prelude's constants
#define WEDM_PIN(x) asm volatile("" : "+v"(h.x))
    WEDM_PIN(base_critical_density); WEDM_PIN(gap_coefficient); WEDM_PIN(max_critical_density);
    WEDM_PIN(sigmoid_steepness); WEDM_PIN(ignition_a); WEDM_PIN(ignition_b); WEDM_PIN(ignition_c); WEDM_PIN(ln2);
#undef WEDM_PIN
}

// per-lane geometry the every-step path needs (uniform values or the env's rows)
struct Geom {
    double cavity_coeff;
    float k, tuf;
    int32_t n_seg, az_start, az_end, cb, ct;
    double k64, tuf64, a64;  // stencil_mode 1 only: k_cond, temp_update_factor, A as float64
};

// per-lane coefficients that survive across substeps (recomputed only when their
// inputs change): wire.py:304-312 advection, wire.py:349-374 convection
struct Persist {
    float adv, conv_base, conv_zone;
    int32_t adv_on;
    double adv64;  // stencil_mode 1 only
};

// float32 per-step coefficients the scalar prelude hands to the stencil pass
struct Coef {
    float jf;  // joule_factor (wire.py:98)
    float q;   // plasma heat (wire.py:298)
    int32_t joule_on, pidx;
    double jf64, q64;  // stencil_mode 1 only: the same two before their rounding to float32
};

struct Tables {  // device copies of wedm_params' per-mode tables
    const double* mode_current;
    const double* crater_mean;
    const double* crater_std;
    const double* crater_depth;
    const int32_t* crater_valid;
};

struct Cold {  // everything reachable only through rare branches
    const wedm_params* p;
    wedm_geom_ptrs g;
    wedm_action_ptrs a;
    wedm_state_ptrs s;
    Tables tb;
    const double* replay;   // wedm_bind_rng_replay: [step][WEDM_REPLAY_SLOTS][stride], or NULL
    int64_t replay_steps;
    // host-visible word (pinned, mapped): a wave of a kernel instantiation WITHOUT the frozen-lane tile code that finds a
    // terminated environment at its start sets it; wedm_step reads it (plain host read, no synchronisation) and launches
    // the FROZEN_OK instantiation from then on.  Only speed depends on it, never a result.
    int32_t* frozen_seen;
};

// The kernels never touch their by-value `Cold` argument directly: they read it THROUGH the
// kernel-argument segment (constant address space -> s_load at the point of use), with the
// segment pointer laundered at every use so that no load is hoisted out of the rare branch it
// sits in.  By value, the ~40 SGPRs of pointers competed with the every-step constants and the
// substep loop carried 200-370 v_readlane / v_writelane SGPR-spill instructions.
#define WEDM_AS4 __attribute__((address_space(4)))
typedef const WEDM_AS4 Cold* ColdPtr;
struct ColdRef {
    ColdPtr base;
    __device__ __forceinline__ ColdPtr get() const {
        ColdPtr p = base;
        asm volatile("" : "+s"(p));
        return p;
    }
    __device__ __forceinline__ ColdPtr operator->() const { return get(); }
};

// Hide a pointer from loop-invariant code motion: loads through the result cannot be
// hoisted out of the (rare) branch they sit in, so they cost SGPRs only there.
typedef const wedm_params* ParamsPtr;
template <class T>
__device__ __forceinline__ const T* opaque(const T* p) {
    asm volatile("" : "+s"(p));
    return p;
}
// The same, but the result points into the CONSTANT address space (wedm_params is written by the host between
// launches only), so a uniform load through it is an `s_load` (scalar cache, its own counter) instead of the
// `flat_load` the laundered generic pointer gets.  For the few reads at kernel entry / exit: a flat load there is
// queued behind the single-microsecond kernel's stream of wire rows and returns only after all of them.  NOT for the
// rare branches of the microsecond loop: there the scalar results cost SGPRs and v_movs into the float64 VALU
// operands, and measured 4-5 % slower on every fused workload than the vector loads.
template <class T>
__device__ __forceinline__ const WEDM_AS4 T* opaque_const(const T* p) {
    asm volatile("" : "+s"(p));
    return (const WEDM_AS4 T*)(unsigned long long)p;
}

// launch-level switches live in the device copy of wedm_params: read where they are used (kernel entry / exit)
#define WEDM_AUTORESET(cold) (opaque((cold)->p)->autoreset != 0)
#define WEDM_REWARD_ON(cold) (opaque((cold)->p)->reward_mode != 0 && (cold)->s.reward != nullptr)
// (the single-microsecond stream kernel's flavour: scalar loads, see opaque_const)
#define WEDM_AUTORESET_SCALAR(cold) (opaque_const((cold)->p)->autoreset != 0)
#define WEDM_REWARD_ON_SCALAR(cold) (opaque_const((cold)->p)->reward_mode != 0 && (cold)->s.reward != nullptr)

#define WEDM_ROW(ptr, row) ((ptr) + (int64_t)(row) * stride + e)

// cold geometry scalars: the env's row when geometry is per environment, else the uniform value
#define WEDM_COLD_GEOM_F64(cold, hot, row, field) \
    ((hot).per_env_geometry ? (cold)->g.f64[(int64_t)(row) * (cold)->s.stride + e] : opaque((cold)->p)->field)
#define WEDM_COLD_GEOM_I32(cold, hot, row, field) \
    ((hot).per_env_geometry ? (cold)->g.i32[(int64_t)(row) * (cold)->s.stride + e] : opaque((cold)->p)->field)

// ignition.py:98-113 with the module's initial cache (ignition.py:79-81): mode None (0, before the
// first latch) hits the fresh cache and yields 60 A whatever `default_current_mode` is; that
// parameter only serves modes outside currents.json.
// After a reset with reset_semantics 1 (the reference's own: the module object lives on) the cache may hold a mode of the
// previous episode: None then misses it and resolves through `default_current_mode`.  That None is encoded as mode -1
// (written by the reset from flag WEDM_B_MODE_CACHED), so no step ever reads the flag.
__device__ __forceinline__ double peak_current(const ColdRef cold, int32_t mode, int64_t e) {
    (void)e;
    if (mode == 0) return 60.0;
    return (mode >= 1 && mode <= WEDM_MAX_MODE) ? cold->tb.mode_current[mode] : opaque(cold->p)->default_current;
}

__device__ __forceinline__ void load_env(const ColdRef cold, int64_t e, Env& v) {
    const ColdPtr c = cold.get();
    const int64_t stride = c->s.stride;
    const struct { const double* f64; const int32_t* i32; const int8_t* i8; } s{c->s.f64, c->s.i32, c->s.i8};
    v.wp = *WEDM_ROW(s.f64, WEDM_F_WORKPIECE_POS); v.x = *WEDM_ROW(s.f64, WEDM_F_WIRE_POS);
    v.v = *WEDM_ROW(s.f64, WEDM_F_WIRE_VEL); v.prev_a = *WEDM_ROW(s.f64, WEDM_F_PREV_ACCEL);
    v.debris = *WEDM_ROW(s.f64, WEDM_F_DEBRIS_VOLUME); v.rho = *WEDM_ROW(s.f64, WEDM_F_DEBRIS_DENSITY);
    v.flow = *WEDM_ROW(s.f64, WEDM_F_FLOW); v.last_gap = *WEDM_ROW(s.f64, WEDM_F_LAST_GAP);
    v.last_rho = *WEDM_ROW(s.f64, WEDM_F_LAST_DENSITY); v.wire_last_flow = *WEDM_ROW(s.f64, WEDM_F_WIRE_LAST_FLOW);
    v.V = *WEDM_ROW(s.f64, WEDM_F_VOLTAGE); v.I = *WEDM_ROW(s.f64, WEDM_F_CURRENT);
    v.y = *WEDM_ROW(s.f64, WEDM_F_SPARK_Y); v.last_crater = *WEDM_ROW(s.f64, WEDM_F_LAST_CRATER);
    v.cavity = *WEDM_ROW(s.f64, WEDM_F_CAVITY); v.tdelta = *WEDM_ROW(s.f64, WEDM_F_TARGET_DELTA);
    v.tvolt = *WEDM_ROW(s.f64, WEDM_F_TARGET_VOLTAGE); v.on = *WEDM_ROW(s.f64, WEDM_F_ON_TIME);
    v.off = *WEDM_ROW(s.f64, WEDM_F_OFF_TIME); v.tpos = *WEDM_ROW(s.f64, WEDM_F_TARGET_POS);
    v.unwind = *WEDM_ROW(s.f64, WEDM_F_UNWIND_VEL); v.vacc = *WEDM_ROW(s.f64, WEDM_F_VOLT_ACC);
    v.h_base = (float)*WEDM_ROW(s.f64, WEDM_F_H_BASE); v.h_zone = (float)*WEDM_ROW(s.f64, WEDM_F_H_ZONE);
    v.tmax = (float)*WEDM_ROW(s.f64, WEDM_F_TMAX);
    v.time = *WEDM_ROW(s.i32, WEDM_I_TIME); v.tss = *WEDM_ROW(s.i32, WEDM_I_SINCE_SERVO);
    v.tsov = *WEDM_ROW(s.i32, WEDM_I_SINCE_OPEN_V); v.tsi = *WEDM_ROW(s.i32, WEDM_I_SINCE_IGNITION);
    v.tse = *WEDM_ROW(s.i32, WEDM_I_SINCE_SPARK_END); v.dur = *WEDM_ROW(s.i32, WEDM_I_SPARK_DUR);
    v.rnd_rem = *WEDM_ROW(s.i32, WEDM_I_RANDOM_SHORT_REM); v.deb_rem = *WEDM_ROW(s.i32, WEDM_I_DEBRIS_SHORT_REM);
    v.tcrit = *WEDM_ROW(s.i32, WEDM_I_TIME_CRITICAL); v.mode = *WEDM_ROW(s.i32, WEDM_I_CURRENT_MODE);
    v.episode = *WEDM_ROW(s.i32, WEDM_I_EPISODE);
    v.key0 = (uint32_t)*WEDM_ROW(s.i32, WEDM_I_KEY_LO); v.key1 = (uint32_t)*WEDM_ROW(s.i32, WEDM_I_KEY_HI);
    v.sparks = *WEDM_ROW(s.i32, WEDM_I_SPARK_COUNT);
    v.state = *WEDM_ROW(s.i8, WEDM_B_SPARK_STATE); v.is_short = *WEDM_ROW(s.i8, WEDM_B_IS_SHORT);
    v.broken = *WEDM_ROW(s.i8, WEDM_B_WIRE_BROKEN); v.reached = *WEDM_ROW(s.i8, WEDM_B_TARGET_REACHED);
    v.done = *WEDM_ROW(s.i8, WEDM_B_DONE); v.ctrl = *WEDM_ROW(s.i8, WEDM_B_CTRL_STEP);
    v.err = *WEDM_ROW(s.i8, WEDM_B_ERROR);
    v.ipk = 0.0;
}

// The high word of `state.time` (row WEDM_I_TIME_HI; the reference counts in unbounded Python ints, wire_edm.py:135).
// The microsecond loop only carries the low word; a launch advances an environment by less than 2^32 us, so the low
// word wrapped inside this launch exactly when it ended below where it started.  Called by the writer lane BEFORE the
// TIME row is stored: the row still holds the launch's starting value (reinit_env stores the reset value there).
// `span` = n_substeps * dt_us: a low word that ended at or above it cannot have wrapped -- the rows are read only
// behind that (almost never taken) test.
__device__ __forceinline__ void store_time_hi(const ColdRef cold, int64_t e, const Env& v, uint32_t span) {
    if ((uint32_t)v.time < span) {
        const ColdPtr c = cold.get();
        const int64_t stride = c->s.stride;
        int32_t* const i32 = c->s.i32;
        const uint32_t t0 = (uint32_t)*WEDM_ROW(i32, WEDM_I_TIME);
        if ((uint32_t)v.time < t0) *WEDM_ROW(i32, WEDM_I_TIME_HI) += 1;
    }
}

// Row WEDM_B_DONE: the lane's frozen flag, or -- with keep_stepping_terminated, where nothing freezes -- `terminated` of
// the last step as the reference's step() returns it: both of its causes are sticky flags (wire_edm.py:129-130,172-179).
__device__ __forceinline__ int32_t done_row(const ColdRef cold, const Env& v) {
    return opaque(cold->p)->keep_stepping_terminated ? (v.broken | v.reached) : v.done;
}

__device__ __forceinline__ void store_env(const ColdRef cold, int64_t e, const Env& v) {
    const ColdPtr c = cold.get();
    const int64_t stride = c->s.stride;
    const struct { double* f64; int32_t* i32; int8_t* i8; } s{c->s.f64, c->s.i32, c->s.i8};
    *WEDM_ROW(s.f64, WEDM_F_WORKPIECE_POS) = v.wp; *WEDM_ROW(s.f64, WEDM_F_WIRE_POS) = v.x;
    *WEDM_ROW(s.f64, WEDM_F_WIRE_VEL) = v.v; *WEDM_ROW(s.f64, WEDM_F_PREV_ACCEL) = v.prev_a;
    *WEDM_ROW(s.f64, WEDM_F_DEBRIS_VOLUME) = v.debris; *WEDM_ROW(s.f64, WEDM_F_DEBRIS_DENSITY) = v.rho;
    *WEDM_ROW(s.f64, WEDM_F_FLOW) = v.flow; *WEDM_ROW(s.f64, WEDM_F_LAST_GAP) = v.last_gap;
    *WEDM_ROW(s.f64, WEDM_F_LAST_DENSITY) = v.last_rho; *WEDM_ROW(s.f64, WEDM_F_WIRE_LAST_FLOW) = v.wire_last_flow;
    *WEDM_ROW(s.f64, WEDM_F_VOLTAGE) = v.V; *WEDM_ROW(s.f64, WEDM_F_CURRENT) = v.I;
    *WEDM_ROW(s.f64, WEDM_F_SPARK_Y) = v.y; *WEDM_ROW(s.f64, WEDM_F_LAST_CRATER) = v.last_crater;
    *WEDM_ROW(s.f64, WEDM_F_CAVITY) = v.cavity; *WEDM_ROW(s.f64, WEDM_F_TARGET_DELTA) = v.tdelta;
    *WEDM_ROW(s.f64, WEDM_F_TARGET_VOLTAGE) = v.tvolt; *WEDM_ROW(s.f64, WEDM_F_ON_TIME) = v.on;
    *WEDM_ROW(s.f64, WEDM_F_OFF_TIME) = v.off; *WEDM_ROW(s.f64, WEDM_F_VOLT_ACC) = v.vacc;
    *WEDM_ROW(s.f64, WEDM_F_H_BASE) = (double)v.h_base; *WEDM_ROW(s.f64, WEDM_F_H_ZONE) = (double)v.h_zone;
    *WEDM_ROW(s.f64, WEDM_F_TMAX) = (double)v.tmax;
    *WEDM_ROW(s.i32, WEDM_I_TIME) = v.time; *WEDM_ROW(s.i32, WEDM_I_SINCE_SERVO) = v.tss;
    *WEDM_ROW(s.i32, WEDM_I_SINCE_OPEN_V) = v.tsov; *WEDM_ROW(s.i32, WEDM_I_SINCE_IGNITION) = v.tsi;
    *WEDM_ROW(s.i32, WEDM_I_SINCE_SPARK_END) = v.tse; *WEDM_ROW(s.i32, WEDM_I_SPARK_DUR) = v.dur;
    *WEDM_ROW(s.i32, WEDM_I_RANDOM_SHORT_REM) = v.rnd_rem; *WEDM_ROW(s.i32, WEDM_I_DEBRIS_SHORT_REM) = v.deb_rem;
    *WEDM_ROW(s.i32, WEDM_I_TIME_CRITICAL) = v.tcrit; *WEDM_ROW(s.i32, WEDM_I_CURRENT_MODE) = v.mode;
    *WEDM_ROW(s.i32, WEDM_I_SPARK_COUNT) = v.sparks;
    *WEDM_ROW(s.i8, WEDM_B_SPARK_STATE) = (int8_t)v.state; *WEDM_ROW(s.i8, WEDM_B_IS_SHORT) = (int8_t)v.is_short;
    *WEDM_ROW(s.i8, WEDM_B_WIRE_BROKEN) = (int8_t)v.broken; *WEDM_ROW(s.i8, WEDM_B_TARGET_REACHED) = (int8_t)v.reached;
    *WEDM_ROW(s.i8, WEDM_B_DONE) = (int8_t)done_row(cold, v); *WEDM_ROW(s.i8, WEDM_B_CTRL_STEP) = (int8_t)v.ctrl;
    *WEDM_ROW(s.i8, WEDM_B_ERROR) = (int8_t)v.err;
}

// ---- single-microsecond launches (wedm_step_split2): the state crosses HBM once per microsecond, so
// only the rows a microsecond READS are loaded and only the rows it can have CHANGED are stored.
// Write-only rows (assigned by every step before any use): last_crater, cavity, tmax, the control-step
// flag; with the ignition module enabled also current and is_short_circuit.
// h64: when given, the two convection coefficients are handed back as loaded (float64) and v.h_base / v.h_zone are left for
// the caller to convert LATER: the conversion is the first use of loaded data, and placed here it made the compiler wait
// for the state rows (a whole memory round trip) before the single-microsecond kernel could request its wire rows.
__device__ __forceinline__ void load_env_inputs(const ColdRef cold, int64_t e, Env& v, bool ignition_on, double* h64 = nullptr,
                                                bool keep_stepping = false) {
    const ColdPtr c = cold.get();
    const int64_t stride = c->s.stride;
    const struct { const double* f64; const int32_t* i32; const int8_t* i8; } s{c->s.f64, c->s.i32, c->s.i8};
    v.wp = *WEDM_ROW(s.f64, WEDM_F_WORKPIECE_POS); v.x = *WEDM_ROW(s.f64, WEDM_F_WIRE_POS);
    v.v = *WEDM_ROW(s.f64, WEDM_F_WIRE_VEL); v.prev_a = *WEDM_ROW(s.f64, WEDM_F_PREV_ACCEL);
    v.debris = *WEDM_ROW(s.f64, WEDM_F_DEBRIS_VOLUME); v.rho = *WEDM_ROW(s.f64, WEDM_F_DEBRIS_DENSITY);
    v.flow = *WEDM_ROW(s.f64, WEDM_F_FLOW); v.last_gap = *WEDM_ROW(s.f64, WEDM_F_LAST_GAP);
    v.last_rho = *WEDM_ROW(s.f64, WEDM_F_LAST_DENSITY); v.wire_last_flow = *WEDM_ROW(s.f64, WEDM_F_WIRE_LAST_FLOW);
    v.V = *WEDM_ROW(s.f64, WEDM_F_VOLTAGE); v.y = *WEDM_ROW(s.f64, WEDM_F_SPARK_Y);
    v.tdelta = *WEDM_ROW(s.f64, WEDM_F_TARGET_DELTA); v.tvolt = *WEDM_ROW(s.f64, WEDM_F_TARGET_VOLTAGE);
    v.on = *WEDM_ROW(s.f64, WEDM_F_ON_TIME); v.off = *WEDM_ROW(s.f64, WEDM_F_OFF_TIME);
    v.tpos = *WEDM_ROW(s.f64, WEDM_F_TARGET_POS); v.unwind = *WEDM_ROW(s.f64, WEDM_F_UNWIND_VEL);
    v.vacc = *WEDM_ROW(s.f64, WEDM_F_VOLT_ACC);
    if (h64) { h64[0] = *WEDM_ROW(s.f64, WEDM_F_H_BASE); h64[1] = *WEDM_ROW(s.f64, WEDM_F_H_ZONE); v.h_base = 0.0f; v.h_zone = 0.0f; }
    else { v.h_base = (float)*WEDM_ROW(s.f64, WEDM_F_H_BASE); v.h_zone = (float)*WEDM_ROW(s.f64, WEDM_F_H_ZONE); }
    v.time = *WEDM_ROW(s.i32, WEDM_I_TIME); v.tss = *WEDM_ROW(s.i32, WEDM_I_SINCE_SERVO);
    v.tsov = *WEDM_ROW(s.i32, WEDM_I_SINCE_OPEN_V); v.tsi = *WEDM_ROW(s.i32, WEDM_I_SINCE_IGNITION);
    v.tse = *WEDM_ROW(s.i32, WEDM_I_SINCE_SPARK_END); v.dur = *WEDM_ROW(s.i32, WEDM_I_SPARK_DUR);
    v.rnd_rem = *WEDM_ROW(s.i32, WEDM_I_RANDOM_SHORT_REM); v.deb_rem = *WEDM_ROW(s.i32, WEDM_I_DEBRIS_SHORT_REM);
    v.tcrit = *WEDM_ROW(s.i32, WEDM_I_TIME_CRITICAL); v.mode = *WEDM_ROW(s.i32, WEDM_I_CURRENT_MODE);
    v.episode = *WEDM_ROW(s.i32, WEDM_I_EPISODE);
    v.key0 = (uint32_t)*WEDM_ROW(s.i32, WEDM_I_KEY_LO); v.key1 = (uint32_t)*WEDM_ROW(s.i32, WEDM_I_KEY_HI);
    v.sparks = *WEDM_ROW(s.i32, WEDM_I_SPARK_COUNT);
    v.state = *WEDM_ROW(s.i8, WEDM_B_SPARK_STATE);
    v.broken = *WEDM_ROW(s.i8, WEDM_B_WIRE_BROKEN); v.reached = *WEDM_ROW(s.i8, WEDM_B_TARGET_REACHED);
    v.done = *WEDM_ROW(s.i8, WEDM_B_DONE); v.err = *WEDM_ROW(s.i8, WEDM_B_ERROR);
    v.I = 0.0; v.is_short = 0;
    if (!ignition_on) {  // the caller forces the spark (single_spark_animation.py): both are inputs then
        v.I = *WEDM_ROW(s.f64, WEDM_F_CURRENT);
        v.is_short = *WEDM_ROW(s.i8, WEDM_B_IS_SHORT);
    }
    v.last_crater = 0.0; v.cavity = 0.0; v.tmax = 0.0f; v.ctrl = 0;
    // keep_stepping_terminated: a broken wire's temperature monitor stands still (wire.py:260-261), so the maximum is an
    // input for such a lane (wave-uniform switch; the default mode never loads the row)
    if (keep_stepping) v.tmax = (float)*WEDM_ROW(s.f64, WEDM_F_TMAX);
    v.ipk = 0.0;
}

// ---- the same inputs with HALF the load instructions (stream kernel, an even number of lanes per environment): the
// vector-memory front end of a CU handles one load instruction in ~16-18 cycles whatever it fetches, all 8 waves of the CU
// queue their requests at the same moment, and ~45 of a wave's ~67 loads are state rows that both lanes of an
// environment fetch redundantly.  Here the even lane of a pair loads row A and the odd lane row B IN ONE INSTRUCTION, and
// the two swap through DPP once the data are there (quad_perm [1,0,3,2]: no LDS, no wait counter).
__device__ __forceinline__ int32_t swap_with_neighbour(int32_t x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false); }
__device__ __forceinline__ double swap_with_neighbour(double x) {
    const uint64_t u = d2bits(x);
    const uint32_t lo = (uint32_t)swap_with_neighbour((int32_t)(uint32_t)u), hi = (uint32_t)swap_with_neighbour((int32_t)(uint32_t)(u >> 32));
    return bits2d(((uint64_t)hi << 32) | lo);
}
struct PairRaw { double f[11]; int32_t i[7]; int32_t b[3]; double cur; int32_t shrt; };  // a lane's own row of every pair, as loaded

#define WEDM_PAIR_F64(X) X(0, WEDM_F_WORKPIECE_POS, WEDM_F_WIRE_POS) X(1, WEDM_F_WIRE_VEL, WEDM_F_PREV_ACCEL) \
    X(2, WEDM_F_DEBRIS_VOLUME, WEDM_F_DEBRIS_DENSITY) X(3, WEDM_F_FLOW, WEDM_F_LAST_GAP) X(4, WEDM_F_LAST_DENSITY, WEDM_F_WIRE_LAST_FLOW) \
    X(5, WEDM_F_VOLTAGE, WEDM_F_SPARK_Y) X(6, WEDM_F_TARGET_DELTA, WEDM_F_TARGET_VOLTAGE) X(7, WEDM_F_ON_TIME, WEDM_F_OFF_TIME) \
    X(8, WEDM_F_TARGET_POS, WEDM_F_UNWIND_VEL) X(9, WEDM_F_VOLT_ACC, WEDM_F_H_BASE) X(10, WEDM_F_H_ZONE, WEDM_F_TMAX)
#define WEDM_PAIR_I32(X) X(0, WEDM_I_TIME, WEDM_I_SINCE_SERVO) X(1, WEDM_I_SINCE_OPEN_V, WEDM_I_SINCE_IGNITION) \
    X(2, WEDM_I_SINCE_SPARK_END, WEDM_I_SPARK_DUR) X(3, WEDM_I_RANDOM_SHORT_REM, WEDM_I_DEBRIS_SHORT_REM) \
    X(4, WEDM_I_TIME_CRITICAL, WEDM_I_CURRENT_MODE) X(5, WEDM_I_EPISODE, WEDM_I_KEY_LO) X(6, WEDM_I_KEY_HI, WEDM_I_SPARK_COUNT)
#define WEDM_PAIR_I8(X) X(0, WEDM_B_SPARK_STATE, WEDM_B_WIRE_BROKEN) X(1, WEDM_B_TARGET_REACHED, WEDM_B_DONE) X(2, WEDM_B_ERROR, WEDM_B_ERROR)

// requests: 21 loads (+ 2 when the caller forces the spark) instead of 40
__device__ __forceinline__ void load_env_inputs_paired_issue(const ColdRef cold, int64_t e, bool odd, bool ignition_on, PairRaw& r) {
    const ColdPtr c = cold.get();
    const int64_t stride = c->s.stride;
    const struct { const double* f64; const int32_t* i32; const int8_t* i8; } s{c->s.f64, c->s.i32, c->s.i8};
#define WEDM_LD(k, A, B) r.f[k] = s.f64[(int64_t)(odd ? (int)(B) : (int)(A)) * stride + e];
    WEDM_PAIR_F64(WEDM_LD)
#undef WEDM_LD
#define WEDM_LD(k, A, B) r.i[k] = s.i32[(int64_t)(odd ? (int)(B) : (int)(A)) * stride + e];
    WEDM_PAIR_I32(WEDM_LD)
#undef WEDM_LD
#define WEDM_LD(k, A, B) r.b[k] = s.i8[(int64_t)(odd ? (int)(B) : (int)(A)) * stride + e];
    WEDM_PAIR_I8(WEDM_LD)
#undef WEDM_LD
    r.cur = 0.0; r.shrt = 0;
    if (!ignition_on) {  // the caller forces the spark (single_spark_animation.py): current and the short flag are inputs then
        r.cur = *WEDM_ROW(s.f64, WEDM_F_CURRENT);
        r.shrt = *WEDM_ROW(s.i8, WEDM_B_IS_SHORT);
    }
}

// the wait for those loads, where the caller wants it (see env_loaded_here)
__device__ __forceinline__ void pair_raw_loaded_here(PairRaw& r) {
#pragma unroll
    for (int k = 0; k < 11; ++k) asm volatile("" : "+v"(r.f[k]));
#pragma unroll
    for (int k = 0; k < 7; ++k) asm volatile("" : "+v"(r.i[k]));
#pragma unroll
    for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(r.b[k]));
    asm volatile("" : "+v"(r.cur), "+v"(r.shrt));
}

// the swap: every lane ends up with both rows of every pair (what load_env_inputs would have loaded)
__device__ __forceinline__ void load_env_inputs_paired_finish(const PairRaw& r, bool odd, Env& v, bool keep_stepping, double* h64) {
    double fa[11], fb[11];
    int32_t ia[7], ib[7], ba[3], bb[3];
#pragma unroll
    for (int k = 0; k < 11; ++k) { const double o = swap_with_neighbour(r.f[k]); fa[k] = odd ? o : r.f[k]; fb[k] = odd ? r.f[k] : o; }
#pragma unroll
    for (int k = 0; k < 7; ++k) { const int32_t o = swap_with_neighbour(r.i[k]); ia[k] = odd ? o : r.i[k]; ib[k] = odd ? r.i[k] : o; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { const int32_t o = swap_with_neighbour(r.b[k]); ba[k] = odd ? o : r.b[k]; bb[k] = odd ? r.b[k] : o; }
    v.wp = fa[0]; v.x = fb[0]; v.v = fa[1]; v.prev_a = fb[1]; v.debris = fa[2]; v.rho = fb[2]; v.flow = fa[3]; v.last_gap = fb[3];
    v.last_rho = fa[4]; v.wire_last_flow = fb[4]; v.V = fa[5]; v.y = fb[5]; v.tdelta = fa[6]; v.tvolt = fb[6]; v.on = fa[7]; v.off = fb[7];
    v.tpos = fa[8]; v.unwind = fb[8]; v.vacc = fa[9];
    h64[0] = fb[9]; h64[1] = fa[10]; v.h_base = 0.0f; v.h_zone = 0.0f;
    v.tmax = keep_stepping ? (float)fb[10] : 0.0f;   // (an input only where a broken wire is stepped on: load_env_inputs)
    v.time = ia[0]; v.tss = ib[0]; v.tsov = ia[1]; v.tsi = ib[1]; v.tse = ia[2]; v.dur = ib[2]; v.rnd_rem = ia[3]; v.deb_rem = ib[3];
    v.tcrit = ia[4]; v.mode = ib[4]; v.episode = ia[5]; v.key0 = (uint32_t)ib[5]; v.key1 = (uint32_t)ia[6]; v.sparks = ib[6];
    v.state = ba[0]; v.broken = bb[0]; v.reached = ba[1]; v.done = bb[1]; v.err = ba[2];
    v.I = r.cur; v.is_short = r.shrt;
    v.last_crater = 0.0; v.cavity = 0.0; v.ctrl = 0;
    v.ipk = 0.0;
}

// Rows that are final once the scalar prelude has run (stored while the wire is being walked).
// `quiet_only`: every step of the launch took quiet_prelude() for the whole wave, which never assigns the
// second group (workpiece position, dielectric / convection caches, latched action, short timers,
// spark count, error flag): those rows are left alone.
__device__ __forceinline__ void store_env_after_prelude(const ColdRef cold, int64_t e, const Env& v, bool quiet_only) {
    const ColdPtr c = cold.get();
    const int64_t stride = c->s.stride;
    const struct { double* f64; int32_t* i32; int8_t* i8; } s{c->s.f64, c->s.i32, c->s.i8};
    *WEDM_ROW(s.f64, WEDM_F_DEBRIS_VOLUME) = v.debris; *WEDM_ROW(s.f64, WEDM_F_DEBRIS_DENSITY) = v.rho;
    *WEDM_ROW(s.f64, WEDM_F_VOLTAGE) = v.V; *WEDM_ROW(s.f64, WEDM_F_CURRENT) = v.I;
    *WEDM_ROW(s.f64, WEDM_F_SPARK_Y) = v.y; *WEDM_ROW(s.f64, WEDM_F_LAST_CRATER) = v.last_crater;
    *WEDM_ROW(s.f64, WEDM_F_CAVITY) = v.cavity;
    *WEDM_ROW(s.i32, WEDM_I_SPARK_DUR) = v.dur;
    *WEDM_ROW(s.i8, WEDM_B_SPARK_STATE) = (int8_t)v.state; *WEDM_ROW(s.i8, WEDM_B_IS_SHORT) = (int8_t)v.is_short;
    *WEDM_ROW(s.i8, WEDM_B_CTRL_STEP) = (int8_t)v.ctrl;
    if (quiet_only) return;
    *WEDM_ROW(s.f64, WEDM_F_WORKPIECE_POS) = v.wp;
    *WEDM_ROW(s.f64, WEDM_F_FLOW) = v.flow; *WEDM_ROW(s.f64, WEDM_F_LAST_GAP) = v.last_gap;
    *WEDM_ROW(s.f64, WEDM_F_LAST_DENSITY) = v.last_rho; *WEDM_ROW(s.f64, WEDM_F_WIRE_LAST_FLOW) = v.wire_last_flow;
    *WEDM_ROW(s.f64, WEDM_F_TARGET_DELTA) = v.tdelta; *WEDM_ROW(s.f64, WEDM_F_TARGET_VOLTAGE) = v.tvolt;
    *WEDM_ROW(s.f64, WEDM_F_ON_TIME) = v.on; *WEDM_ROW(s.f64, WEDM_F_OFF_TIME) = v.off;
    *WEDM_ROW(s.f64, WEDM_F_H_BASE) = (double)v.h_base; *WEDM_ROW(s.f64, WEDM_F_H_ZONE) = (double)v.h_zone;
    *WEDM_ROW(s.i32, WEDM_I_RANDOM_SHORT_REM) = v.rnd_rem; *WEDM_ROW(s.i32, WEDM_I_DEBRIS_SHORT_REM) = v.deb_rem;
    *WEDM_ROW(s.i32, WEDM_I_CURRENT_MODE) = v.mode; *WEDM_ROW(s.i32, WEDM_I_SPARK_COUNT) = v.sparks;
    *WEDM_ROW(s.i8, WEDM_B_ERROR) = (int8_t)v.err;
}

// Rows the scalar epilogue assigns (mechanics, clocks, temperature monitor, termination, voltage sum).
__device__ __forceinline__ void store_env_after_epilogue(const ColdRef cold, int64_t e, const Env& v) {
    const ColdPtr c = cold.get();
    const int64_t stride = c->s.stride;
    const struct { double* f64; int32_t* i32; int8_t* i8; } s{c->s.f64, c->s.i32, c->s.i8};
    *WEDM_ROW(s.f64, WEDM_F_WIRE_POS) = v.x; *WEDM_ROW(s.f64, WEDM_F_WIRE_VEL) = v.v;
    *WEDM_ROW(s.f64, WEDM_F_PREV_ACCEL) = v.prev_a; *WEDM_ROW(s.f64, WEDM_F_TMAX) = (double)v.tmax;
    *WEDM_ROW(s.f64, WEDM_F_VOLT_ACC) = v.vacc;
    *WEDM_ROW(s.i32, WEDM_I_TIME) = v.time; *WEDM_ROW(s.i32, WEDM_I_SINCE_SERVO) = v.tss;
    *WEDM_ROW(s.i32, WEDM_I_SINCE_OPEN_V) = v.tsov; *WEDM_ROW(s.i32, WEDM_I_SINCE_IGNITION) = v.tsi;
    *WEDM_ROW(s.i32, WEDM_I_SINCE_SPARK_END) = v.tse; *WEDM_ROW(s.i32, WEDM_I_TIME_CRITICAL) = v.tcrit;
    *WEDM_ROW(s.i8, WEDM_B_WIRE_BROKEN) = (int8_t)v.broken; *WEDM_ROW(s.i8, WEDM_B_TARGET_REACHED) = (int8_t)v.reached;
    *WEDM_ROW(s.i8, WEDM_B_DONE) = (int8_t)done_row(cold, v);
}

// Make the compiler treat every loaded state register as USED here: it inserts the wait for the state
// loads at this point (and afterwards knows that none of them is pending).  Needed where vector-memory
// operations the compiler cannot count follow (the LDS-DMA loop of wedm_step_stream): a later first use of a
// state register would otherwise be a conservative `s_waitcnt vmcnt(0)` behind them.
__device__ __forceinline__ void env_loaded_here(Env& v) {
    asm volatile("" : "+v"(v.wp), "+v"(v.x), "+v"(v.v), "+v"(v.prev_a), "+v"(v.debris), "+v"(v.rho), "+v"(v.flow),
                      "+v"(v.last_gap), "+v"(v.last_rho), "+v"(v.wire_last_flow), "+v"(v.V), "+v"(v.I), "+v"(v.y),
                      "+v"(v.last_crater), "+v"(v.cavity));
    asm volatile("" : "+v"(v.tdelta), "+v"(v.tvolt), "+v"(v.on), "+v"(v.off), "+v"(v.tpos), "+v"(v.unwind), "+v"(v.vacc),
                      "+v"(v.h_base), "+v"(v.h_zone), "+v"(v.tmax), "+v"(v.time), "+v"(v.tss), "+v"(v.tsov), "+v"(v.tsi),
                      "+v"(v.tse));
    asm volatile("" : "+v"(v.dur), "+v"(v.rnd_rem), "+v"(v.deb_rem), "+v"(v.tcrit), "+v"(v.mode), "+v"(v.episode),
                      "+v"(v.sparks), "+v"(v.key0), "+v"(v.key1), "+v"(v.state), "+v"(v.is_short), "+v"(v.broken),
                      "+v"(v.reached), "+v"(v.done), "+v"(v.ctrl), "+v"(v.err));
}

// ----------------------------------------------------------------- signal trace
// Row r of each state block as the registers hold it (the value store_env would write).
__device__ __forceinline__ double env_f64_row(const Env& v, int row) {
    switch (row) {
        case WEDM_F_WORKPIECE_POS: return v.wp;
        case WEDM_F_WIRE_POS: return v.x;
        case WEDM_F_WIRE_VEL: return v.v;
        case WEDM_F_PREV_ACCEL: return v.prev_a;
        case WEDM_F_DEBRIS_VOLUME: return v.debris;
        case WEDM_F_DEBRIS_DENSITY: return v.rho;
        case WEDM_F_FLOW: return v.flow;
        case WEDM_F_LAST_GAP: return v.last_gap;
        case WEDM_F_LAST_DENSITY: return v.last_rho;
        case WEDM_F_WIRE_LAST_FLOW: return v.wire_last_flow;
        case WEDM_F_VOLTAGE: return v.V;
        case WEDM_F_CURRENT: return v.I;
        case WEDM_F_SPARK_Y: return v.y;
        case WEDM_F_LAST_CRATER: return v.last_crater;
        case WEDM_F_CAVITY: return v.cavity;
        case WEDM_F_TARGET_DELTA: return v.tdelta;
        case WEDM_F_TARGET_VOLTAGE: return v.tvolt;
        case WEDM_F_ON_TIME: return v.on;
        case WEDM_F_OFF_TIME: return v.off;
        case WEDM_F_TARGET_POS: return v.tpos;
        case WEDM_F_UNWIND_VEL: return v.unwind;
        case WEDM_F_H_BASE: return (double)v.h_base;
        case WEDM_F_H_ZONE: return (double)v.h_zone;
        case WEDM_F_TMAX: return (double)v.tmax;
        case WEDM_F_VOLT_ACC: return v.vacc;
    }
    return 0.0;
}
__device__ __forceinline__ int32_t env_i32_row(const Env& v, int row) {
    switch (row) {
        case WEDM_I_TIME: return v.time;
        case WEDM_I_SINCE_SERVO: return v.tss;
        case WEDM_I_SINCE_OPEN_V: return v.tsov;
        case WEDM_I_SINCE_IGNITION: return v.tsi;
        case WEDM_I_SINCE_SPARK_END: return v.tse;
        case WEDM_I_SPARK_DUR: return v.dur;
        case WEDM_I_RANDOM_SHORT_REM: return v.rnd_rem;
        case WEDM_I_DEBRIS_SHORT_REM: return v.deb_rem;
        case WEDM_I_TIME_CRITICAL: return v.tcrit;
        case WEDM_I_CURRENT_MODE: return v.mode;
        case WEDM_I_EPISODE: return v.episode;
        case WEDM_I_KEY_LO: return (int32_t)v.key0;
        case WEDM_I_KEY_HI: return (int32_t)v.key1;
        case WEDM_I_SPARK_COUNT: return v.sparks;
    }
    return 0;
}
// (`keep_stepping`: wedm_params.keep_stepping_terminated -- row DONE is then `terminated` of the step, see done_row())
__device__ __forceinline__ int32_t env_i8_row(const Env& v, int row, bool keep_stepping) {
    switch (row) {
        case WEDM_B_SPARK_STATE: return v.state;
        case WEDM_B_IS_SHORT: return v.is_short;
        case WEDM_B_WIRE_BROKEN: return v.broken;
        case WEDM_B_TARGET_REACHED: return v.reached;
        case WEDM_B_DONE: return keep_stepping ? (v.broken | v.reached) : v.done;
        case WEDM_B_CTRL_STEP: return v.ctrl;
        case WEDM_B_ERROR: return v.err;
    }
    return 0;
}

// Column of environment e in the trace buffers, or -1 when it is not traced.
__device__ __forceinline__ int64_t trace_column(const wedm_trace_desc& tr, int64_t e) {
    const int64_t col = e - tr.env_lo;
    return (col >= 0 && col < tr.env_count) ? col : -1;
}

// The selected scalar rows of one environment into ring slot `slot` (one lane per environment).
__device__ __forceinline__ void trace_scalars(const wedm_trace_desc& tr, int64_t col, const Env& v, int slot, bool keep_stepping) {
    const int64_t cnt = tr.env_count;
    const uint32_t mf = tr.f64_mask, mi = tr.i32_mask, mb = tr.i8_mask;
    // loop over the SET bits (wave-uniform): the cost follows the number of traced rows, not the 45
    // rows that exist (one scalar branch per row tested cost ~13 % with a single traced row)
    if (mf) {
        double* dst = tr.f64 + (int64_t)slot * __builtin_popcount(mf) * cnt + col;
        for (uint32_t m = mf; m; m &= m - 1u) { *dst = env_f64_row(v, __builtin_ctz(m)); dst += cnt; }
    }
    if (mi) {
        int32_t* dst = tr.i32 + (int64_t)slot * __builtin_popcount(mi) * cnt + col;
        for (uint32_t m = mi; m; m &= m - 1u) { *dst = env_i32_row(v, __builtin_ctz(m)); dst += cnt; }
    }
    if (mb) {
        int8_t* dst = tr.i8 + (int64_t)slot * __builtin_popcount(mb) * cnt + col;
        for (uint32_t m = mb; m; m &= m - 1u) { *dst = (int8_t)env_i8_row(v, __builtin_ctz(m), keep_stepping); dst += cnt; }
    }
}

__device__ __forceinline__ void load_geom(const Hot& hot, const ColdRef cold, int64_t e, Geom& g) {
    const ColdPtr c = cold.get();
    const int64_t stride = c->s.stride;
    if (hot.per_env_geometry) {
        const struct { const double* f64; const int32_t* i32; } gp{c->g.f64, c->g.i32};
        g.cavity_coeff = *WEDM_ROW(gp.f64, WEDM_G_CAVITY_COEFF);
        g.k = (float)*WEDM_ROW(gp.f64, WEDM_G_K_COND); g.tuf = (float)*WEDM_ROW(gp.f64, WEDM_G_TUF);
        g.n_seg = *WEDM_ROW(gp.i32, WEDM_GI_N_SEG);
        g.az_start = *WEDM_ROW(gp.i32, WEDM_GI_AZ_START); g.az_end = *WEDM_ROW(gp.i32, WEDM_GI_AZ_END);
        g.cb = *WEDM_ROW(gp.i32, WEDM_GI_CONTACT_BOTTOM); g.ct = *WEDM_ROW(gp.i32, WEDM_GI_CONTACT_TOP);
        g.k64 = *WEDM_ROW(gp.f64, WEDM_G_K_COND); g.tuf64 = *WEDM_ROW(gp.f64, WEDM_G_TUF);
        g.a64 = *WEDM_ROW(gp.f64, WEDM_G_A_SURF);
    } else {
        const wedm_params* p = c->p;
        g.cavity_coeff = p->cavity_coeff;
        g.k = (float)p->k_cond; g.tuf = (float)p->tuf;
        g.n_seg = p->n_seg; g.az_start = p->az_start; g.az_end = p->az_end;
        g.cb = p->contact_bottom; g.ct = p->contact_top;
        g.k64 = p->k_cond; g.tuf64 = p->tuf; g.a64 = p->a_surf;
    }
}

// The cold parameter block as scalar_prelude reads it: through the laundered GENERIC pointer (vector `flat_load`s: what
// the fused kernels measured fastest, see opaque_const) or, SCOLD, through the constant address space (scalar loads).
// The single-microsecond stream kernel needs the second: vector loads return in order, and its general prelude runs
// while the lane's whole wire is still in flight -- a flat load issued there comes back only after every wire word.
template <bool SCOLD> struct ColdParams;
template <> struct ColdParams<false> {
    typedef const wedm_params* type;
    static __device__ __forceinline__ type get(const wedm_params* p) { return opaque(p); }
};
template <> struct ColdParams<true> {
    typedef const WEDM_AS4 wedm_params* type;
    static __device__ __forceinline__ type get(const wedm_params* p) { return opaque_const(p); }
};

// Entry `lane` of the per-mode crater tables in lane `lane` (the stream kernel requests them with its very first loads,
// like the peak-current table): a lookup is then a cross-lane read instead of a vector load queued behind the wire.
struct LaneTables { double mean, sd, depth; int32_t valid; };

// wire.py:349-374: h_eff * A products the stencil uses (float32 x float32, wire.py:109)
template <bool SCOLD = false>
__device__ __forceinline__ void refresh_convection(const Hot& hot, const ColdRef cold, int64_t e, const Env& s,
                                                   Persist& ps) {
    const float A = (float)(hot.per_env_geometry ? cold->g.f64[(int64_t)WEDM_G_A_SURF * cold->s.stride + e]
                                                 : ColdParams<SCOLD>::get(cold->p)->a_surf);
    ps.conv_base = s.h_base * A;
    ps.conv_zone = s.h_zone * A;
}

// once per launch: coefficients whose inputs no module changes (wire.py:304-312)
// SCALAR_LOADS (the single-microsecond stream kernel): the uniform constants through the constant address space
template <bool SCALAR_LOADS = false>
__device__ __forceinline__ void init_persist(const Hot& hot, const ColdRef cold, int64_t e, const Env& s, Persist& ps) {
    double adv = 0.0;
    if (__builtin_fabs(s.unwind) > 1e-6) {
        if (SCALAR_LOADS) {
            const auto pc = opaque_const(cold->p);
            const double s_area = hot.per_env_geometry ? cold->g.f64[(int64_t)WEDM_G_S_AREA * cold->s.stride + e] : pc->s_area;
            adv = pc->rho_c * __builtin_fabs(s.unwind) * s_area;
        } else {
            const double s_area = WEDM_COLD_GEOM_F64(cold, hot, WEDM_G_S_AREA, s_area);
            adv = cold->p->rho_c * __builtin_fabs(s.unwind) * s_area;
        }
    }
    ps.adv_on = __builtin_fabs(adv) > 1e-9;  // wire.py:115
    ps.adv = ps.adv_on ? (float)adv : 0.0f;
    ps.adv64 = ps.adv_on ? adv : 0.0;
    if (SCALAR_LOADS) {  // refresh_convection()
        const float A = (float)(hot.per_env_geometry ? cold->g.f64[(int64_t)WEDM_G_A_SURF * cold->s.stride + e]
                                                     : opaque_const(cold->p)->a_surf);
        ps.conv_base = s.h_base * A;
        ps.conv_zone = s.h_zone * A;
    } else {
        refresh_convection(hot, cold, e, s, ps);
    }
}

// --------------------------------------------------- scalar prelude (modules 1-4a)
// wire_edm.py:117-121 latch, ignition.py:175-319, material.py:79-174,
// dielectric.py:82-163, wire.py:271-312.  Returns the per-step stencil coefficients.
// Running get_crater_statistics (material.py:207-227), kept in HBM.  No-return atomics: the
// wave does not wait for memory (a load-modify-store here stalled the wave for two HBM round
// trips at almost every third microsecond, since some lane of a wave nearly always sparks:
// -5 % on the 400-segment workload).  One lane per environment updates, in spark order, so the
// sums are the same sequence of IEEE additions as the oracle's.
__device__ __forceinline__ void crater_stats_update(double* st, int64_t sd, double vol) {
    (void)__hip_atomic_fetch_add(st + WEDM_S_CRATER_SUM * sd, vol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_add(st + WEDM_S_CRATER_SUMSQ * sd, vol * vol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_min(st + WEDM_S_CRATER_MIN * sd, vol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_max(st + WEDM_S_CRATER_MAX * sd, vol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// `writer`: the one lane of an environment's L lanes that updates per-environment global memory
// beyond the state blocks (the running statistics).
struct QuietTry {  // what a failed quiet_prelude() hands on: the step's Philox words, if it drew them (wave-uniform flag)
    W4 w;
    bool have_w;
};

// REPLAY: the variates come from the caller's table (wedm_bind_rng_replay) instead of Philox — the reference's own
// draws, so that a native-seed run of the reference can be followed on the device (validation mode, global kernel).
// SCOLD / lt: the single-microsecond stream kernel's flavour -- cold parameters by scalar loads, crater tables from the
// lanes' registers (ColdParams, LaneTables): no vector load on the path a fresh spark takes.
template <bool REPLAY = false, bool SCOLD = false>
__device__ __forceinline__ Coef scalar_prelude(const Hot& p, const ColdRef cold, const Geom& g, int64_t e,
                                               uint32_t gid, Env& s, Persist& ps, bool writer,
                                               const QuietTry& qt = QuietTry{W4{0u, 0u, 0u, 0u}, false},
                                               const LaneTables* lt = nullptr, const LaneTables* lt1 = nullptr) {
    double rv[WEDM_REPLAY_SLOTS] = {0.0, 0.0, 0.0, 0.0, 0.0};
    // The cold parameter block and the pointer block, fetched ONCE per call (laundered: nothing read through them can
    // be hoisted out of the microsecond loop).  Every rare branch below used to launder its own copy: two dependent
    // scalar loads (kernel-argument segment, then the field) and a wait at each of ~15 sites of a fresh-spark step.
    const ColdPtr cc0 = cold.get();
    const typename ColdParams<SCOLD>::type c0 = ColdParams<SCOLD>::get(cc0->p);
    // the uniform constants of the rare branches, requested in one batch (scalar loads: one wait at the first use)
    const double seg = c0->segment_len, eff = c0->plasma_efficiency, rho_elec = c0->rho_elec, jg_u = c0->joule_geom;
    const double h_u = c0->workpiece_height, kerf_u = c0->kerf_base;
    const double ref_gap = c0->reference_gap, obstruction = c0->debris_obstruction_coeff;
    const int zs_u = c0->zone_start;
    if (REPLAY) {
        const ColdPtr c = cc0;
        int64_t step = (int64_t)s.time / p.dt_us;
        if (step >= c->replay_steps) { s.err = 1; step = c->replay_steps - 1; }
        const double* rp = c->replay + step * (int64_t)WEDM_REPLAY_SLOTS * c->s.stride + e;
#pragma unroll
        for (int q = 0; q < WEDM_REPLAY_SLOTS; ++q) rv[q] = rp[(int64_t)q * c->s.stride];
    }
    // ---- control-step latch (wire_edm.py:117-121,162-170)
    s.ctrl = s.tss >= p.servo_interval;
    if (s.ctrl) {
        const ColdPtr c = cc0;
        s.tdelta = c->a.servo[e];
        s.tvolt = c->a.target_voltage[e];
        s.mode = c->a.current_mode[e];
        s.on = c->a.on_time[e];
        s.off = c->a.off_time[e];
        s.tss = 0;
        s.ipk = peak_current(cold, s.mode, e);
    }
    const uint32_t t = (uint32_t)s.time, ep = (uint32_t)s.episode;

#ifdef WEDM_ABL_NO_IGN
    if (false) {
#else
    if (!p.disable_ignition) {
#endif
        // ---- short-circuit detection (ignition.py:197-245), written branch-free: in a
        // wave the lanes are in different generator states, so every per-lane `if` would be
        // executed by the whole wave anyway; selects avoid the exec-mask bookkeeping.  Real
        // branches remain only around rare, heavy work (exp, Philox, cold parameters).
        const bool timer_r = s.rnd_rem > 0;                 // random-short timer runs (checked first)
        const bool timer_d = !timer_r && (s.deb_rem > 0);   // debris-short timer runs
        const bool timers = timer_r || timer_d;
        const double d0 = s.wp - s.x;                       // unclamped gap (ignition.py:353)
        const double gap = d0 > 0.0 ? d0 : 0.0;
        const bool hard = gap < p.hard_short_gap;           // ignition.py:127-128
        double crit = p.base_critical_density + p.gap_coefficient * gap;
        crit = crit < p.max_critical_density ? crit : p.max_critical_density;
        const double ex = -p.sigmoid_steepness * (s.rho - crit);
        // every uniform is >= 2^-33 and 1/(1+e^ex) < 2^-33 for ex > 24: the roll cannot succeed,
        // so the exponential need not be evaluated there (same decision, exactly)
        double p_d = (hard || ex < -500) ? 1.0 : 0.0;
        // (a native NumPy uniform can be arbitrarily small: with injected variates the whole branch of ignition.py:136-146)
        if (!timers && !hard && ex <= (REPLAY ? 500.0 : 24.0) && ex >= -500) p_d = 1.0 / (1.0 + portable_exp(ex));
        double p_r = 0.0;
        if (p.has_random_short) {  // ignition.py:221-230; max_probability == 0 by default
            const auto c = c0;
            if (gap >= c->random_short_max_gap) p_r = 0.0;
            else if (gap <= c->random_short_min_gap) p_r = c->random_short_max_probability;
            else
                p_r = (1.0 - (gap - c->random_short_min_gap) / (c->random_short_max_gap - c->random_short_min_gap)) *
                      c->random_short_max_probability;
        }
        const bool idle = s.state == 0;
        W4 w{0u, 0u, 0u, 0u};
#ifndef WEDM_ABL_NO_PHILOX
        // (the words are a pure function of key, time, episode and environment: when the quiet attempt of this
        // step has already drawn them for the whole wave they are taken over instead of being computed again)
        if (qt.have_w) w = qt.w;
        else if (!timers && (p_d > 0.0 || p_r > 0.0 || idle)) w = philox4(s.key0, s.key1, t, ep, gid, 0u);
#else
        w = W4{t * 2654435761u + gid, 1u, 0xffffffffu - (t ^ gid) * 40503u, 7u};
#endif
        // p_d and p_r are exactly 0 in the ordinary gap regime: the rolls cannot succeed and the
        // integer -> double conversions are skipped for the whole wave
        bool new_d = false, new_r = false;
        if (__any(!timers && (p_d > 0.0 || p_r > 0.0))) {
            new_d = !timers && ((REPLAY ? rv[WEDM_RS_DEBRIS_ROLL] : u32_to_unit(w.x)) < p_d);
            new_r = !timers && !new_d && ((REPLAY ? rv[WEDM_RS_RANDOM_ROLL] : u32_to_unit(w.y)) < p_r);
        }
        if (new_d || new_r) {  // rare: a short begins (durations are cold parameters)
            const auto c = c0;
            if (new_d) s.deb_rem = c->debris_short_duration;
            else s.rnd_rem = c->random_short_duration;
        }
        s.rnd_rem = timer_r ? s.rnd_rem - 1 : s.rnd_rem;
        s.deb_rem = timer_d ? s.deb_rem - 1 : s.deb_rem;
        const bool shrt = timers || new_d || new_r;
        s.is_short = shrt ? 1 : 0;

        // `x or default` getters (ignition.py:329-343)
        const double Vt = s.tvolt != 0.0 ? s.tvolt : p.default_target_voltage;
        const double on = s.on != 0.0 ? s.on : p.default_on_time;
        const double off = s.off != 0.0 ? s.off : p.default_off_time;
        const double Vs = Vt * p.spark_voltage_factor;
        const double Ipk = s.ipk;

        // ---- spark state machine (ignition.py:186-195, 247-319) as selects
        const bool spk = s.state == 1, pls = s.state == -1, rst = s.state == -2;
        const double lam = p.ln2 / (p.ignition_a * (d0 * d0) + p.ignition_b * d0 + p.ignition_c);
        const bool ign = idle && !shrt && ((REPLAY ? rv[WEDM_RS_IGNITION_ROLL] : u32_to_unit(w.z)) < lam);   // _should_ignite
        const bool to_pulse = idle && shrt;                           // short during idle -> pulse
        const int32_t dur1 = s.dur + 1;
        const double ddur = (double)dur1;
        const bool end_on = (spk || pls) && (ddur >= on);             // spark / pulse finished
        const bool end_rest = rst && (ddur >= on + off);              // rest finished
        const bool burning = (ign || to_pulse) || ((spk || pls) && !end_on);
        double V = shrt ? 0.0 : s.V;                                  // ignition.py:182-183
        V = (idle && !shrt) ? (ign ? Vs : Vt) : V;
        V = (spk && !shrt) ? (end_on ? 0.0 : Vs) : V;
        V = (rst && !shrt) ? (end_rest ? Vt : 0.0) : V;
        s.V = V;
        s.I = burning ? Ipk : 0.0;
        // `_get_peak_current` is called exactly where `burning` holds; with a latched mode it fills the module's current
        // cache (ignition.py:98-113).  The first such call after a mode was latched falls on an ignition, a short pulse
        // or the latch step itself, and every one of those runs this general prelude: flag WEDM_B_MODE_CACHED.
        if (writer && burning && s.mode > 0 && (ign || to_pulse || s.ctrl))
            cc0->s.i8[(int64_t)WEDM_B_MODE_CACHED * cc0->s.stride + e] = 1;
        if (ign) {  // rare: spark location, Generator.uniform(0, h)
            const double h = p.per_env_geometry ? cc0->g.f64[(int64_t)WEDM_G_HEIGHT * cc0->s.stride + e] : h_u;
            s.y = REPLAY ? rv[WEDM_RS_SPARK_Y] : 0.0 + (h - 0.0) * u32_to_unit(w.w);
        }
        s.y = (to_pulse || end_rest) ? __builtin_nan("") : s.y;
        s.dur = idle ? ((ign || to_pulse) ? 0 : s.dur) : (end_rest ? 0 : dur1);
        s.state = ign ? 1 : to_pulse ? -1 : end_on ? -2 : end_rest ? 0 : s.state;
    }  // !disable_ignition
    const bool fresh = (s.state == 1) && (s.dur == 0);  // material.py:83, dielectric.py:95

    // `lt` / `lt1` (stream kernel): the crater-table entries of this lane's mode AS IT WAS BEFORE THIS STEP and of mode I1,
    // looked up by the caller across lanes; a step that latches a new mode reads the tables in memory
    const bool use_lt = lt != nullptr && !__any(s.ctrl != 0);

    // ---- material removal (material.py:79-174)
    if (fresh) {
        const ColdPtr cc = cc0;
        const Tables tb{cc->tb.mode_current, cc->tb.crater_mean, cc->tb.crater_std, cc->tb.crater_depth, cc->tb.crater_valid};
        int m = s.mode <= 0 ? 1 : s.mode;  // None (0, or -1: see peak_current) -> "I1" (material.py:104-105)
        const bool in_range = m >= 1 && m <= WEDM_MAX_MODE;
        if (!in_range) m = 1;
        // everything this branch reads from memory is requested here, before the crater normal (~230 instructions
        // with no memory access) is computed: one latency instead of a chain of them
        const int32_t valid = use_lt ? lt->valid : tb.crater_valid[m];
        double mean = use_lt ? lt->mean : tb.crater_mean[m], sd = use_lt ? lt->sd : tb.crater_std[m],
               depth = use_lt ? lt->depth : tb.crater_depth[m];
        const double kerf_base = p.per_env_geometry ? cc0->g.f64[(int64_t)WEDM_G_KERF_BASE * cc0->s.stride + e] : kerf_u;
        const double h = p.per_env_geometry ? cc0->g.f64[(int64_t)WEDM_G_HEIGHT * cc0->s.stride + e] : h_u;
        double* const clog = cc->s.crater_log;
        const int64_t clog_cap = cc->s.crater_log_capacity, sstride = cc->s.stride;
        double* const stats = cc->s.stats;
        if (!in_range || !valid) {  // unknown mode (the reference raises ValueError, material.py:108-113)
            s.err = 1;
            if (use_lt) { mean = lt1->mean; sd = lt1->sd; depth = lt1->depth; }
            else { mean = tb.crater_mean[1]; sd = tb.crater_std[1]; depth = tb.crater_depth[1]; }
        }
        double vol;
        if (REPLAY) {
            vol = rv[WEDM_RS_CRATER_UM3];
        } else {
            const double z = philox_std_normal(s.key0, s.key1, t, ep, gid);
            vol = mean + sd * z;
        }
        if (!(vol > 0)) vol = 0;
        if (writer && clog && clog_cap > 0)  // crater_volumes_um3.append (material.py:133)
            clog[((int64_t)s.sparks % clog_cap) * sstride + e] = vol;
        s.sparks += 1;
        if (writer && stats) crater_stats_update(stats + e, sstride, vol);
        double crater = vol / 1e9;
        s.last_crater = crater;
        if (crater > 0) {
            double kerf = kerf_base + depth / 1000.0;
            double dx = 0.0;
            if (kerf > 0 && h > 0) dx = crater / (kerf * h) * 1000.0;
            s.wp += dx;
        }
    } else {
        s.last_crater = 0.0;
    }

    // ---- dielectric / debris (dielectric.py:82-163)
#ifndef WEDM_ABL_NO_DIEL
    {
        double d = s.wp - s.x;
        double gap_um = d > 0.001 ? d : 0.001;
        s.cavity = g.cavity_coeff * (gap_um * 0.001);
        if (fresh && s.last_crater > 0) s.debris += s.last_crater;
        if (s.cavity > 0) {
            double q = s.debris / s.cavity;
            s.rho = q < 1.0 ? q : 1.0;
        } else {
            s.rho = 0.0;
        }
        if (__builtin_fabs(gap_um - s.last_gap) > 0.01 || __builtin_fabs(s.rho - s.last_rho) > 0.001) {
            double cube = cube_cr(gap_um / ref_gap);
            double gap_factor = cube < 1.0 ? cube : 1.0;
            double kd = obstruction * s.rho;
            double df;
            if (kd < 2.0) df = kd < 0.5 ? (1 - 0.5 * kd) / (1 + 0.5 * kd) : portable_exp(-kd);
            else df = portable_exp(-obstruction * s.rho);
            s.flow = gap_factor * df;
            s.last_gap = gap_um;
            s.last_rho = s.rho;
        }
        if (s.flow > 0.001 && s.debris > 0.001) {
            double nv = s.debris - p.debris_removal_per_us * s.flow;
            s.debris = nv > 0.0 ? nv : 0.0;
        }
    }
#endif

    // ---- wire prelude (wire.py:271-312, 349-374)
    Coef cf;
    {
        const double I = s.I, I2 = I * I;
        // (`!s.broken`: with keep_stepping_terminated a broken wire's module returns before it looks at its convection cache,
        // wire.py:260-261; elsewhere a broken wire never gets here)
        if (!s.broken && __builtin_fabs(s.flow - s.wire_last_flow) > 0.01) {
            const auto c = c0;
            double ve = c->convection_velocity_factor * s.unwind;
            ve = ve > -0.9 ? ve : -0.9;
            double hb = c->base_convection * (1.0 + ve);
            double fl = 0.1 * c->base_convection;
            hb = fl > hb ? fl : hb;
            double he = hb * (1.0 + c->convection_flow_enhancement * s.flow);
            s.h_base = (float)hb;
            s.h_zone = (float)he;
            s.wire_last_flow = s.flow;
            refresh_convection<SCOLD>(p, cold, e, s, ps);
        }
        cf.pidx = -1;
        cf.q = 0.0f;
        cf.q64 = 0.0;
        if (s.state == 1 && s.y == s.y) {
            const int zone_start = p.per_env_geometry ? cc0->g.i32[(int64_t)WEDM_GI_ZONE_START * cc0->s.stride + e] : zs_u;
            int idx = seg != 0 ? zone_start + spark_cell_offset(s.y, seg) : zone_start;
            if (idx >= 0 && idx < g.n_seg) {
                cf.pidx = idx;
                cf.q64 = eff * s.V * I;
                cf.q = (float)cf.q64;
            }
        }
        cf.joule_on = I2 > 1e-6;
        cf.jf = 0.0f;
        cf.jf64 = 0.0;
        if (cf.joule_on) {
            const double joule_geom = p.per_env_geometry ? cc0->g.f64[(int64_t)WEDM_G_JOULE_GEOM * cc0->s.stride + e] : jg_u;
            cf.jf64 = joule_geom * I2 * rho_elec;
            cf.jf = (float)cf.jf64;
        }
    }
    return cf;
}

// ------------------------------------------------- quiet-step fast path (wave-uniform)
// In an ordinary microsecond nothing discrete happens to an environment: no short timer runs,
// the generator idles or rests, the action is not latched, the gap is far from a hard short, the
// debris sigmoid cannot fire (exponent > 24), the flow / convection caches stay valid and the
// ignition roll fails.  For such a step scalar_prelude() reduces to the straight line below
// (every assignment is what the general code computes under exactly these conditions).  The
// fast path is taken only if EVERY live lane of the wave qualifies and none ignites; nothing is
// written before that is known, so otherwise the general path runs on untouched state.
// Returns true when the step was handled (the stencil coefficients are then {0, 0, off, none}).
// DENSE (the instantiation for densely sparking batches, chosen per launch from the spark density of the previous ones):
// the same straight line also carries a spark that ignited in an EARLIER microsecond and keeps burning or ends now
// (ignition.py:270-287 with no short: duration + 1, voltage = spark voltage or 0, current = peak or 0, state 1 -> -2
// when the ON time is over) and hands its plasma / Joule coefficients to the stencil (wire.py:284-301) — for such a
// lane scalar_prelude() computes exactly these assignments.  Only an igniting lane (fresh spark: crater, debris, cache
// refresh), a short, or a control-step latch still sends the wave through the general path.  Bit-identical; it costs
// registers, so batches that spark rarely run the instantiation without it.
template <bool DENSE>
__device__ __forceinline__ bool quiet_prelude_t(const Hot& p, const ColdRef cold, const Geom& g, int64_t e, uint32_t gid,
                                                Env& s, QuietTry& qt, Coef& cf) {
    qt.have_w = false;
    if (p.disable_ignition || p.has_random_short) return false;
    const bool live = !s.done;
    const double d0 = s.wp - s.x;                       // unclamped gap (>= hard_short_gap > 0.001 below)
    const bool idle = s.state == 0, rest = s.state == -2;
    const bool spk = DENSE && s.state == 1;             // ignited in an earlier microsecond
    double crit = p.base_critical_density + p.gap_coefficient * d0;
    crit = crit < p.max_critical_density ? crit : p.max_critical_density;
    const double ex = -p.sigmoid_steepness * (s.rho - crit);
    bool q = (s.tss < p.servo_interval) && (s.rnd_rem == 0) && (s.deb_rem == 0) && (idle || rest || spk) &&
             (d0 >= p.hard_short_gap) && (ex > 24.0);
    // dielectric (dielectric.py:87-139): this step's density and the cache test
    const double cavity = g.cavity_coeff * (d0 * 0.001);
    double rho = 0.0;
    if (cavity > 0) {
        const double qq = s.debris / cavity;
        rho = qq < 1.0 ? qq : 1.0;
    }
    q = q && !(__builtin_fabs(d0 - s.last_gap) > 0.01 || __builtin_fabs(rho - s.last_rho) > 0.001) &&
        !(__builtin_fabs(s.flow - s.wire_last_flow) > 0.01) && (d0 > 0.001);
    if (!__all(q || !live)) return false;
    // ignition roll of the idle lanes (ignition.py:321-327)
    bool ign = false;
    if (__any(idle && live)) {
        const double lam = p.ln2 / (p.ignition_a * (d0 * d0) + p.ignition_b * d0 + p.ignition_c);
        const W4 w = philox4(s.key0, s.key1, (uint32_t)s.time, (uint32_t)s.episode, gid, 0u);
        ign = idle && live && (u32_to_unit(w.z) < lam);
#ifndef WEDM_NO_PHILOX_REUSE
        qt.w = w;          // every lane of the wave computed its words: the general path need not repeat them
        qt.have_w = true;
#endif
    }
    if (__any(ign)) return false;
    bool burning = false;
    if (live) {
        const double Vt = s.tvolt != 0.0 ? s.tvolt : p.default_target_voltage;
        const double on = s.on != 0.0 ? s.on : p.default_on_time;
        const double off = s.off != 0.0 ? s.off : p.default_off_time;
        const int32_t dur1 = s.dur + 1;
        const bool end_rest = rest && ((double)dur1 >= on + off);
        const bool end_on = spk && ((double)dur1 >= on);
        burning = spk && !end_on;
        s.ctrl = 0;
        s.is_short = 0;
        double V = idle ? Vt : (end_rest ? Vt : 0.0);
        if (DENSE) V = spk ? (end_on ? 0.0 : Vt * p.spark_voltage_factor) : V;
        s.V = V;
        s.I = burning ? s.ipk : 0.0;
        s.y = end_rest ? __builtin_nan("") : s.y;
        s.dur = idle ? s.dur : (end_rest ? 0 : dur1);
        s.state = end_rest ? 0 : (end_on ? -2 : s.state);
        s.last_crater = 0.0;
        s.cavity = cavity;
        s.rho = rho;
        if (s.flow > 0.001 && s.debris > 0.001) {
            const double nv = s.debris - p.debris_removal_per_us * s.flow;
            s.debris = nv > 0.0 ? nv : 0.0;
        }
    }
    if (DENSE && __any(burning)) {  // wire.py:284-301, 96-100 for the lanes that keep burning
        // the five cold constants of this block in one batch of scalar loads (one wait instead of four chains of two)
        const ColdPtr cc0 = cold.get();
        const ParamsPtr c = opaque(cc0->p);
        const double seg = c->segment_len, eff = c->plasma_efficiency, rho_elec = c->rho_elec;
        const int zone_start = (p.per_env_geometry && burning) ? cc0->g.i32[(int64_t)WEDM_GI_ZONE_START * cc0->s.stride + e] : c->zone_start;
        const double joule_geom = (p.per_env_geometry && burning) ? cc0->g.f64[(int64_t)WEDM_G_JOULE_GEOM * cc0->s.stride + e] : c->joule_geom;
        if (burning && s.y == s.y) {
            const int idx = seg != 0 ? zone_start + spark_cell_offset(s.y, seg) : zone_start;
            if (idx >= 0 && idx < g.n_seg) {
                cf.pidx = idx;
                cf.q64 = eff * s.V * s.I;
                cf.q = (float)cf.q64;
            }
        }
        if (burning) {
            const double I2 = s.I * s.I;
            cf.joule_on = I2 > 1e-6;
            if (cf.joule_on) {
                cf.jf64 = joule_geom * I2 * rho_elec;
                cf.jf = (float)cf.jf64;
            }
        }
    }
    return true;
}

__device__ __forceinline__ bool quiet_prelude(const Hot& p, const Geom& g, uint32_t gid, Env& s, QuietTry& qt) {
    Coef unused{0.0f, 0.0f, 0, -1};
    return quiet_prelude_t<false>(p, ColdRef{nullptr}, g, 0, gid, s, qt, unused);
}

__device__ __forceinline__ bool quiet_prelude(const Hot& p, const Geom& g, uint32_t gid, Env& s) {
    QuietTry qt;
    return quiet_prelude(p, g, gid, s, qt);
}

// a - 2*b exactly as the reference rounds it: 2*b is exact in binary floating point, so
// `a - (2*b)` rounds once, which is what one fused multiply-add fma(-2, b, a) computes.  The only
// FMA in the stencil: everywhere else a product is rounded before it is added (wire.py:91-120).
__device__ __forceinline__ float sub_twice(float a, float b) { return __builtin_fmaf(-2.0f, b, a); }
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v sub_twice(f2v a, f2v b) {
    const f2v m2 = {-2.0f, -2.0f};
    return __builtin_elementwise_fma(m2, b, a);
}

// one cell of wire.py:58-123, float32 op for op; tm1/tc/tp1 are OLD temperatures
__device__ __forceinline__ float stencil_cell(int i, int n_seg, float tm1, float tc, float tp1, const Geom& g,
                                              const Coef& c, const Persist& ps, float tref, float alpha,
                                              float tdiel) {
    float d;
    if (i < n_seg - 1) {
        float a = sub_twice(tm1, tc);  // T[i-1] - 2*T[i]
        d = g.k * (a + tp1);
    } else {
        d = g.k * (tm1 - tc);
    }
    if (c.joule_on && i >= g.cb && i <= g.ct) {
        float rho_T = 1.0f + alpha * (tc - tref);
        d = d + c.jf * rho_T;
    }
    if (i == c.pidx) d = d + c.q;
    bool in_zone = (i >= g.az_start) && (i < g.az_end);
    float conv = in_zone ? ps.conv_zone : ps.conv_base;
    d = d - conv * (tc - tdiel);
    if (ps.adv_on) d = d + ps.adv * (tm1 - tc);
    return tc + d * g.tuf;
}

// The same cell as Numba types wire.py:58-123 (`@njit`: float32 array elements promoted to float64 in
// every expression, the result rounded where it is stored into the float32 `dT_dt[i]` / `T[i]`; without
// fastmath re-association) — wedm_params.stencil_mode 1.  h_base / h_zone are the float32 `h_eff_zone`
// entries (wire.py:205), everything else float64 constants.
struct StencilF64 {  // the float64 constants of that typing (read from the cold parameters by the F64 instantiations)
    double tref, alpha, tdiel;
};
__device__ __forceinline__ float stencil_cell_f64(int i, int n_seg, float tm1, float tc, float tp1, const Geom& g,
                                                  const Coef& c, const Persist& ps, const StencilF64& h, float h_base,
                                                  float h_zone) {
    const double m = (double)tm1, t = (double)tc;
    float d;
    if (i < n_seg - 1) d = (float)(g.k64 * (m - 2.0 * t + (double)tp1));
    else d = (float)(g.k64 * (m - t));
    if (c.joule_on && i >= g.cb && i <= g.ct) {
        const double rho_T = 1.0 + h.alpha * (t - h.tref);
        d = (float)((double)d + c.jf64 * rho_T);
    }
    if (i == c.pidx) d = (float)((double)d + c.q64);
    const bool in_zone = (i >= g.az_start) && (i < g.az_end);
    const double conv = (double)(in_zone ? h_zone : h_base) * g.a64;
    d = (float)((double)d - conv * (t - h.tdiel));
    if (ps.adv_on) d = (float)((double)d + ps.adv64 * (m - t));
    return (float)(t + (double)d * g.tuf64);
}

// An INTERIOR cell (1 <= i <= n - 2) in that typing, coefficients handed in (the fused kernel's tile code, stencil_mode 1):
// the same sequence of float64 expressions and float32 roundings as stencil_cell_f64, with the Joule and advection terms
// always applied -- their coefficient is 0.0 where the reference skips them, and (float)((double)d + 0.0 * x) == d.
__device__ __forceinline__ float interior_cell_f64(float tm1, float tc, float tp1, double k64, double tuf64, double conv64,
                                                   double tdiel64, double adv64, double jfe64, double alpha64, double tref64) {
    const double m = (double)tm1, t = (double)tc;
    float d = (float)(k64 * (m - 2.0 * t + (double)tp1));
    const double rho_T = 1.0 + alpha64 * (t - tref64);
    d = (float)((double)d + jfe64 * rho_T);
    d = (float)((double)d - conv64 * (t - tdiel64));
    d = (float)((double)d + adv64 * (m - t));
    return (float)(t + (double)d * tuf64);
}

// ------------------------------------------- scalar epilogue (modules 4b, 5, env)
// wire.py:376-388, wire_edm.py:129-146,172-179, mechanics.py:79-114
// Two halves.  The MONITOR is everything that reads the step's maximum temperature (wire.py:376-388: time above the critical
// temperature, the break test); the MOTION is what follows a step whose wire did not break (the driver's voltage sum,
// mechanics, clocks, termination) and reads nothing the monitor writes except `broken`.  scalar_epilogue() is the two in the
// reference's order; a kernel whose scalar physics runs ahead of its stencil (wedm_step_served) calls the motion half first
// where it can PROVE that the step cannot break the wire, and the monitor half when the maximum arrives.
__device__ __forceinline__ void epilogue_voltage_sum(Env& s) {
    // the driver appends state.voltage after EVERY step(), the early-return one included
    // (experiments/run_simulation.py:256-264): running sum in step order
#ifndef WEDM_ABL_NO_VACC
    s.vacc = s.vacc + s.V;
#endif
}

// wire.py:376-388 (`_check_wire_breaking`); returns with s.broken set when the wire broke in this step
__device__ __forceinline__ void epilogue_monitor(const Hot& p, Env& s, float tmax) {
    int32_t tcrit = tmax > p.tcrit ? s.tcrit + 1 : 0;
    // keep_stepping_terminated only (elsewhere a broken wire is frozen and never gets here): the wire module returns at
    // once on a broken wire (wire.py:260-261) -- no stencil (the caller's walk left this lane's wire alone and its `tmax`
    // is meaningless), no temperature monitor
    if (__any(s.broken != 0)) {
        tmax = s.broken ? s.tmax : tmax;
        tcrit = s.broken ? s.tcrit : tcrit;
    }
    s.tmax = tmax;
    s.tcrit = tcrit;
    if (tmax > p.tbreak) s.broken = 1;
}

// mechanics.py:79-114, wire_edm.py:135-146,172-179 for a step whose wire is not broken
__device__ __forceinline__ void epilogue_motion(const Hot& p, Env& s) {
#ifndef WEDM_ABL_NO_MECH
    {
        double x = s.x, v = s.v, a_nom;
        if (p.control_mode == 0) {
            double x_error = x - (x + s.tdelta);
            a_nom = p.damping_coeff * v + p.stiffness_coeff * x_error;
        } else {
            a_nom = -p.omega_n * (v - s.tdelta);
        }
        if (a_nom > p.max_acceleration) a_nom = p.max_acceleration;
        else if (a_nom < -p.max_acceleration) a_nom = -p.max_acceleration;
        double da = a_nom - s.prev_a;
        if (da > p.max_jerk_dt) da = p.max_jerk_dt;
        else if (da < -p.max_jerk_dt) da = -p.max_jerk_dt;
        double a = s.prev_a + da;
        s.prev_a = a;
        v += a * p.dt_s;
        if (v > p.max_speed) v = p.max_speed;
        else if (v < -p.max_speed) v = -p.max_speed;
        x += v * p.dt_s;
        s.v = v;
        s.x = x;
    }
#endif
    s.time = (int32_t)((uint32_t)s.time + (uint32_t)p.dt_us);  // low word of the 64-bit clock: wraps (store_time_hi)
    s.tss += p.dt_us;
    s.tsov = (int32_t)((uint32_t)s.tsov + (uint32_t)p.dt_us);  // never reset by the reference either: in step with `time`
    if (s.state == 1) { s.tsi += p.dt_us; s.tse = 0; }
    else { s.tse += p.dt_us; s.tsi = 0; }
    if (s.x > s.wp + 100) { s.broken = 1; s.done = p.done_value; }
    else if (s.wp >= s.tpos) { s.reached = 1; s.done = p.done_value; }
}

__device__ __forceinline__ void scalar_epilogue(const Hot& p, Env& s, float tmax) {
    epilogue_voltage_sum(s);
    epilogue_monitor(p, s, tmax);
    if (s.broken) {  // early return before mechanics and clocks
        s.done = p.done_value;
        return;
    }
    epilogue_motion(p, s);
}

// The wire of a lane whose wire is broken stays as it is: around its walk a kernel ORs `broken` into the lane's frozen
// flag (`freeze_wire`) and afterwards takes it out again (`unfreeze_wire`).  Without keep_stepping_terminated both are
// no-ops (a broken wire is frozen anyway: done_value 1); with it they make the walk skip exactly the lanes whose wire
// module would return at once (wire.py:260-261) while prelude and epilogue go on running.
// Lanes past the end of the batch carry done = WEDM_DEAD_LANE (bit 1), which neither of the two ever clears: such a lane
// never runs physics and never stores, in either mode.
#define WEDM_DEAD_LANE 2
__device__ __forceinline__ void freeze_wire(Env& s) { s.done |= s.broken; }
__device__ __forceinline__ void unfreeze_wire(const Hot& p, Env& s) { s.done &= (p.done_value ? ~0 : WEDM_DEAD_LANE); }

__device__ __forceinline__ void write_obs(const ColdRef cold, int64_t e, const Env& s) {
    const ColdPtr c = cold.get();
    float* obs = c->s.obs;
    if (!obs || c->p->obs_dim < 8) return;
    const int64_t stride = c->s.stride;
    float* o = obs + e;
    o[0 * stride] = (float)(s.wp - s.x);
    o[1 * stride] = (float)s.v;
    o[2 * stride] = (float)s.V;
    o[3 * stride] = (float)s.I;
    o[4 * stride] = (float)s.state;
    o[5 * stride] = (float)s.rho;
    o[6 * stride] = (float)s.flow;
    o[7 * stride] = s.tmax;
}

// Control step (wire_edm.py:117-121): the observation, the interval's voltage sum (row
// WEDM_F_VOLT_SUM: this step included, plus the previous control step's sample the accumulator was
// restarted from) and the restart of the accumulator.  Every lane of an environment restarts its
// replica; only `writer` touches memory.
__device__ __forceinline__ void control_step_outputs(const ColdRef cold, int64_t e, Env& s, bool writer) {
    if (writer) {
        write_obs(cold, e, s);
        const ColdPtr c = cold.get();
        c->s.f64[(int64_t)WEDM_F_VOLT_SUM * c->s.stride + e] = s.vacc;
    }
    s.vacc = s.V;
}

// wedm_params.autoreset: what wedm_reset_kernel(mask = DONE, reseed = 0) writes for one environment,
// applied to the registers at the start of a launch (next-step autoreset).  `writer` also clears the
// per-environment memory outside the register state (statistics, observation) and stores the rows
// store_env() never writes.  The caller sets the environment's wire to the spool temperature.
__device__ __forceinline__ void reinit_env(const ColdRef cold, int64_t e, Env& s, bool writer) {
    const ColdPtr c = cold.get();
    const ParamsPtr p = opaque(c->p);
    const int32_t episode = s.episode + 1;
    const uint32_t k0 = s.key0, k1 = s.key1;
    const float spool = (float)p->spool_T;
    // reset_semantics 1: the reference's reset() builds a new EDMState only (wire_edm.py:106-114); what its module objects
    // hold lives on: short timers, debris volume, flow / density / convection caches, `prev_accel`, the crater list
    const bool keep_modules = p->reset_semantics != 0;
    s.wp = p->initial_gap; s.x = 0.0; s.v = 0.0;
    s.rho = 0.0; s.V = 0.0; s.I = 0.0; s.y = __builtin_nan(""); s.last_crater = 0.0; s.cavity = 0.0;
    s.tdelta = 0.0; s.tvolt = 0.0; s.on = 0.0; s.off = 0.0; s.tpos = p->target_cutting_distance; s.unwind = 0.2;
    s.vacc = 0.0;
    s.tmax = spool;
    s.time = 0; s.tss = 0; s.tsov = 0; s.tsi = 0; s.tse = 0; s.dur = 0; s.tcrit = 0;
    // state.current_mode = None: 0, or -1 where the module's surviving current cache names a mode (peak_current)
    s.mode = (keep_modules && c->s.i8[(int64_t)WEDM_B_MODE_CACHED * c->s.stride + e]) ? -1 : 0;
    s.episode = episode; s.key0 = k0; s.key1 = k1;
    s.state = 0; s.is_short = 0; s.broken = 0; s.reached = 0; s.done = 0; s.ctrl = 0; s.err = 0;
    if (!keep_modules) {
        s.prev_a = 0.0; s.debris = 0.0; s.flow = 0.0; s.last_gap = -1.0; s.last_rho = -1.0; s.wire_last_flow = 0.0;
        s.h_base = 0.0f; s.h_zone = 0.0f; s.rnd_rem = 0; s.deb_rem = 0; s.sparks = 0;
    }
    if (writer) {
        const int64_t stride = c->s.stride;
        *WEDM_ROW(c->s.f64, WEDM_F_WORKPIECE_POS) = s.wp;   // the reward's "position at the start of the launch"
        *WEDM_ROW(c->s.f64, WEDM_F_TARGET_POS) = s.tpos;
        *WEDM_ROW(c->s.f64, WEDM_F_UNWIND_VEL) = s.unwind;
        *WEDM_ROW(c->s.f64, WEDM_F_VOLT_SUM) = 0.0;
        *WEDM_ROW(c->s.i32, WEDM_I_EPISODE) = episode;
        *WEDM_ROW(c->s.i32, WEDM_I_TIME) = 0;      // store_time_hi's "low word at the start of the launch"
        *WEDM_ROW(c->s.i32, WEDM_I_TIME_HI) = 0;
        if (!keep_modules) {
            *WEDM_ROW(c->s.i8, WEDM_B_MODE_CACHED) = 0;
            if (c->s.stats) {
                *WEDM_ROW(c->s.stats, WEDM_S_CRATER_SUM) = 0.0; *WEDM_ROW(c->s.stats, WEDM_S_CRATER_SUMSQ) = 0.0;
                *WEDM_ROW(c->s.stats, WEDM_S_CRATER_MIN) = __builtin_inf(); *WEDM_ROW(c->s.stats, WEDM_S_CRATER_MAX) = -__builtin_inf();
            }
        }
        if (c->s.obs)
            for (int q = 0; q < p->obs_dim; ++q) c->s.obs[(int64_t)q * stride + e] = 0.0f;
    }
}

// wedm_params.reward_mode 1, called by the writer lane right BEFORE store_env(): the workpiece-position
// row still holds the value the launch started from (reinit_env stored the reset value there).
__device__ __forceinline__ void write_reward(const ColdRef cold, int64_t e, const Env& s) {
    const ColdPtr c = cold.get();
    if (!c->s.reward) return;
    const double wp0 = c->s.f64[(int64_t)WEDM_F_WORKPIECE_POS * c->s.stride + e];
    const double pen = opaque(c->p)->reward_break_penalty;
    c->s.reward[e] = (float)(s.wp - wp0) - (float)pen * (s.broken ? 1.0f : 0.0f);
}

}  // namespace wedm
