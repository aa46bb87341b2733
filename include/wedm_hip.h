/*
 * wedm_hip.h — C-ABI of the MI355X batched Wire-EDM step.
 *
 * This is the drop-in boundary for ONE hot path of geduardo/SPARC (`wedm` 0.2.0):
 * the per-microsecond physics step of `WireEDMEnv.step()`
 * (reference src/wedm/envs/wire_edm.py:116-157 and the five module `update()`s it
 * calls: modules/ignition.py:175-195, material.py:79-96, dielectric.py:82-163,
 * wire.py:259-347 (+ the stencil wire.py:58-123), mechanics.py:79-114).
 *
 * The reference is pure Python and has no FFI; the binding a maintainer would add
 * is a `ctypes.CDLL` load of `libwedm_hip.so`.  Every entry
 * point below therefore cites the Python method it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; all pointers are DEVICE pointers unless noted;
 *   - the caller owns every byte of state (torch tensors on the host side); the
 *     library owns only the opaque handle and one small constant table;
 *   - every call returns an int32 status (WEDM_OK or a negative wedm_status);
 *     nothing throws across the boundary; wedm_last_error() gives the text;
 *   - launches are asynchronous on the `hipStream_t` passed as `void* stream`
 *     (NULL = the null stream); the library creates no streams and no threads;
 *   - a handle is not thread-safe: one handle per device per process.
 *
 * State layout: struct-of-arrays, field-major, environment-minor.  Field `f` of
 * environment `e` lives at  block[f * stride + e]  so that the 64 lanes of a
 * wavefront (64 consecutive environments) read 64 consecutive elements.
 * The wire temperature (float32) is QUAD-INTERLEAVED (ABI v4): four consecutive
 * segments of one environment form one 16-byte word, words are environment-minor,
 *     T[((seg >> 2) * stride + e) * 4 + (seg & 3)]          (WEDM_T_INDEX below),
 * i.e. T[ceil(n_seg_max / 4)][stride][4].  A lane that owns a run of consecutive
 * segments of one environment loads / stores it 16 bytes at a time
 * (global_load_dwordx4), and the 64 lanes of a wavefront still touch contiguous
 * 1-KB runs.  The cells of the last word past n_seg_max are padding: written at
 * reset (spool temperature), never read into a result, never rewritten by a step.
 * (ABI v3 had T[seg][env]: one dword per lane and instruction, which is what bound
 * the one-launch-per-microsecond kernels: DESIGN.md section 4.2.)
 */
#ifndef WEDM_HIP_H
#define WEDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WEDM_ABI_VERSION 4

/* element index of wire segment `seg` of environment `e` in the T block */
#define WEDM_T_INDEX(seg, stride, e) ((((int64_t)((seg) >> 2) * (int64_t)(stride) + (int64_t)(e)) << 2) + ((seg) & 3))
/* 16-byte words (rows of the T block) a wire of n_seg_max segments occupies */
#define WEDM_T_QUADS(n_seg_max) (((n_seg_max) + 3) >> 2)

/* ------------------------------------------------------------------ status */
typedef enum wedm_status {
    WEDM_OK = 0,
    WEDM_ERR_BAD_ARG = -1,     /* NULL pointer, non-positive size, bad enum      */
    WEDM_ERR_NOT_BOUND = -2,   /* wedm_step/reset before wedm_bind_state         */
    WEDM_ERR_HIP = -3,         /* a HIP runtime call failed (text in last_error) */
    WEDM_ERR_NO_DEVICE = -4,   /* no gfx950 device visible                       */
    WEDM_ERR_BAD_MODE = -5,    /* current_mode without crater data (material.py:108-113) */
    WEDM_ERR_UNSUPPORTED = -6  /* n_seg exceeds what the fused kernel can stage  */
} wedm_status;

/* ------------------------------------------------------- float64 state rows
 * One row per mutable Python-float attribute of EDMState (core/state.py:25-91)
 * plus the module-private floats the reference keeps outside the state
 * (dielectric.py:69-80, mechanics.py:60, wire.py:224,205).                   */
enum wedm_f64_field {
    WEDM_F_WORKPIECE_POS = 0,  /* state.workpiece_position        [um]   */
    WEDM_F_WIRE_POS,           /* state.wire_position             [um]   */
    WEDM_F_WIRE_VEL,           /* state.wire_velocity             [um/s] */
    WEDM_F_PREV_ACCEL,         /* MechanicsModule.prev_accel (mechanics.py:60) */
    WEDM_F_DEBRIS_VOLUME,      /* DielectricModule.debris_volume / state.debris_volume [mm^3] */
    WEDM_F_DEBRIS_DENSITY,     /* state.debris_density                   */
    WEDM_F_FLOW,               /* state.flow_rate == DielectricModule._last_flow_condition */
    WEDM_F_LAST_GAP,           /* DielectricModule._last_gap_um (init -1)       */
    WEDM_F_LAST_DENSITY,       /* DielectricModule._last_debris_density (init -1) */
    WEDM_F_WIRE_LAST_FLOW,     /* WireModule._last_flow_condition (init 0)      */
    WEDM_F_VOLTAGE,            /* state.voltage (None -> 0)              */
    WEDM_F_CURRENT,            /* state.current (None -> 0)              */
    WEDM_F_SPARK_Y,            /* state.spark_status[1]; NaN encodes None */
    WEDM_F_LAST_CRATER,        /* state.last_crater_volume        [mm^3] */
    WEDM_F_CAVITY,             /* state.cavity_volume             [mm^3] */
    WEDM_F_TARGET_DELTA,       /* state.target_delta   (latched action)  */
    WEDM_F_TARGET_VOLTAGE,     /* state.target_voltage (latched; 0 encodes None) */
    WEDM_F_ON_TIME,            /* state.ON_time        (latched; 0 encodes None) */
    WEDM_F_OFF_TIME,           /* state.OFF_time       (latched; 0 encodes None) */
    WEDM_F_TARGET_POS,         /* state.target_position                  */
    WEDM_F_UNWIND_VEL,         /* state.wire_unwinding_velocity          */
    WEDM_F_H_BASE,             /* WireModule.h_eff_zone outside the zone (float32 value) */
    WEDM_F_H_ZONE,             /* WireModule.h_eff_zone inside the zone  (float32 value) */
    WEDM_F_TMAX,               /* max(T) after the last step (float32 value)    */
    /* The driver's 1 ms voltage history (experiments/run_simulation.py:258-281) as a running sum:
     * VOLT_ACC = sum of state.voltage after every step since (and including) the last control
     * step, in step order; at a control step the kernels publish VOLT_SUM = VOLT_ACC (this step
     * included) and restart VOLT_ACC from this step's voltage.  With servo_interval = 1000 us
     * VOLT_SUM is exactly the sum of the <= 1001 samples `create_voltage_controller` averages
     * (run_simulation.py:262-270): this interval's microseconds plus the previous control step's. */
    WEDM_F_VOLT_ACC,
    WEDM_F_VOLT_SUM,
    WEDM_F64_COUNT
};

/* --------------------------------------------------------- int32 state rows */
enum wedm_i32_field {
    WEDM_I_TIME = 0,           /* state.time [us], LOW 32 bits (unsigned, wraps); the high word is WEDM_I_TIME_HI:
                                  the reference counts in unbounded Python ints (wire_edm.py:135).  Also Philox
                                  counter word 0 (an episode's variates repeat after 2^32 us = 71.6 simulated minutes) */
    WEDM_I_SINCE_SERVO,        /* state.time_since_servo                */
    WEDM_I_SINCE_OPEN_V,       /* state.time_since_open_voltage         */
    WEDM_I_SINCE_IGNITION,     /* state.time_since_spark_ignition       */
    WEDM_I_SINCE_SPARK_END,    /* state.time_since_spark_end            */
    WEDM_I_SPARK_DUR,          /* state.spark_status[2]                 */
    WEDM_I_RANDOM_SHORT_REM,   /* IgnitionModule.random_short_remaining */
    WEDM_I_DEBRIS_SHORT_REM,   /* IgnitionModule.debris_short_remaining */
    WEDM_I_TIME_CRITICAL,      /* state.time_in_critical_temp           */
    WEDM_I_CURRENT_MODE,       /* state.current_mode: n for "I<n>", 0 encodes None; -1 = None while the ignition module's
                                  current cache still names a mode of an earlier episode (reset_semantics 1 only) */
    WEDM_I_EPISODE,            /* resets seen by this environment (RNG counter word) */
    WEDM_I_KEY_LO,             /* Philox key, low  32 bits of the reset seed */
    WEDM_I_KEY_HI,             /* Philox key, high 32 bits of the reset seed */
    WEDM_I_SPARK_COUNT,        /* fresh sparks since reset (len(crater_volumes_um3), material.py:133) */
    WEDM_I_TIME_HI,            /* state.time >> 32: bumped by wedm_step when the low word wraps inside a launch (a launch
                                  advances an environment by less than 2^31 us: wedm_step refuses n_substeps * dt_us >= 2^31) */
    WEDM_I32_COUNT
};

/* ---------------------------------------------------------- int8 state rows */
enum wedm_i8_field {
    WEDM_B_SPARK_STATE = 0,    /* state.spark_status[0]: 0 idle, 1 spark, -1 short, -2 rest */
    WEDM_B_IS_SHORT,           /* state.is_short_circuit                */
    WEDM_B_WIRE_BROKEN,        /* state.is_wire_broken                  */
    WEDM_B_TARGET_REACHED,     /* state.is_target_distance_reached      */
    WEDM_B_DONE,               /* terminated: the environment is frozen until reset */
    WEDM_B_CTRL_STEP,          /* info["control_step"] of the LAST substep run */
    WEDM_B_ERROR,              /* sticky: 1 = fresh spark with a mode that has no crater data */
    WEDM_B_MODE_CACHED,        /* IgnitionModule._cached_current_mode is not None (ignition.py:79-81,98-113): a peak current
                                  has been looked up for a latched mode.  Survives a reset with reset_semantics 1; while
                                  state.current_mode is None, a set flag makes the lookup answer `default_current`
                                  instead of the fresh module's 60 A */
    WEDM_I8_COUNT
};

/* ------------------------------------------------ running statistics (optional)
 * Accumulated inside the kernels at every fresh spark with fire-and-forget float64 atomics
 * (one lane per environment, so the order of the additions is the order of the sparks); what
 * the reference computes from its `crater_volumes_um3` list in
 * `MaterialRemovalModule.get_crater_statistics` (material.py:207-227).  Count =
 * WEDM_I_SPARK_COUNT; mean = sum / n; std = sqrt(sumsq / n - mean^2).                        */
enum wedm_stat_field {
    WEDM_S_CRATER_SUM = 0,     /* sum of the sampled crater volumes            [um^3]   */
    WEDM_S_CRATER_SUMSQ,       /* sum of their squares                         [um^6]   */
    WEDM_S_CRATER_MIN,         /* smallest crater volume so far (+inf while n == 0)     */
    WEDM_S_CRATER_MAX,         /* largest crater volume so far  (-inf while n == 0)     */
    WEDM_STAT_COUNT
};

/* -------------------------------------- per-environment geometry (optional)
 * BASELINE config 5: workpiece_height / wire_diameter differ per environment.
 * When bound, these rows override the uniform values in wedm_params.       */
enum wedm_geom_f64_field {
    WEDM_G_HEIGHT = 0,         /* config.workpiece_height [mm]                       */
    WEDM_G_KERF_BASE,          /* base_overcut + wire_diameter (material.py:158-160) */
    WEDM_G_CAVITY_COEFF,       /* pi * r * h              (dielectric.py:62)         */
    WEDM_G_K_COND,             /* k * S / dy              (wire.py:174-176)          */
    WEDM_G_TUF,                /* dt / (rho c S dy)       (wire.py:195)              */
    WEDM_G_A_SURF,             /* 2 pi r dy               (wire.py:171)              */
    WEDM_G_S_AREA,             /* pi r^2                  (wire.py:170)              */
    WEDM_G_JOULE_GEOM,         /* dy / S                  (wire.py:183)              */
    WEDM_GEOM_F64_COUNT
};
enum wedm_geom_i32_field {
    WEDM_GI_N_SEG = 0,         /* wire.py:149      */
    WEDM_GI_ZONE_START,        /* wire.py:151,156  (plasma index base)  */
    WEDM_GI_AZ_START,          /* wire.py:207      (h_eff zone start)   */
    WEDM_GI_AZ_END,            /* wire.py:208      (h_eff zone end)     */
    WEDM_GI_CONTACT_BOTTOM,    /* wire.py:239-248  */
    WEDM_GI_CONTACT_TOP,       /* wire.py:242-251  */
    WEDM_GEOM_I32_COUNT
};

/* ----------------------------------------------------------------- params
 * Everything that is constant during a run.  Derived constants are computed by
 * the HOST in the reference's own float arithmetic (Python floats) and passed in
 * verbatim: the device never re-derives them (SURVEY.md §7 "floor-division
 * geometry quirks").                                                        */
#define WEDM_MAX_MODE 19

typedef struct wedm_params {
    /* EnvironmentConfig (core/env_config.py:17-35) */
    int32_t servo_interval;       /* [us] */
    int32_t dt_us;                /* [us] */
    int32_t control_mode;         /* 0 = position, 1 = velocity (mechanics.py:62-67) */
    int32_t per_env_geometry;     /* 0 = uniform values below, 1 = geometry rows bound */
    double initial_gap;
    double target_cutting_distance;

    /* uniform geometry (WireModule.__init__, wire.py:143-257) */
    int32_t n_seg, zone_start, az_start, az_end, contact_bottom, contact_top;
    double workpiece_height, kerf_base, cavity_coeff;
    double k_cond, tuf, a_surf, s_area, joule_geom;
    double segment_len;           /* [mm] for plasma_idx = zone_start + int(y // segment_len) */

    /* wire material + WireModuleParameters */
    double spool_T, temp_ref, rho_elec, alpha_rho, rho_c;   /* rho_c = density * specific_heat */
    double plasma_efficiency, base_convection, convection_velocity_factor, convection_flow_enhancement;
    double critical_temperature, breaking_temperature;
    double dielectric_temperature;

    /* IgnitionModuleParameters (ignition.py:17-57) */
    double base_critical_density, gap_coefficient, max_critical_density, hard_short_gap;
    double sigmoid_steepness;
    int32_t debris_short_duration, random_short_duration;
    double random_short_min_gap, random_short_max_gap, random_short_max_probability;
    double ignition_a, ignition_b, ignition_c, ln2;
    double default_target_voltage, default_on_time, default_off_time, default_current;
    double spark_voltage_factor;

    /* DielectricModuleParameters (dielectric.py:15-31) */
    double reference_gap, debris_obstruction_coeff, debris_removal_per_us;

    /* MechanicsModuleParameters (mechanics.py:12-23), pre-multiplied as in mechanics.py:48-57 */
    double dt_s, damping_coeff, stiffness_coeff, omega_n;
    double max_acceleration, max_jerk_dt, max_speed;

    /* tables: index n = mode "I<n>", n in 1..19; index 0 unused.
     * mode_current: currents.json.  crater_*: area_corrected.json
     * (ellipsoid_volume_half, ellipsoid_volume_std, depth); crater_valid[n] = 0
     * where the reference raises ValueError (material.py:108-113).           */
    double mode_current[WEDM_MAX_MODE + 1];
    double crater_mean[WEDM_MAX_MODE + 1];
    double crater_std[WEDM_MAX_MODE + 1];
    double crater_depth[WEDM_MAX_MODE + 1];
    int32_t crater_valid[WEDM_MAX_MODE + 1];

    /* sharding: global id of local environment 0 (Philox counter word 2) */
    uint32_t env_id_offset;
    int32_t obs_dim;              /* columns of the obs matrix written at control steps (0 = none) */
    int32_t disable_ignition;     /* 1: skip IgnitionModule.update — the monkeypatch of
                                     experiments/single_spark_animation.py:218-223 (spark forced by the caller) */
    /* Next-step autoreset inside the launch (the termination of wire_edm.py:172-179 handled without the
     * host): 1 = an environment that is DONE when a wedm_step begins is re-initialised by that launch
     * exactly as wedm_reset(mask = its DONE flag, reseed = 0) would (episode + 1, module state cleared,
     * wire at the spool temperature, statistics and observation zeroed) and then steps on.  Its DONE /
     * terminal flags therefore stay readable between two launches.                                   */
    int32_t autoreset;
    /* 0 = the reward row is never written (the reference's `_calculate_reward` is a TODO returning 0.0,
     * wire_edm.py:185-187); 1 = every wedm_step writes, for the environments it stepped, the float32
     * reward of that launch: (workpiece_position after - workpiece_position at the start of the launch
     * [after an autoreset]) - reward_break_penalty * is_wire_broken.                                  */
    int32_t reward_mode;
    /* typing of the wire stencil (wire.py:58-123): 0 = float32 op for op (what the reference computes when
     * NumPy-2 scalar promotion evaluates it, i.e. without Numba); 1 = float64 expressions rounded at each
     * float32 store (how Numba types the same lines; without fastmath re-association).                 */
    int32_t stencil_mode;
    /* what wedm_reset (and the in-launch autoreset) re-initialises: 0 = everything, module-private state included (a
     * fresh environment: the default, documented deviation); 1 = what the reference's reset() does (wire_edm.py:106-114):
     * a new EDMState only -- the module objects live on, so the ignition short timers and current cache
     * (ignition.py:75-81), the debris volume and the flow / density caches (dielectric.py:69-80), `prev_accel`
     * (mechanics.py:60), the convection cache and coefficients (wire.py:205,224) and the crater list / statistics
     * (material.py:133) carry over into the next episode.                                                        */
    int32_t reset_semantics;
    /* 0 = a terminated environment is frozen until it is reset (the default, documented deviation); 1 = it keeps being
     * stepped as the reference does when step() is called after `terminated` (wire_edm.py:116-157 has no guard): after a
     * wire break the wire module returns at once (wire.py:260-261) and the step returns before mechanics and clocks
     * (wire_edm.py:129-130); after the cutting target everything goes on.  WEDM_B_DONE then holds `terminated` of the
     * last step (is_wire_broken or is_target_distance_reached) and freezes nothing.                              */
    int32_t keep_stepping_terminated;
    double reward_break_penalty;  /* reward_mode 1 */
} wedm_params;

typedef struct wedm_state_ptrs {
    double* f64;        /* [WEDM_F64_COUNT][stride] */
    int32_t* i32;       /* [WEDM_I32_COUNT][stride] */
    int8_t* i8;         /* [WEDM_I8_COUNT ][stride] */
    float* T;           /* [WEDM_T_QUADS(n_seg_max)][stride][4], see WEDM_T_INDEX */
    float* obs;         /* [obs_dim][stride] or NULL */
    int64_t stride;     /* >= num_envs, multiple of 64 recommended */
    double* stats;      /* [WEDM_STAT_COUNT][stride] or NULL (statistics not kept) */
    float* reward;      /* [stride] or NULL; written when wedm_params.reward_mode != 0 */
    /* `MaterialRemovalModule.crater_volumes_um3` (material.py:133): every sampled crater volume [um^3] of an
     * environment, in spark order — crater number k (0-based since the reset) lands in
     * crater_log[(k % crater_log_capacity) * stride + env]; WEDM_I_SPARK_COUNT says how many there are.  NULL /
     * capacity 0: not kept.                                                                              */
    double* crater_log; /* [crater_log_capacity][stride] or NULL */
    int64_t crater_log_capacity;
} wedm_state_ptrs;

typedef struct wedm_geom_ptrs {
    const double* f64;  /* [WEDM_GEOM_F64_COUNT][stride] */
    const int32_t* i32; /* [WEDM_GEOM_I32_COUNT][stride] */
} wedm_geom_ptrs;

/* Action leaves of WireEDMEnv.action_space (wire_edm.py:84-98), one value per env.
 * float64 because `_apply_action` keeps the caller's precision (`float(x[0])`,
 * wire_edm.py:162-170).                                                        */
typedef struct wedm_action_ptrs {
    const double* servo;           /* -> state.target_delta   */
    const double* target_voltage;  /* -> state.target_voltage */
    const double* on_time;         /* -> state.ON_time        */
    const double* off_time;        /* -> state.OFF_time       */
    const int32_t* current_mode;   /* -> state.current_mode "I<n>" */
} wedm_action_ptrs;

/* Device-side signal trace: while wedm_step runs, the kernels copy the selected rows of the
 * state blocks (and optionally the whole wire temperature) of a contiguous range of
 * environments into caller-owned ring buffers every `every` microseconds.  Replaces the
 * per-microsecond `SimulationLogger.collect(env.state, info)` of the reference driver
 * (utils/logger.py:110-160, experiments/run_simulation.py:257) and its 1 ms voltage history
 * (run_simulation.py:262-270) without leaving the fused launch.  Sample m (1-based count of
 * microseconds stepped since wedm_bind_trace) is taken after the step when m % every == 0 and
 * lands in ring slot (m / every - 1) % capacity.  Rows are packed in ascending row order.   */
typedef struct wedm_trace_desc {
    double* f64;        /* [capacity][popcount(f64_mask)][env_count], NULL iff f64_mask == 0 */
    int32_t* i32;       /* [capacity][popcount(i32_mask)][env_count], NULL iff i32_mask == 0 */
    int8_t* i8;         /* [capacity][popcount(i8_mask) ][env_count], NULL iff i8_mask  == 0 */
    float* T;           /* [capacity][n_seg_max][env_count] or NULL (no temperature trace)   */
    uint32_t f64_mask;  /* bit r: record row r of wedm_f64_field */
    uint32_t i32_mask;  /* bit r: record row r of wedm_i32_field */
    uint32_t i8_mask;   /* bit r: record row r of wedm_i8_field  */
    int32_t env_lo;     /* first traced environment (local index) */
    int32_t env_count;  /* number of traced environments          */
    int32_t every;      /* sample period in microseconds, >= 1    */
    int32_t capacity;   /* ring capacity in samples, >= 1         */
    int32_t reserved0;
} wedm_trace_desc;

/* Variate injection (validation mode).  Instead of its Philox stream an environment consumes caller-provided
 * variates: what the reference drew from its own NumPy Generator(PCG64) at the call sites ignition.py:233,239,
 * 261,327 and material.py:127 (`env.np_random`, seeded by reset(seed), wire_edm.py:55,107), laid out by physics
 * step since the reset and by slot,  table[(step * WEDM_REPLAY_SLOTS + slot) * stride + env],  NaN where the
 * reference drew nothing in that step.  With it the device follows a native-seed run of the reference
 * (fixture F1) directly.  Runs on the global-memory kernel only.                                       */
enum wedm_replay_slot {
    WEDM_RS_DEBRIS_ROLL = 0,   /* Generator.random() compared with p_debris   (ignition.py:233) */
    WEDM_RS_RANDOM_ROLL,       /* Generator.random() compared with p_random   (ignition.py:239) */
    WEDM_RS_IGNITION_ROLL,     /* Generator.random() compared with lambda     (ignition.py:327) */
    WEDM_RS_SPARK_Y,           /* Generator.uniform(0, workpiece_height)      (ignition.py:261) */
    WEDM_RS_CRATER_UM3,        /* Generator.normal(mean, std) [um^3]          (material.py:127) */
    WEDM_REPLAY_SLOTS
};

typedef struct wedm_ctx wedm_ctx;

/* version of this header the library was built against */
int32_t wedm_abi_version(void);

/* fingerprint of the build: the first 16 hex digits of the sha256 over the kernel sources (wedm_kernels.hip,
 * wedm_device.h, this header) and the compiler flags, baked in by the build (__graft_entry__.build_hip).  Measurement
 * records (profiles/valu.json, profiles/traffic.json) carry the id of the library they were counted on, and bench.py
 * prices a live duration with a recorded instruction / byte count only when the ids agree.  "unknown" for a library
 * built by hand without -DWEDM_BUILD_ID.                                                                        */
const char* wedm_build_id(void);

/* replaces WireEDMEnv.__init__ (wire_edm.py:22-101): validates sizes, copies
 * `params`, selects the device that is current at call time.                  */
int32_t wedm_create(const wedm_params* params, int32_t num_envs, int32_t n_seg_max, wedm_ctx** out);
int32_t wedm_destroy(wedm_ctx* ctx);

/* replaces the EDMState() allocation (wire_edm.py:58,110): adopts caller-owned memory */
int32_t wedm_bind_state(wedm_ctx* ctx, const wedm_state_ptrs* state);
int32_t wedm_bind_geometry(wedm_ctx* ctx, const wedm_geom_ptrs* geom);

/* replaces WireEDMEnv.reset (wire_edm.py:106-114) for the environments whose
 * mask byte is non-zero (mask == NULL: all).  `reseed` bit 0: re-key their RNG with
 * `seed` (else their episode counter is bumped).  Module-private state is reset too
 * unless wedm_params.reset_semantics is 1 (the reference's own reset); `reseed` bit 1
 * (WEDM_RESET_FRESH) resets it whatever the semantics -- what constructing the module
 * objects does in WireEDMEnv.__init__ (wire_edm.py:60-82): a handle's first reset. */
#define WEDM_RESET_RESEED 1
#define WEDM_RESET_FRESH 2
int32_t wedm_reset(wedm_ctx* ctx, const uint8_t* mask, uint64_t seed, int32_t reseed, void* stream);

/* replaces `n_substeps` consecutive WireEDMEnv.step(action) calls
 * (wire_edm.py:116-157) with the same action.  n_substeps == 1 is the
 * reference's 1 us step.                                                      */
int32_t wedm_step(wedm_ctx* ctx, int32_t n_substeps, const wedm_action_ptrs* action, void* stream);

/* binds (table != NULL) or removes (table == NULL) the variate table described at wedm_replay_slot; `n_steps`
 * physics steps are covered (an environment stepped beyond them gets its ERROR flag set).                 */
int32_t wedm_bind_rng_replay(wedm_ctx* ctx, const double* table, int64_t n_steps);

/* binds (desc != NULL) or removes (desc == NULL) the signal trace; resets the sample counter.
 * Terminated environments keep being sampled (their frozen state).                        */
int32_t wedm_bind_trace(wedm_ctx* ctx, const wedm_trace_desc* desc);

/* samples written since wedm_bind_trace (host-side count; the ring holds the last
 * min(count, capacity) of them, the newest in slot (count - 1) % capacity)                */
int64_t wedm_trace_samples(wedm_ctx* ctx);

/* selects the kernel used by wedm_step: 0 = auto, 1 = global-memory stencil (one pass
 * over T in HBM per substep), 2 = LDS-staged predicated stencil (any geometry),
 * 3 = LDS-staged fused stencil walking a wave-uniform tile table (uniform geometry),
 * 4 = the same with two chunks per lane advanced by packed float32 math,
 * 5 = global-memory stencil with the wire split over the four waves of a block (single
 * microseconds, any geometry), 6 = stream kernel (single microseconds, uniform geometry: the whole
 * chunk of a lane requested up front, tile walk in LDS, no barrier; the automatic choice for
 * n_substeps == 1 where one round of blocks covers the batch), 7 = register kernel (one environment per
 * lane with its whole wire in registers: no LDS, the scalar physics once per environment; uniform geometry,
 * at most 128 segments, either typing of the stencil; own instantiation for launches with a trace sample),
 * 8 = wide register kernel (4, 8 or 16 lanes per environment -- the fewest that hold the wire at 32 cells per lane --
 * with the wire in their registers and per-cell zone / contact coefficients: no LDS, no tile table; uniform geometry,
 * 9 to 512 segments, either typing of the stencil; the automatic choice for fused launches of a batch that one round of blocks
 * covers: environments x lanes <= 65 536; own instantiations for wires whose length is not a multiple of 8 and for
 * launches with a trace sample), 9 = served kernel (kernel 4's walk -- 4 or 8 lanes per environment, the wire in LDS --
 * with the float64 scalar physics of a block's environments on a FIFTH wave of the block, one lane per environment, one
 * microsecond ahead of the four walking waves where it can prove that the step does not break the wire; coefficients and
 * maxima cross through LDS; uniform geometry, float32 stencil, freeze-on-termination; launches with a trace sample take
 * kernel 4), 10 = kernel 2's cell-by-cell form by name (since round 4 kernel 2 itself is the packed form -- two virtual
 * chunks per lane advanced in float2 registers, per-cell coefficients from the lane's own indices -- wherever the stencil is
 * float32, and the same walk with float64-typed cells under stencil_mode 1; the cell-by-cell form remains for A/B timing), 11 = the served form of kernel 2 (its walk
 * on three waves of a block, the scalar physics on the fourth, as kernel 9; by name only: not faster at BASELINE configs[4]),
 * 12 = the served form of kernel 7 (two walker waves with the wire in registers + a scalar wave with all 64 lanes busy; by name
 * only: 1.666e10 against kernel 7's 1.674e10 at the headline batch, spills; at most 128 segments, no trace sample).
 * All variants produce bit-identical results.  With wedm_params.stencil_mode 1 only 0, 1, 2 (= 10), 3, 6 (launches of one microsecond
 * without a trace sample, chunks of at most 64 cells), 7 and 8 are accepted (7 / 8: the
 * register walks with every interior cell in Numba's typing -- 18 float64 operations per cell -- and, for kernel 8, a
 * 256-register instantiation at two blocks per CU for batches beyond one wave per SIMD; 3: the tile walk with per-cell
 * coefficients; no packed, served or single-microsecond form -- 0 takes kernel 7 from 20 480 environments of at most 128
 * segments, kernel 8 for other wires of 9 to 512 segments, else 3 / 2 / 1).                       */
int32_t wedm_set_kernel(wedm_ctx* ctx, int32_t variant);

/* lanes that share one environment in kernels 2, 3, 4 and 6: 0 = auto, or 1, 2, 4, 8 (16: not kernel 4); kernel 7: 1, anything
 * else = 2; kernel 8: 0 = the fewest, or 4, 8, 16 if 32 cells per lane cover the wire.  A lane count set here also keeps
 * the automatic choice (kernel 0) off the register kernels.                                                          */
int32_t wedm_set_lanes(wedm_ctx* ctx, int32_t lanes);

/* name / launch geometry of the kernel the last wedm_step used (for profiles) */
const char* wedm_last_kernel(wedm_ctx* ctx);

const char* wedm_last_error(wedm_ctx* ctx);

/* sizeof(wedm_params) the library was compiled with (layout cross-check for bindings) */
int64_t wedm_sizeof_params(void);

/* Blocks of the last launch's kernel that the occupancy API admits per compute unit at that launch's block size and
 * dynamic LDS (hipOccupancyMaxActiveBlocksPerMultiprocessor), or a negative status.  Diagnostic: measurement scripts
 * record it next to a kernel's name; nothing in the library depends on it.                                          */
int32_t wedm_last_occupancy(wedm_ctx* ctx);

/* TEST HOOK: evaluates one of the device math primitives the physics relies on,
 * element-wise on device arrays, so tests can compare them bit for bit with the CPU.
 * kind: 0 exp, 1 log, 2 correctly-rounded cube, 3 sqrt, 4 Python floor-division
 * a // b, 5 a / b, 6 the four Philox step uniforms (u0 + 2 u1 + 4 u2 + 8 u3) at (time=a, env=b),
 * 7 Philox polar normal at (time=a, env=b), 8 the spark's cell offset int(a // b) as the kernels
 * compute it (one division + a fused remainder where a >= 0 and b > 0).  Key 0x9abcdef012345678,
 * episode 3.                                                                               */
int32_t wedm_debug_math(int32_t kind, const double* a, const double* b, double* out, int32_t n, void* stream);

/* TEST HOOK: fills the LDS of every compute unit of the current device with `value`.  The step kernels stage only the
 * cells of a wire that exist; rows of their LDS image past a wire's end are never written, and a result that depended
 * on one would go unnoticed as long as the previous kernel happened to leave plausible temperatures there.  The GPU
 * tests poison the LDS (1e30) before every test.                                                                  */
int32_t wedm_debug_poison_lds(float value, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WEDM_HIP_H */
