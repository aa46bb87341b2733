/*
 * wedm_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the per-microsecond step of geduardo/SPARC's
 * `wedm.WireEDMEnv` (reference: /root/reference/src/wedm, pure Python).  It is the
 * checker the HIP path is compared against.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load it; the product (sparc_amd/) never
 * does, and has no CPU path of its own.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py replays fixtures generated
 * by running the reference itself in the build container (tools/gen_golden.py,
 * stub-imported as SURVEY.md §8c describes) and requires bit-equality of every
 * recorded quantity (discrete state, float64 scalars, float32 temperatures) in
 * math mode WEDM_ORACLE_MATH_LIBM.
 */
#ifndef WEDM_ORACLE_H
#define WEDM_ORACLE_H

#include <stdint.h>

#include "../include/wedm_hip.h" /* SoA layout + wedm_params for the batch driver */

#ifdef __cplusplus
extern "C" {
#endif

#define WEDM_ORACLE_MAX_SEG 4096

/* How pow/exp/log are evaluated.
 *  LIBM     : glibc pow()/exp()/log() — what CPython/NumPy call in the reference
 *             (`gap**2`, `(g/g_ref)**3`, np.exp, np.log).  Used to pin the oracle
 *             against the reference fixtures.
 *  PORTABLE : gap*gap, a correctly-rounded cube and the table-free exp/log below,
 *             all built from IEEE-754 basic operations only, so that a GPU can
 *             reproduce them bit for bit.  glibc pow() is NOT correctly rounded
 *             (≈0.09 % of x**2 differ from x*x by 1 ulp, measured here), so the
 *             two modes can differ by 1 ulp in lambda and in the flow factor.  */
enum { WEDM_ORACLE_MATH_LIBM = 0, WEDM_ORACLE_MATH_PORTABLE = 1 };

/* How the f32 stencil expressions are typed.
 *  F32 : every operation in float32 — what the reference computes when run with
 *        the Numba stub under NumPy-2 promotion rules (the only way it runs here).
 *  F64 : float64 expression, rounded at each float32 store — how real Numba types
 *        wire.py:58-123 (without fastmath re-association).  Reported, not pinned. */
enum { WEDM_ORACLE_STENCIL_F32 = 0, WEDM_ORACLE_STENCIL_F64 = 1 };

enum { WEDM_ORACLE_RNG_PHILOX = 0, WEDM_ORACLE_RNG_REPLAY = 1 };

/* Raw (un-derived) configuration: EnvironmentConfig + the five *ModuleParameters
 * + the brass row of data/wire_materials.json.                                  */
typedef struct wedm_oracle_config {
    /* EnvironmentConfig, core/env_config.py:17-35 */
    double workpiece_height, wire_diameter;
    int32_t dt, servo_interval;
    double initial_gap, target_cutting_distance;
    /* WireMaterial, core/material_db.py:10-21 */
    double density, specific_heat, thermal_conductivity, electrical_resistivity;
    double temperature_coefficient, melting_point, breaking_temperature;
    /* IgnitionModuleParameters, modules/ignition.py:17-57 */
    double base_critical_density, gap_coefficient, max_critical_density, hard_short_gap;
    double sigmoid_steepness;
    int32_t debris_short_duration, random_short_duration;
    double random_short_min_gap, random_short_max_gap, random_short_max_probability;
    double ignition_a_coeff, ignition_b_coeff, ignition_c_coeff;
    double default_target_voltage, default_on_time, default_off_time;
    int32_t default_current_mode, pad0;
    double spark_voltage_factor;
    /* WireModuleParameters, modules/wire.py:16-54 */
    double buffer_len_bottom, buffer_len_top, segment_len, spool_T;
    double contact_offset_bottom, contact_offset_top;
    double base_convection_coefficient, plasma_efficiency;
    double convection_velocity_factor, convection_flow_enhancement;
    int32_t compute_zone_mean, zone_mean_interval;
    double critical_temp_threshold, wire_breaking_temp_factor;
    /* MaterialModuleParameters, modules/material.py:17-22 */
    double base_overcut;
    /* DielectricModuleParameters, modules/dielectric.py:15-31 */
    double base_flow_rate, debris_removal_efficiency, debris_obstruction_coeff;
    double reference_gap, dielectric_temperature;
    int32_t ion_channel_duration;
    /* MechanicsModuleParameters, modules/mechanics.py:12-23 */
    int32_t control_mode; /* 0 position, 1 velocity */
    double omega_n, zeta, max_acceleration, max_jerk, max_speed;
} wedm_oracle_config;

/* Constants the reference derives once in the module constructors. */
typedef struct wedm_oracle_consts {
    int32_t servo_interval, dt_us, control_mode;
    int32_t n_seg, zone_start, zone_end, az_start, az_end, contact_bottom, contact_top;
    double initial_gap, target_cutting_distance;
    double workpiece_height, kerf_base, cavity_coeff;
    double k_cond, tuf, a_surf, s_area, joule_geom, segment_len;
    double spool_T, temp_ref, rho_elec, alpha_rho, rho_c;
    double plasma_efficiency, base_convection, convection_velocity_factor, convection_flow_enhancement;
    double critical_temperature, breaking_temperature, dielectric_temperature;
    double base_critical_density, gap_coefficient, max_critical_density, hard_short_gap, sigmoid_steepness;
    int32_t debris_short_duration, random_short_duration;
    double random_short_min_gap, random_short_max_gap, random_short_max_probability;
    double ignition_a, ignition_b, ignition_c, ln2;
    double default_target_voltage, default_on_time, default_off_time, default_current;
    double spark_voltage_factor;
    double reference_gap, debris_obstruction_coeff, debris_removal_per_us;
    double dt_s, damping_coeff, stiffness_coeff, omega_n, max_acceleration, max_jerk_dt, max_speed;
    double mode_current[WEDM_MAX_MODE + 1];
    double crater_mean[WEDM_MAX_MODE + 1], crater_std[WEDM_MAX_MODE + 1], crater_depth[WEDM_MAX_MODE + 1];
    int32_t crater_valid[WEDM_MAX_MODE + 1];
} wedm_oracle_consts;

typedef struct wedm_oracle_rng {
    int32_t mode;            /* WEDM_ORACLE_RNG_* */
    int32_t draws_this_step; /* number of variates consumed by the last step */
    uint64_t seed;           /* Philox key */
    uint32_t env_id;         /* Philox counter word 2 (global environment id) */
    uint32_t episode;        /* Philox counter word 1 */
    const double* replay;    /* REPLAY: variates in the reference's draw order */
    int64_t replay_len, replay_pos;
} wedm_oracle_rng;

/* One environment: EDMState (core/state.py:25-91) + module-private state. */
typedef struct wedm_oracle_env {
    wedm_oracle_consts c;
    wedm_oracle_rng rng;
    int32_t math_mode, stencil_mode;
    int32_t disable_ignition; /* experiments/single_spark_animation.py:218-223 */
    int32_t error;            /* 1: fresh spark with a mode that has no crater data */
    /* time tracking */
    int32_t time, time_since_servo, time_since_open_voltage;
    int32_t time_since_spark_ignition, time_since_spark_end;
    /* electrical */
    double voltage, current;
    /* generator settings; 0 encodes None (the reference's `x or default`) */
    double target_voltage, on_time, off_time;
    int32_t current_mode; /* n for "I<n>", 0 = None */
    /* motion */
    double workpiece_position, wire_position, wire_velocity, wire_unwinding_velocity;
    /* thermal */
    int32_t time_in_critical_temp;
    /* spark_status = [state, y, duration]; y NaN = None */
    int32_t spark_state, spark_dur;
    double spark_y;
    /* dielectric */
    double dielectric_temperature, debris_volume, debris_density, cavity_volume, flow_rate;
    double last_crater_volume;
    /* flags */
    int32_t is_short_circuit, is_wire_broken, is_target_reached;
    /* servo */
    double target_delta, target_position;
    /* module-private */
    int32_t random_short_remaining, debris_short_remaining; /* ignition.py:75-76 */
    double diel_last_gap, diel_last_density;                /* dielectric.py:78-80 */
    double wire_last_flow;                                  /* wire.py:224 */
    float h_base, h_zone;                                   /* wire.py:205 (two distinct values) */
    double prev_accel;                                      /* mechanics.py:60 */
    /* the driver's 1 ms voltage history as a running sum (experiments/run_simulation.py:258-281;
     * rows WEDM_F_VOLT_ACC / WEDM_F_VOLT_SUM of include/wedm_hip.h) */
    double volt_acc, volt_sum;
    int32_t spark_count;                                    /* len(crater_volumes_um3) */
    /* running statistics of crater_volumes_um3 (material.py:207-227): sum, sum of squares, min, max */
    double crater_stat_sum, crater_stat_sumsq, crater_stat_min, crater_stat_max;
    /* crater_volumes_um3 (material.py:133), batch driver only: where this environment's list lives (SoA column) */
    double* crater_log;
    int64_t crater_log_capacity, crater_log_stride;
    float tmax;
    /* step() outputs */
    int32_t last_terminated, last_ctrl_step, last_early_return;
    /* IgnitionModule._cached_current_mode is not None (ignition.py:79-81,98-113): a peak current has been looked up for a
     * latched mode; survives wedm_oracle_reset_reference() */
    int32_t mode_cached;
    float T[WEDM_ORACLE_MAX_SEG];
    float dT[WEDM_ORACLE_MAX_SEG];
} wedm_oracle_env;

typedef struct wedm_oracle_action {
    double servo, target_voltage, on_time, off_time;
    int32_t current_mode;
} wedm_oracle_action;

void wedm_oracle_default_config(wedm_oracle_config* cfg);
/* WireEDMEnv.__init__ + module constructors: derive constants from the raw config */
int32_t wedm_oracle_derive(const wedm_oracle_config* cfg, wedm_oracle_consts* out);
/* fresh environment followed by one reset(): wire_edm.py:22-114 */
int32_t wedm_oracle_init(wedm_oracle_env* env, const wedm_oracle_config* cfg);
void wedm_oracle_reset(wedm_oracle_env* env);
/* WireEDMEnv.reset as the reference does it on a USED environment (wire_edm.py:106-114): a new EDMState only; what the
 * module objects hold lives on (short timers, current cache, debris volume, flow / density / convection caches,
 * prev_accel, crater list and statistics) */
void wedm_oracle_reset_reference(wedm_oracle_env* env);
/* WireEDMEnv.step: wire_edm.py:116-157.  Returns terminated (0/1). */
int32_t wedm_oracle_step(wedm_oracle_env* env, const wedm_oracle_action* action);

/* RNG pieces, exported for known-answer tests and for the fixture generator */
void wedm_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void wedm_oracle_step_uniforms(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t time, double out[4]);
void wedm_oracle_uniform_pair(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t time,
                              uint32_t stream, double out[2]);
double wedm_oracle_std_normal(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t time,
                              int32_t* n_pairs);
double wedm_oracle_exp(double x, int32_t math_mode);
double wedm_oracle_log(double x, int32_t math_mode);
double wedm_oracle_cube(double x, int32_t math_mode);
double wedm_oracle_py_floordiv(double vx, double wx);

/* Batch driver on the C-ABI's struct-of-arrays layout (HOST pointers).  Same
 * semantics as wedm_step()/wedm_reset() in include/wedm_hip.h; OpenMP over
 * environments with `n_threads` threads (<= 0: all cores).                    */
int32_t wedm_oracle_reset_batch(const wedm_params* p, const wedm_state_ptrs* s, int32_t num_envs,
                                const uint8_t* mask, uint64_t seed, int32_t reseed);
int32_t wedm_oracle_step_batch(const wedm_params* p, const wedm_state_ptrs* s, const wedm_geom_ptrs* g,
                               const wedm_action_ptrs* a, int32_t num_envs, int32_t n_seg_max, int32_t n_substeps,
                               int32_t math_mode, int32_t stencil_mode, int32_t n_threads);
int32_t wedm_oracle_max_threads(void);
int64_t wedm_oracle_sizeof(int32_t which);

#ifdef __cplusplus
}
#endif
#endif
