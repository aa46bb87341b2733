/*
 * wedm_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see wedm_oracle.h).
 *
 * Plain-C restatement of the reference's per-microsecond step.  Each function
 * cites the reference lines it follows (paths relative to /root/reference/).
 * Arithmetic follows CPython/NumPy evaluation order exactly: Python floats are
 * IEEE doubles evaluated left to right with no fused multiply-add, so this file
 * MUST be compiled with -ffp-contract=off (the Makefile does).
 *
 * Third-party arithmetic on the path (not in the reference tree):
 *   - NumPy Generator(PCG64) `.random() / .uniform() / .normal()` (numpy>=1.21,
 *     unpinned in pyproject.toml:34; 2.2.6 installed here).  Not restated: the
 *     reference reaches its RNG through the pluggable `env.np_random` attribute, so
 *     parity is anchored on REPLAY of the variates the reference actually drew
 *     (fixtures F1,F5..F7) and on INJECTION of this file's Philox variates into the
 *     reference (fixture F3).
 *   - glibc pow/exp/log through CPython `**` and np.exp/np.log: math mode LIBM.
 */
#include "wedm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ tables */
/* modules/currents.json: machine current [A] of mode "I<n>" */
static const double k_mode_current[WEDM_MAX_MODE + 1] = {
    0.0, 30, 35, 40, 50, 60, 68, 80, 95, 110, 130, 155, 180, 215, 255, 305, 360, 425, 500, 600};

/* modules/area_corrected.json: {ellipsoid_volume_half, ellipsoid_volume_std, depth}
 * exists for the odd modes I1..I17 only. */
static const struct { int mode; double mean, std, depth; } k_crater[] = {
    {1, 2163.6174, 447.5072, 2.9967},   {3, 2376.5835, 523.6196, 3.0443},
    {5, 4866.9691, 899.0243, 3.5302},   {7, 5556.8153, 1167.296, 3.6478},
    {9, 6219.2736, 1284.4152, 3.7556},  {11, 6029.8571, 1005.804, 3.7252},
    {13, 8913.8402, 2949.127, 4.1537},  {15, 26468.9924, 6472.3303, 5.9966},
    {17, 59549.9184, 8997.5034, 6.2647}};

/* ------------------------------------------------------- Python semantics */
/* CPython float_divmod (Objects/floatobject.c), the `//` of two floats.
 * wire.py:151-154 relies on its quirks: int(30.0 // 0.2) == 149. */
double wedm_oracle_py_floordiv(double vx, double wx) {
    double mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod != 0.0) {
        if ((wx < 0) != (mod < 0)) {
            mod += wx;
            div -= 1.0;
        }
    }
    double floordiv;
    if (div != 0.0) {
        floordiv = floor(div);
        if (div - floordiv > 0.5) floordiv += 1.0;
    } else {
        floordiv = copysign(0.0, vx / wx);
    }
    return floordiv;
}

static int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

/* --------------------------------------------------------- portable math
 * exp/log from IEEE basic operations only (argument reduction by ln2 hi/lo and
 * a degree-5 / degree-7 minimax polynomial, the classic Sun fdlibm scheme), so a
 * GPU evaluating the same expression tree gets the same bits.                 */
static double bits_to_double(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static uint64_t double_to_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

static double portable_exp(double x) {
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03;
    const double P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06;
    const double P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.0) return INFINITY;
    if (x < -708.0) return 0.0;
    double hi = x, lo = 0.0;
    int k = 0;
    double ax = fabs(x);
    if (ax > 0.34657359027997264) { /* 0.5 ln2 */
        k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
        hi = x - (double)k * ln2hi;
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (ax < 3.725290298461914e-09) { /* 2^-28 */
        return 1.0 + x;
    }
    double t = x * x;
    double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    return y * bits_to_double((uint64_t)(k + 1023) << 52);
}

/* x must be positive, finite and normal (the only use: polar-method s in (0,1)) */
static double portable_log(double x) {
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01;
    const double Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01;
    const double Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01;
    const double Lg7 = 1.479819860511658591e-01;
    uint64_t ux = double_to_bits(x);
    uint32_t hx = (uint32_t)(ux >> 32);
    int k = (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    uint32_t i = (hx + 0x95f64u) & 0x100000u; /* mantissa >= sqrt(2): halve it */
    ux = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffu);
    k += (int)(i >> 20);
    double f = bits_to_double(ux) - 1.0;
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

/* LIBM mode = glibc exp.  NOTE: the reference calls np.exp, which NumPy 2.2.6 evaluates with its
 * own AVX512F kernel on this host: 1 ulp away from glibc in 4.6 % of arguments (measured).  Only
 * fast_exp for k*rho >= 0.5 (dielectric.py:34-41) carries the value into the state; every committed
 * fixture is still reproduced bit for bit, random scenarios agree to ~1e-16 there (DESIGN.md par. 3). */
double wedm_oracle_exp(double x, int32_t math_mode) {
    return math_mode == WEDM_ORACLE_MATH_LIBM ? exp(x) : portable_exp(x);
}
double wedm_oracle_log(double x, int32_t math_mode) {
    return math_mode == WEDM_ORACLE_MATH_LIBM ? log(x) : portable_log(x);
}
/* `x ** 3` on Python floats is libm pow(x, 3.0) */
double wedm_oracle_cube(double x, int32_t math_mode) {
    if (math_mode == WEDM_ORACLE_MATH_LIBM) return pow(x, 3.0);
    double p = x * x, e = fma(x, x, -p);
    double q = p * x, eq = fma(p, x, -q);
    return q + (eq + e * x);
}
static double oracle_square(double x, int32_t math_mode) {
    return math_mode == WEDM_ORACLE_MATH_LIBM ? pow(x, 2.0) : x * x;
}

/* ------------------------------------------------------------------- RNG
 * Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
 * SC'11).  Counter = {time, episode, global env id, stream}, key = 64-bit seed.
 *   stream 0    : ONE call per microsecond gives the four uniforms a step can need:
 *                 word 0 debris-short roll, word 1 random-short roll, word 2
 *                 ignition roll, word 3 spark location.  u = (w + 0.5) * 2^-32,
 *                 strictly inside (0,1): a probability below 2^-33 never fires.
 *   stream 1+j  : pair j of the polar method for the crater normal, 53-bit doubles
 *                 built like NumPy's 32-bit bit generators:
 *                 ((a >> 5) * 2^26 + (b >> 6)) / 2^53.                          */
void wedm_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void philox_words(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t time, uint32_t stream,
                         uint32_t w[4]) {
    uint32_t ctr[4] = {time, episode, env_id, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    wedm_oracle_philox4x32_10(ctr, key, w);
}

/* the four per-step uniforms (stream 0) */
void wedm_oracle_step_uniforms(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t time, double out[4]) {
    uint32_t w[4];
    philox_words(seed, env_id, episode, time, 0u, w);
    for (int i = 0; i < 4; ++i) out[i] = ((double)w[i] + 0.5) * 2.3283064365386963e-10; /* 2^-32 */
}

void wedm_oracle_uniform_pair(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t time,
                              uint32_t stream, double out[2]) {
    uint32_t w[4];
    philox_words(seed, env_id, episode, time, stream, w);
    out[0] = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) / 9007199254740992.0;
    out[1] = ((double)(w[2] >> 5) * 67108864.0 + (double)(w[3] >> 6)) / 9007199254740992.0;
}

/* Marsaglia polar method: needs only +,*,/,sqrt (all correctly rounded) and log. */
double wedm_oracle_std_normal(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t time,
                              int32_t* n_pairs) {
    for (uint32_t j = 0; j < 64; ++j) {
        double u[2];
        wedm_oracle_uniform_pair(seed, env_id, episode, time, 1u + j, u);
        double v1 = 2.0 * u[0] - 1.0, v2 = 2.0 * u[1] - 1.0;
        double s = v1 * v1 + v2 * v2;
        if (s < 1.0 && s != 0.0) {
            if (n_pairs) *n_pairs = (int32_t)j + 1;
            return v1 * sqrt(-2.0 * portable_log(s) / s); /* always the portable log: this is the build's RNG */
        }
    }
    if (n_pairs) *n_pairs = 64;
    return 0.0; /* probability (1 - pi/4)^64 ~ 1e-43 */
}

static double rng_replay_next(wedm_oracle_env* env) {
    wedm_oracle_rng* r = &env->rng;
    r->draws_this_step++;
    if (r->replay_pos >= r->replay_len) {
        env->error |= 2; /* trace exhausted */
        return 0.5;
    }
    return r->replay[r->replay_pos++];
}

/* env.np_random.random(): slot 0 debris roll (ignition.py:233), 1 random-short roll
 * (ignition.py:239), 2 ignition roll (ignition.py:327) */
static double rng_random(wedm_oracle_env* env, int slot) {
    if (env->rng.mode == WEDM_ORACLE_RNG_REPLAY) return rng_replay_next(env);
    env->rng.draws_this_step++;
    double u[4];
    wedm_oracle_step_uniforms(env->rng.seed, env->rng.env_id, env->rng.episode, (uint32_t)env->time, u);
    return u[slot];
}
/* env.np_random.uniform(0, workpiece_height), ignition.py:261-263.
 * NumPy: low + (high - low) * next_double. */
static double rng_uniform_y(wedm_oracle_env* env) {
    if (env->rng.mode == WEDM_ORACLE_RNG_REPLAY) return rng_replay_next(env);
    env->rng.draws_this_step++;
    double u[4];
    wedm_oracle_step_uniforms(env->rng.seed, env->rng.env_id, env->rng.episode, (uint32_t)env->time, u);
    return 0.0 + (env->c.workpiece_height - 0.0) * u[3];
}
/* env.np_random.normal(mean, std), material.py:127.  NumPy: loc + scale * z. */
static double rng_normal(wedm_oracle_env* env, double mean, double std) {
    if (env->rng.mode == WEDM_ORACLE_RNG_REPLAY) return rng_replay_next(env);
    env->rng.draws_this_step++;
    double z = wedm_oracle_std_normal(env->rng.seed, env->rng.env_id, env->rng.episode,
                                      (uint32_t)env->time, NULL);
    return mean + std * z;
}

/* ---------------------------------------------------------- construction */
void wedm_oracle_default_config(wedm_oracle_config* c) {
    memset(c, 0, sizeof(*c));
    /* core/env_config.py:17-35 */
    c->workpiece_height = 20.0; c->wire_diameter = 0.2; c->dt = 1; c->servo_interval = 1000;
    c->initial_gap = 50.0; c->target_cutting_distance = 500.0;
    /* data/wire_materials.json "brass" */
    c->density = 8400; c->specific_heat = 377; c->thermal_conductivity = 120;
    c->electrical_resistivity = 6.4e-8; c->temperature_coefficient = 0.0039;
    c->melting_point = 1173; c->breaking_temperature = 1500;
    /* modules/ignition.py:17-57 */
    c->base_critical_density = 0.3; c->gap_coefficient = 0.02; c->max_critical_density = 0.95;
    c->hard_short_gap = 2.0; c->sigmoid_steepness = 500.0;
    c->debris_short_duration = 50; c->random_short_duration = 100;
    c->random_short_min_gap = 2.0; c->random_short_max_gap = 50.0; c->random_short_max_probability = 0.0;
    c->ignition_a_coeff = 0.48; c->ignition_b_coeff = -3.69; c->ignition_c_coeff = 14.05;
    c->default_target_voltage = 80.0; c->default_on_time = 3.0; c->default_off_time = 80.0;
    c->default_current_mode = 5; c->spark_voltage_factor = 0.3;
    /* modules/wire.py:16-54 */
    c->buffer_len_bottom = 30.0; c->buffer_len_top = 30.0; c->segment_len = 0.2; c->spool_T = 293.15;
    c->contact_offset_bottom = 10.0; c->contact_offset_top = 10.0;
    c->base_convection_coefficient = 14000; c->plasma_efficiency = 0.1;
    c->convection_velocity_factor = 0.5; c->convection_flow_enhancement = 1.0;
    c->compute_zone_mean = 0; c->zone_mean_interval = 100;
    c->critical_temp_threshold = 0.9; c->wire_breaking_temp_factor = 1.1;
    /* modules/material.py:17-22 */
    c->base_overcut = 0.12;
    /* modules/dielectric.py:15-31 */
    c->base_flow_rate = 100.0; c->debris_removal_efficiency = 0.01; c->debris_obstruction_coeff = 1.0;
    c->reference_gap = 25.0; c->dielectric_temperature = 293.15; c->ion_channel_duration = 6;
    /* modules/mechanics.py:12-23 */
    c->control_mode = 0; c->omega_n = 235.0; c->zeta = 0.38;
    c->max_acceleration = 3.0e5; c->max_jerk = 1.0e8; c->max_speed = 3.0e4;
}

int32_t wedm_oracle_derive(const wedm_oracle_config* cfg, wedm_oracle_consts* o) {
    memset(o, 0, sizeof(*o));
    /* core/env_config.py:73-90 validate() */
    if (!(cfg->workpiece_height > 0) || !(cfg->wire_diameter > 0) || !(cfg->initial_gap > 0) ||
        !(cfg->target_cutting_distance > 0) || cfg->dt <= 0 || cfg->servo_interval <= 0)
        return -1;
    o->servo_interval = cfg->servo_interval;
    o->dt_us = cfg->dt;
    o->control_mode = cfg->control_mode;
    o->initial_gap = cfg->initial_gap;
    o->target_cutting_distance = cfg->target_cutting_distance;
    o->workpiece_height = cfg->workpiece_height;

    /* ---- WireModule.__init__, wire.py:143-257 ---- */
    double seg = cfg->segment_len;
    double total_L = cfg->buffer_len_bottom + cfg->workpiece_height + cfg->buffer_len_top; /* :144-148 */
    int32_t n_seg = imax(1, (int32_t)(total_L / seg));                                    /* :149 */
    if (n_seg > WEDM_ORACLE_MAX_SEG) return -2;
    int32_t zone_start = (int32_t)wedm_oracle_py_floordiv(cfg->buffer_len_bottom, seg);   /* :151 */
    int32_t zone_end = zone_start + (int32_t)wedm_oracle_py_floordiv(cfg->workpiece_height, seg); /* :152-154 */
    zone_end = imin(zone_end, n_seg);                                                     /* :155 */
    zone_start = imin(zone_start, zone_end);                                              /* :156 */
    double r_wire = cfg->wire_diameter / 2.0;                                             /* :158 */
    double delta_y = seg * 1e-3;                                                          /* :169 */
    double rm = r_wire * 1e-3;
    double S = M_PI * pow(rm, 2.0);                                                       /* :170 `** 2` */
    double A = 2 * M_PI * rm * delta_y;                                                   /* :171 */
    o->k_cond = cfg->thermal_conductivity * S / delta_y;                                  /* :174-176 */
    double denominator = cfg->density * cfg->specific_heat * S * delta_y;                 /* :177-182 */
    o->joule_geom = (S != 0) ? delta_y / S : 0.0;                                         /* :183 */
    o->rho_elec = cfg->electrical_resistivity;
    o->alpha_rho = cfg->temperature_coefficient;
    o->temp_ref = 293.15;                                                                 /* :192 */
    if (denominator == 0) return -3;
    o->tuf = 1e-6 / denominator;                                                          /* :195 */
    o->a_surf = A;
    o->s_area = S;
    o->segment_len = seg;
    o->rho_c = cfg->density * cfg->specific_heat;                                         /* :305-310 prefix */
    o->az_start = imin(zone_start, n_seg - 1);                                            /* :207 */
    o->az_end = imin(zone_end, n_seg);                                                    /* :208 */
    o->critical_temperature = cfg->melting_point * cfg->critical_temp_threshold;          /* :216-218 */
    o->breaking_temperature = cfg->breaking_temperature;                                  /* :220 */
    double cb_pos = cfg->buffer_len_bottom - cfg->contact_offset_bottom;                  /* :229-231 */
    double ct_pos = cfg->buffer_len_bottom + cfg->workpiece_height + cfg->contact_offset_top; /* :232-236 */
    int32_t cb = imax(0, (int32_t)(cb_pos / seg));                                        /* :239-241 */
    int32_t ct = imin(n_seg - 1, (int32_t)(ct_pos / seg));                                /* :242-244 */
    cb = imax(0, imin(cb, zone_start - 1));                                               /* :247-249 */
    ct = imin(n_seg - 1, imax(ct, zone_end));                                             /* :250-252 */
    o->n_seg = n_seg; o->zone_start = zone_start; o->zone_end = zone_end;
    o->contact_bottom = cb; o->contact_top = ct;
    o->spool_T = cfg->spool_T;
    o->plasma_efficiency = cfg->plasma_efficiency;
    o->base_convection = cfg->base_convection_coefficient;
    o->convection_velocity_factor = cfg->convection_velocity_factor;
    o->convection_flow_enhancement = cfg->convection_flow_enhancement;

    /* ---- IgnitionModule, ignition.py:62-83 ---- */
    o->base_critical_density = cfg->base_critical_density;
    o->gap_coefficient = cfg->gap_coefficient;
    o->max_critical_density = cfg->max_critical_density;
    o->hard_short_gap = cfg->hard_short_gap;
    o->sigmoid_steepness = cfg->sigmoid_steepness;
    o->debris_short_duration = cfg->debris_short_duration;
    o->random_short_duration = cfg->random_short_duration;
    o->random_short_min_gap = cfg->random_short_min_gap;
    o->random_short_max_gap = cfg->random_short_max_gap;
    o->random_short_max_probability = cfg->random_short_max_probability;
    o->ignition_a = cfg->ignition_a_coeff;
    o->ignition_b = cfg->ignition_b_coeff;
    o->ignition_c = cfg->ignition_c_coeff;
    o->ln2 = log(2.0); /* np.log(2), ignition.py:362 */
    o->default_target_voltage = cfg->default_target_voltage;
    o->default_on_time = cfg->default_on_time;
    o->default_off_time = cfg->default_off_time;
    for (int i = 0; i <= WEDM_MAX_MODE; ++i) o->mode_current[i] = k_mode_current[i];
    o->default_current = k_mode_current[cfg->default_current_mode];                       /* :98-113 */
    o->spark_voltage_factor = cfg->spark_voltage_factor;

    /* ---- MaterialRemovalModule, material.py:28-50,140-174 ---- */
    o->kerf_base = cfg->base_overcut + cfg->wire_diameter;                                /* :158-160 prefix */
    for (unsigned i = 0; i < sizeof(k_crater) / sizeof(k_crater[0]); ++i) {
        int m = k_crater[i].mode;
        o->crater_mean[m] = k_crater[i].mean;
        o->crater_std[m] = k_crater[i].std;
        o->crater_depth[m] = k_crater[i].depth;
        o->crater_valid[m] = 1;
    }

    /* ---- DielectricModule, dielectric.py:57-80 ---- */
    o->cavity_coeff = M_PI * r_wire * cfg->workpiece_height;                              /* :62 */
    o->reference_gap = cfg->reference_gap;
    o->debris_obstruction_coeff = cfg->debris_obstruction_coeff;
    o->debris_removal_per_us = cfg->debris_removal_efficiency * cfg->base_flow_rate * 1e-6; /* :64-66 */
    o->dielectric_temperature = cfg->dielectric_temperature;

    /* ---- MechanicsModule, mechanics.py:29-67 ---- */
    o->dt_s = cfg->dt * 1e-6;                                                             /* :48 */
    o->damping_coeff = -2.0 * cfg->zeta * cfg->omega_n;                                   /* :52 */
    o->stiffness_coeff = -pow(cfg->omega_n, 2.0);                                         /* :53 */
    o->omega_n = cfg->omega_n;
    o->max_acceleration = cfg->max_acceleration;
    o->max_jerk_dt = cfg->max_jerk * o->dt_s;                                             /* :57 */
    o->max_speed = cfg->max_speed;
    return 0;
}

/* WireEDMEnv.reset (wire_edm.py:106-114) on a freshly constructed environment:
 * new EDMState() with the config's initial gap and target; module-private state
 * at its constructor values. */
void wedm_oracle_reset(wedm_oracle_env* e) {
    const wedm_oracle_consts* c = &e->c;
    e->error = 0;
    e->time = 0; e->time_since_servo = 0; e->time_since_open_voltage = 0;
    e->time_since_spark_ignition = 0; e->time_since_spark_end = 0;
    e->voltage = 0.0; e->current = 0.0; /* None */
    e->target_voltage = 0.0; e->on_time = 0.0; e->off_time = 0.0; e->current_mode = 0; /* None */
    e->workpiece_position = c->initial_gap; /* wire_edm.py:111 */
    e->wire_position = 0.0; e->wire_velocity = 0.0; e->wire_unwinding_velocity = 0.2; /* state.py:52-55 */
    e->time_in_critical_temp = 0;
    e->spark_state = 0; e->spark_dur = 0; e->spark_y = NAN; /* [0, None, 0] */
    e->dielectric_temperature = 0.0;
    e->debris_volume = 0.0; e->debris_density = 0.0; e->cavity_volume = 0.0; e->flow_rate = 0.0;
    e->last_crater_volume = 0.0;
    e->is_short_circuit = 0; e->is_wire_broken = 0; e->is_target_reached = 0;
    e->target_delta = 0.0;
    e->target_position = c->target_cutting_distance; /* wire_edm.py:112 */
    e->random_short_remaining = 0; e->debris_short_remaining = 0;
    e->diel_last_gap = -1.0; e->diel_last_density = -1.0; /* dielectric.py:78-79 */
    e->wire_last_flow = 0.0;                               /* wire.py:224 */
    e->h_base = 0.0f; e->h_zone = 0.0f;                    /* wire.py:205 np.zeros */
    e->prev_accel = 0.0;
    e->volt_acc = 0.0; e->volt_sum = 0.0;
    e->spark_count = 0;
    e->crater_stat_sum = 0.0; e->crater_stat_sumsq = 0.0; e->crater_stat_min = INFINITY; e->crater_stat_max = -INFINITY;
    e->last_terminated = 0; e->last_ctrl_step = 0; e->last_early_return = 0;
    e->mode_cached = 0;
    for (int i = 0; i < c->n_seg; ++i) { e->T[i] = (float)c->spool_T; e->dT[i] = 0.0f; } /* wire.py:264-269 */
    e->tmax = (float)c->spool_T;
}

/* wire_edm.py:106-114 on a used environment: `self.state = EDMState()` + the two configured positions.  Everything the
 * MODULE objects keep outside the state survives: ignition.py:75-81 (short timers, current cache), dielectric.py:69-80
 * (debris volume; flow, gap and density caches), mechanics.py:60 (prev_accel), wire.py:205,224 (convection coefficients
 * and their flow cache), material.py:133 (crater list -> count and statistics).  The new state's empty
 * `wire_temperature` is re-allocated at the spool temperature by the first wire.update (wire.py:264-269). */
void wedm_oracle_reset_reference(wedm_oracle_env* e) {
    const wedm_oracle_consts* c = &e->c;
    e->error = 0;
    e->time = 0; e->time_since_servo = 0; e->time_since_open_voltage = 0;
    e->time_since_spark_ignition = 0; e->time_since_spark_end = 0;
    e->voltage = 0.0; e->current = 0.0;
    e->target_voltage = 0.0; e->on_time = 0.0; e->off_time = 0.0; e->current_mode = 0;
    e->workpiece_position = c->initial_gap;
    e->wire_position = 0.0; e->wire_velocity = 0.0; e->wire_unwinding_velocity = 0.2;
    e->time_in_critical_temp = 0;
    e->spark_state = 0; e->spark_dur = 0; e->spark_y = NAN;
    e->dielectric_temperature = 0.0;
    e->debris_density = 0.0; /* state.debris_density: what ignition reads on the first step (ignition.py:213) */
    e->cavity_volume = 0.0;
    e->last_crater_volume = 0.0;
    e->is_short_circuit = 0; e->is_wire_broken = 0; e->is_target_reached = 0;
    e->target_delta = 0.0;
    e->target_position = c->target_cutting_distance;
    e->volt_acc = 0.0; e->volt_sum = 0.0; /* the driver's history belongs to one run of its loop */
    e->last_terminated = 0; e->last_ctrl_step = 0; e->last_early_return = 0;
    for (int i = 0; i < c->n_seg; ++i) { e->T[i] = (float)c->spool_T; e->dT[i] = 0.0f; }
    e->tmax = (float)c->spool_T;
}

int32_t wedm_oracle_init(wedm_oracle_env* env, const wedm_oracle_config* cfg) {
    memset(env, 0, sizeof(*env));
    int32_t rc = wedm_oracle_derive(cfg, &env->c);
    if (rc != 0) return rc;
    env->math_mode = WEDM_ORACLE_MATH_LIBM;
    env->stencil_mode = WEDM_ORACLE_STENCIL_F32;
    env->rng.mode = WEDM_ORACLE_RNG_PHILOX;
    wedm_oracle_reset(env);
    return 0;
}

/* ------------------------------------------------------------- ignition */
/* ignition.py:329-343: `state.x or default` — None and 0.0 both fall back */
static double get_target_voltage(const wedm_oracle_env* e) {
    return e->target_voltage != 0.0 ? e->target_voltage : e->c.default_target_voltage;
}
static double get_on_time(const wedm_oracle_env* e) {
    return e->on_time != 0.0 ? e->on_time : e->c.default_on_time;
}
static double get_off_time(const wedm_oracle_env* e) {
    return e->off_time != 0.0 ? e->off_time : e->c.default_off_time;
}
/* ignition.py:98-113 with the cache of ignition.py:79-81.  A fresh module starts with
 * `_cached_current_mode = None, _cached_current_value = 60.0`, so `current_mode is None` (before
 * the first control-step latch) HITS the cache and yields 60 A whatever `default_current_mode`
 * says; that parameter only serves modes that are not in currents.json.
 * After a reset that keeps the module (wedm_oracle_reset_reference) the cache may hold a mode of the previous episode:
 * None then misses it and resolves through `default_current_mode`. */
static double get_peak_current(wedm_oracle_env* e) {
    int m = e->current_mode;
    if (m == 0) return e->mode_cached ? e->c.default_current : 60.0;
    e->mode_cached = 1; /* `_cached_current_mode` now names a mode (an unknown one is replaced by the default, :107-110) */
    if (m < 1 || m > WEDM_MAX_MODE) return e->c.default_current;
    return e->c.mode_current[m];
}

/* ignition.py:115-146 */
static double debris_short_probability(const wedm_oracle_env* e, double gap, double debris_density) {
    const wedm_oracle_consts* c = &e->c;
    if (gap < c->hard_short_gap) return 1.0;
    double critical = c->base_critical_density + c->gap_coefficient * gap;
    critical = critical < c->max_critical_density ? critical : c->max_critical_density; /* min(a, b) */
    double delta = debris_density - critical;
    double exponent = -c->sigmoid_steepness * delta;
    if (exponent > 500) return 0.0;
    if (exponent < -500) return 1.0;
    return 1.0 / (1.0 + wedm_oracle_exp(exponent, e->math_mode));
}

/* ignition.py:197-245 */
static void update_short_circuit_detection(wedm_oracle_env* e) {
    const wedm_oracle_consts* c = &e->c;
    double d = e->workpiece_position - e->wire_position;
    double gap = d > 0.0 ? d : 0.0; /* max(0.0, d) */
    double debris_density = e->debris_density;
    if (e->random_short_remaining > 0) {
        e->random_short_remaining -= 1;
        e->is_short_circuit = 1;
        return;
    }
    if (e->debris_short_remaining > 0) {
        e->debris_short_remaining -= 1;
        e->is_short_circuit = 1;
        return;
    }
    double p_debris = debris_short_probability(e, gap, debris_density);
    double p_random;
    if (gap >= c->random_short_max_gap) {
        p_random = 0.0;
    } else if (gap <= c->random_short_min_gap) {
        p_random = c->random_short_max_probability;
    } else {
        double gap_factor = 1.0 - (gap - c->random_short_min_gap) /
                                      (c->random_short_max_gap - c->random_short_min_gap);
        p_random = gap_factor * c->random_short_max_probability;
    }
    if (rng_random(e, 0) < p_debris) {
        e->debris_short_remaining = c->debris_short_duration;
        e->is_short_circuit = 1;
        return;
    }
    if (rng_random(e, 1) < p_random) {
        e->random_short_remaining = c->random_short_duration;
        e->is_short_circuit = 1;
        return;
    }
    e->is_short_circuit = 0;
}

/* ignition.py:348-364 (the dict memo is a pure cache) */
static double get_lambda(const wedm_oracle_env* e) {
    const wedm_oracle_consts* c = &e->c;
    double gap = e->workpiece_position - e->wire_position; /* unclamped */
    double denominator = c->ignition_a * oracle_square(gap, e->math_mode) + c->ignition_b * gap + c->ignition_c;
    return c->ln2 / denominator;
}

/* ignition.py:175-195 and the four handlers :247-319 */
static void ignition_update(wedm_oracle_env* e) {
    update_short_circuit_detection(e);
    if (e->is_short_circuit) e->voltage = 0;
    int s = e->spark_state;
    if (s == 0) { /* _handle_idle_state :247-268 */
        e->current = 0;
        if (e->is_short_circuit) {
            e->spark_state = -1; e->spark_y = NAN; e->spark_dur = 0;
            e->current = get_peak_current(e);
        } else {
            e->voltage = get_target_voltage(e);
            /* _should_ignite :321-327 */
            double lam = get_lambda(e);
            if (rng_random(e, 2) < lam) {
                double y = rng_uniform_y(e);
                e->spark_state = 1; e->spark_y = y; e->spark_dur = 0;
                e->voltage = get_target_voltage(e) * e->c.spark_voltage_factor;
                e->current = get_peak_current(e);
            }
        }
    } else if (s == 1) { /* _handle_spark_state :270-287 */
        int dur = e->spark_dur + 1;
        e->spark_dur = dur;
        if ((double)dur >= get_on_time(e)) {
            e->spark_state = -2;
            e->current = 0;
            if (!e->is_short_circuit) e->voltage = 0;
        } else {
            e->current = get_peak_current(e);
            if (!e->is_short_circuit) e->voltage = get_target_voltage(e) * e->c.spark_voltage_factor;
        }
    } else if (s == -1) { /* _handle_short_state :289-300 */
        int dur = e->spark_dur + 1;
        e->spark_dur = dur;
        if ((double)dur >= get_on_time(e)) {
            e->spark_state = -2;
            e->current = 0;
        } else {
            e->current = get_peak_current(e);
        }
    } else if (s == -2) { /* _handle_rest_state :302-319 */
        int dur = e->spark_dur + 1;
        e->spark_dur = dur;
        double total = get_on_time(e) + get_off_time(e);
        if ((double)dur >= total) {
            e->spark_state = 0; e->spark_y = NAN; e->spark_dur = 0;
            e->current = 0;
            if (!e->is_short_circuit) e->voltage = get_target_voltage(e);
        } else {
            e->current = 0;
            if (!e->is_short_circuit) e->voltage = 0;
        }
    }
}

/* ------------------------------------------------------------- material */
/* material.py:79-96, :98-138, :140-174 */
static void material_update(wedm_oracle_env* e) {
    const wedm_oracle_consts* c = &e->c;
    if (e->spark_state == 1 && e->spark_dur == 0) {
        int mode = e->current_mode;
        if (mode == 0) mode = 1; /* None -> "I1" :104-105 (not ignition's I5) */
        if (mode < 1 || mode > WEDM_MAX_MODE || !c->crater_valid[mode]) {
            e->error |= 1; /* the reference raises ValueError here, :108-113 */
            mode = 1;
        }
        double sampled_um3 = rng_normal(e, c->crater_mean[mode], c->crater_std[mode]); /* :127 */
        if (!(sampled_um3 > 0)) sampled_um3 = 0; /* max(0, x) :130 */
        if (e->crater_log && e->crater_log_capacity > 0) /* crater_volumes_um3.append(sampled_volume_um3) :133 */
            e->crater_log[((int64_t)e->spark_count % e->crater_log_capacity) * e->crater_log_stride] = sampled_um3;
        e->spark_count += 1;                     /* :133 */
        /* running form of get_crater_statistics (material.py:207-227) */
        e->crater_stat_sum += sampled_um3;
        e->crater_stat_sumsq += sampled_um3 * sampled_um3;
        if (sampled_um3 < e->crater_stat_min) e->crater_stat_min = sampled_um3;
        if (sampled_um3 > e->crater_stat_max) e->crater_stat_max = sampled_um3;
        double crater_volume = sampled_um3 / 1e9; /* :136 */
        e->last_crater_volume = crater_volume;
        if (crater_volume > 0) {
            double depth_mm = c->crater_depth[mode] / 1000.0;                /* :157 */
            double kerf = c->kerf_base + depth_mm;                           /* :158-160 */
            double h = c->workpiece_height;
            double dx_um;
            if (kerf > 0 && h > 0) {
                double dx_mm = crater_volume / (kerf * h);                   /* :168 */
                dx_um = dx_mm * 1000.0;                                      /* :169 */
            } else {
                dx_um = 0.0;
            }
            e->workpiece_position += dx_um;                                  /* :93 */
        }
    } else {
        e->last_crater_volume = 0.0;
    }
}

/* ----------------------------------------------------------- dielectric */
/* dielectric.py:34-41 */
static double fast_exp(double x, int32_t math_mode) {
    if (x < 0.5) return (1 - 0.5 * x) / (1 + 0.5 * x);
    return wedm_oracle_exp(-x, math_mode);
}

/* dielectric.py:82-163 */
static void dielectric_update(wedm_oracle_env* e) {
    const wedm_oracle_consts* c = &e->c;
    e->dielectric_temperature = c->dielectric_temperature;
    double d = e->workpiece_position - e->wire_position;
    double gap_um = d > 0.001 ? d : 0.001; /* max(0.001, d) */
    double gap_mm = gap_um * 0.001;
    e->cavity_volume = c->cavity_coeff * gap_mm;
    if (e->spark_state == 1 && e->spark_dur == 0) {
        double crater = e->last_crater_volume;
        if (crater > 0) e->debris_volume += crater;
    }
    if (e->cavity_volume > 0) {
        double q = e->debris_volume / e->cavity_volume;
        e->debris_density = q < 1.0 ? q : 1.0; /* min(1.0, q) */
    } else {
        e->debris_density = 0.0;
    }
    if (fabs(gap_um - e->diel_last_gap) > 0.01 || fabs(e->debris_density - e->diel_last_density) > 0.001) {
        double cube = wedm_oracle_cube(gap_um / c->reference_gap, e->math_mode);
        double gap_factor = cube < 1.0 ? cube : 1.0; /* min(1.0, cube) */
        double kd = c->debris_obstruction_coeff * e->debris_density;
        double debris_factor;
        if (kd < 2.0) debris_factor = fast_exp(kd, e->math_mode);
        else debris_factor = wedm_oracle_exp(-c->debris_obstruction_coeff * e->debris_density, e->math_mode);
        e->flow_rate = gap_factor * debris_factor;
        e->diel_last_gap = gap_um;
        e->diel_last_density = e->debris_density;
    } /* else: flow_rate keeps the cached value (== _last_flow_condition) */
    if (e->flow_rate > 0.001 && e->debris_volume > 0.001) {
        double removed = c->debris_removal_per_us * e->flow_rate;
        double nv = e->debris_volume - removed;
        e->debris_volume = nv > 0.0 ? nv : 0.0; /* max(0.0, nv) */
    }
}

/* ----------------------------------------------------------------- wire */
/* wire.py:349-374 */
static void update_convection_coefficients(wedm_oracle_env* e, double v_unwind, double flow) {
    const wedm_oracle_consts* c = &e->c;
    double ve = c->convection_velocity_factor * v_unwind;
    ve = ve > -0.9 ? ve : -0.9; /* max(-0.9, ve) */
    double h_base = c->base_convection * (1.0 + ve);
    double floor_h = 0.1 * c->base_convection;
    h_base = floor_h > h_base ? floor_h : h_base; /* max(h_base, floor_h) */
    double h_enh = h_base * (1.0 + c->convection_flow_enhancement * flow);
    e->h_base = (float)h_base; /* ndarray.fill casts to float32 */
    e->h_zone = (float)h_enh;
}

/* wire.py:58-123 with NumPy-2 float32 scalar promotion (Numba stubbed out):
 * every Python-float operand is converted to float32 before the operation. */
static void thermal_update_f32(wedm_oracle_env* e, double I_squared, int plasma_idx, double plasma_heat,
                               double adv_coeff) {
    const wedm_oracle_consts* c = &e->c;
    float* T = e->T;
    float* dT = e->dT;
    int n = c->n_seg;
    const float spool = (float)c->spool_T, k = (float)c->k_cond, tref = (float)c->temp_ref;
    const float alpha = (float)c->alpha_rho, A = (float)c->a_surf, tuf = (float)c->tuf;
    const float tdiel = (float)e->dielectric_temperature;
    T[0] = spool;
    for (int i = 0; i < n; ++i) dT[i] = 0.0f;
    if (n > 1) {
        for (int i = 1; i < n - 1; ++i) {
            float t2 = 2.0f * T[i];
            float a = T[i - 1] - t2;
            float b = a + T[i + 1];
            dT[i] = k * b;
        }
        dT[n - 1] = k * (T[n - 2] - T[n - 1]);
    }
    if (I_squared > 1e-6) {
        float jf = (float)(c->joule_geom * I_squared * c->rho_elec);
        for (int i = c->contact_bottom; i <= c->contact_top; ++i) {
            float rho_T = 1.0f + alpha * (T[i] - tref);
            dT[i] = dT[i] + jf * rho_T;
        }
    }
    if (plasma_idx >= 0 && plasma_idx < n) dT[plasma_idx] = dT[plasma_idx] + (float)plasma_heat;
    for (int i = 0; i < n; ++i) {
        float h = (i >= c->az_start && i < c->az_end && c->az_start < c->az_end) ? e->h_zone : e->h_base;
        float conv = h * A;
        dT[i] = dT[i] - conv * (T[i] - tdiel);
    }
    if (fabs(adv_coeff) > 1e-9) {
        float adv = (float)adv_coeff;
        for (int i = 1; i < n; ++i) dT[i] = dT[i] + adv * (T[i - 1] - T[i]);
    }
    for (int i = 0; i < n; ++i) T[i] = T[i] + dT[i] * tuf;
    T[0] = spool;
}

/* Same expressions with Numba's typing: float64 arithmetic, rounded at each
 * float32 store (no fastmath re-association).  Unpinned; used to state the
 * tolerance between the two typings. */
static void thermal_update_f64(wedm_oracle_env* e, double I_squared, int plasma_idx, double plasma_heat,
                               double adv_coeff) {
    const wedm_oracle_consts* c = &e->c;
    float* T = e->T;
    float* dT = e->dT;
    int n = c->n_seg;
    T[0] = (float)c->spool_T;
    for (int i = 0; i < n; ++i) dT[i] = 0.0f;
    if (n > 1) {
        for (int i = 1; i < n - 1; ++i)
            dT[i] = (float)(c->k_cond * ((double)T[i - 1] - 2.0 * (double)T[i] + (double)T[i + 1]));
        dT[n - 1] = (float)(c->k_cond * ((double)T[n - 2] - (double)T[n - 1]));
    }
    if (I_squared > 1e-6) {
        double jf = c->joule_geom * I_squared * c->rho_elec;
        for (int i = c->contact_bottom; i <= c->contact_top; ++i) {
            double rho_T = 1.0 + c->alpha_rho * ((double)T[i] - c->temp_ref);
            dT[i] = (float)((double)dT[i] + jf * rho_T);
        }
    }
    if (plasma_idx >= 0 && plasma_idx < n) dT[plasma_idx] = (float)((double)dT[plasma_idx] + plasma_heat);
    for (int i = 0; i < n; ++i) {
        float h = (i >= c->az_start && i < c->az_end && c->az_start < c->az_end) ? e->h_zone : e->h_base;
        double conv = (double)h * c->a_surf;
        dT[i] = (float)((double)dT[i] - conv * ((double)T[i] - e->dielectric_temperature));
    }
    if (fabs(adv_coeff) > 1e-9)
        for (int i = 1; i < n; ++i) dT[i] = (float)((double)dT[i] + adv_coeff * ((double)T[i - 1] - (double)T[i]));
    for (int i = 0; i < n; ++i) T[i] = (float)((double)T[i] + (double)dT[i] * c->tuf);
    T[0] = (float)c->spool_T;
}

/* wire.py:259-347, :376-388 */
static void wire_update(wedm_oracle_env* e) {
    const wedm_oracle_consts* c = &e->c;
    if (e->is_wire_broken) return;
    double I = e->current; /* `state.current or 0.0` */
    double I_squared = I * I;
    double v_unwind = e->wire_unwinding_velocity;
    double flow = e->flow_rate;
    if (fabs(flow - e->wire_last_flow) > 0.01) {
        update_convection_coefficients(e, v_unwind, flow);
        e->wire_last_flow = flow;
    }
    int plasma_idx = -1;
    double plasma_heat = 0.0;
    if (e->spark_state == 1 && !isnan(e->spark_y)) {
        double y = e->spark_y;
        plasma_idx = (c->segment_len != 0)
                         ? c->zone_start + (int)wedm_oracle_py_floordiv(y, c->segment_len)
                         : c->zone_start;
        if (plasma_idx >= 0 && plasma_idx < c->n_seg) {
            plasma_heat = c->plasma_efficiency * e->voltage * I;
            if (!isfinite(plasma_heat)) plasma_heat = 0.0;
        }
    }
    double adv_coeff;
    if (fabs(v_unwind) > 1e-6) {
        double v_wire = fabs(v_unwind);
        adv_coeff = c->rho_c * v_wire * c->s_area; /* ((density*specific_heat)*v)*S */
    } else {
        adv_coeff = 0.0;
    }
    if (e->stencil_mode == WEDM_ORACLE_STENCIL_F32)
        thermal_update_f32(e, I_squared, plasma_idx, plasma_heat, adv_coeff);
    else
        thermal_update_f64(e, I_squared, plasma_idx, plasma_heat, adv_coeff);

    /* _check_wire_breaking :376-388.  np.max(T) is float32; NumPy 2 compares it
     * with the Python-float thresholds after casting them to float32. */
    float tmax = e->T[0];
    for (int i = 1; i < c->n_seg; ++i) tmax = e->T[i] > tmax ? e->T[i] : tmax;
    e->tmax = tmax;
    if (tmax > (float)c->critical_temperature) e->time_in_critical_temp += 1;
    else e->time_in_critical_temp = 0;
    if (tmax > (float)c->breaking_temperature) e->is_wire_broken = 1;
}

/* ------------------------------------------------------------ mechanics */
/* mechanics.py:69-114 */
static void mechanics_update(wedm_oracle_env* e) {
    const wedm_oracle_consts* c = &e->c;
    double x = e->wire_position, v = e->wire_velocity;
    double a_nom;
    if (c->control_mode == 0) {
        double x_error = x - (x + e->target_delta); /* :71 keep the literal rounding */
        a_nom = c->damping_coeff * v + c->stiffness_coeff * x_error;
    } else {
        double v_error = v - e->target_delta;
        a_nom = -c->omega_n * v_error;
    }
    if (a_nom > c->max_acceleration) a_nom = c->max_acceleration;
    else if (a_nom < -c->max_acceleration) a_nom = -c->max_acceleration;
    double da = a_nom - e->prev_accel;
    if (da > c->max_jerk_dt) da = c->max_jerk_dt;
    else if (da < -c->max_jerk_dt) da = -c->max_jerk_dt;
    double a = e->prev_accel + da;
    e->prev_accel = a;
    v += a * c->dt_s;
    if (v > c->max_speed) v = c->max_speed;
    else if (v < -c->max_speed) v = -c->max_speed;
    x += v * c->dt_s;
    e->wire_velocity = v;
    e->wire_position = x;
}

/* ------------------------------------------------------------------ step */
/* What the reference's driver does with `state.voltage` after every `env.step()`, the terminating one
 * included (experiments/run_simulation.py:256-270): it appends it to a history that keeps the samples of
 * the last 1000 us.  As a running sum in step order: `volt_acc` sums every step since (and including) the
 * last control step; a control step publishes it as `volt_sum` and restarts the accumulator from its own
 * sample.  With servo_interval = 1000 us `volt_sum` is the sum of exactly the <= 1001 samples
 * `create_voltage_controller` averages (run_simulation.py:70-72). */
static void voltage_history_update(wedm_oracle_env* e) {
    e->volt_acc = e->volt_acc + e->voltage;
    if (e->last_ctrl_step) {
        e->volt_sum = e->volt_acc;
        e->volt_acc = e->voltage;
    }
}

/* wire_edm.py:116-157, :162-179 */
int32_t wedm_oracle_step(wedm_oracle_env* e, const wedm_oracle_action* action) {
    const wedm_oracle_consts* c = &e->c;
    e->rng.draws_this_step = 0;
    e->last_early_return = 0;
    int is_ctrl = e->time_since_servo >= c->servo_interval;
    e->last_ctrl_step = is_ctrl;
    if (is_ctrl) { /* _apply_action :162-170 */
        e->target_delta = action->servo;
        e->target_voltage = action->target_voltage;
        e->current_mode = action->current_mode;
        e->on_time = action->on_time;
        e->off_time = action->off_time;
        e->time_since_servo = 0;
    }
    if (!e->disable_ignition) ignition_update(e);
    material_update(e);
    dielectric_update(e);
    wire_update(e);
    if (e->is_wire_broken) { /* :129-130 early return before mechanics and clocks */
        e->last_early_return = 1;
        e->last_terminated = 1;
        voltage_history_update(e);
        return 1;
    }
    mechanics_update(e);
    /* wire_edm.py:135-137.  `time` is an unbounded Python int there; here the low 32 bits (unsigned, wrapping: also the
     * Philox counter word) -- the batch driver carries the high word, row WEDM_I_TIME_HI */
    e->time = (int32_t)((uint32_t)e->time + (uint32_t)c->dt_us);
    e->time_since_servo += c->dt_us;
    e->time_since_open_voltage = (int32_t)((uint32_t)e->time_since_open_voltage + (uint32_t)c->dt_us);
    if (e->spark_state == 1) {
        e->time_since_spark_ignition += c->dt_us;
        e->time_since_spark_end = 0;
    } else {
        e->time_since_spark_end += c->dt_us;
        e->time_since_spark_ignition = 0;
    }
    /* _check_termination :172-179 */
    int terminated = 0;
    if (e->wire_position > e->workpiece_position + 100) {
        e->is_wire_broken = 1;
        terminated = 1;
    } else if (e->workpiece_position >= e->target_position) {
        e->is_target_reached = 1;
        terminated = 1;
    }
    e->last_terminated = terminated;
    voltage_history_update(e);
    return terminated;
}

/* ============================================================ batch driver
 * CPU restatement at the C-ABI's shape: gathers one environment from the SoA
 * blocks, steps it, scatters it back.  Documented batch semantics that the
 * reference (single env) does not define: a terminated environment is frozen
 * (DONE) until reset; reset also clears module-private state.                */
static void consts_from_params(const wedm_params* p, wedm_oracle_consts* c) {
    memset(c, 0, sizeof(*c));
    c->servo_interval = p->servo_interval; c->dt_us = p->dt_us; c->control_mode = p->control_mode;
    c->n_seg = p->n_seg; c->zone_start = p->zone_start; c->zone_end = p->az_end;
    c->az_start = p->az_start; c->az_end = p->az_end;
    c->contact_bottom = p->contact_bottom; c->contact_top = p->contact_top;
    c->initial_gap = p->initial_gap; c->target_cutting_distance = p->target_cutting_distance;
    c->workpiece_height = p->workpiece_height; c->kerf_base = p->kerf_base; c->cavity_coeff = p->cavity_coeff;
    c->k_cond = p->k_cond; c->tuf = p->tuf; c->a_surf = p->a_surf; c->s_area = p->s_area;
    c->joule_geom = p->joule_geom; c->segment_len = p->segment_len;
    c->spool_T = p->spool_T; c->temp_ref = p->temp_ref; c->rho_elec = p->rho_elec;
    c->alpha_rho = p->alpha_rho; c->rho_c = p->rho_c;
    c->plasma_efficiency = p->plasma_efficiency; c->base_convection = p->base_convection;
    c->convection_velocity_factor = p->convection_velocity_factor;
    c->convection_flow_enhancement = p->convection_flow_enhancement;
    c->critical_temperature = p->critical_temperature; c->breaking_temperature = p->breaking_temperature;
    c->dielectric_temperature = p->dielectric_temperature;
    c->base_critical_density = p->base_critical_density; c->gap_coefficient = p->gap_coefficient;
    c->max_critical_density = p->max_critical_density; c->hard_short_gap = p->hard_short_gap;
    c->sigmoid_steepness = p->sigmoid_steepness;
    c->debris_short_duration = p->debris_short_duration; c->random_short_duration = p->random_short_duration;
    c->random_short_min_gap = p->random_short_min_gap; c->random_short_max_gap = p->random_short_max_gap;
    c->random_short_max_probability = p->random_short_max_probability;
    c->ignition_a = p->ignition_a; c->ignition_b = p->ignition_b; c->ignition_c = p->ignition_c; c->ln2 = p->ln2;
    c->default_target_voltage = p->default_target_voltage; c->default_on_time = p->default_on_time;
    c->default_off_time = p->default_off_time; c->default_current = p->default_current;
    c->spark_voltage_factor = p->spark_voltage_factor;
    c->reference_gap = p->reference_gap; c->debris_obstruction_coeff = p->debris_obstruction_coeff;
    c->debris_removal_per_us = p->debris_removal_per_us;
    c->dt_s = p->dt_s; c->damping_coeff = p->damping_coeff; c->stiffness_coeff = p->stiffness_coeff;
    c->omega_n = p->omega_n; c->max_acceleration = p->max_acceleration; c->max_jerk_dt = p->max_jerk_dt;
    c->max_speed = p->max_speed;
    for (int i = 0; i <= WEDM_MAX_MODE; ++i) {
        c->mode_current[i] = p->mode_current[i];
        c->crater_mean[i] = p->crater_mean[i];
        c->crater_std[i] = p->crater_std[i];
        c->crater_depth[i] = p->crater_depth[i];
        c->crater_valid[i] = p->crater_valid[i];
    }
}

static void apply_geometry(const wedm_geom_ptrs* g, int64_t stride, int64_t e, wedm_oracle_consts* c) {
    const double* gf = g->f64 + e;
    const int32_t* gi = g->i32 + e;
    c->workpiece_height = gf[WEDM_G_HEIGHT * stride];
    c->kerf_base = gf[WEDM_G_KERF_BASE * stride];
    c->cavity_coeff = gf[WEDM_G_CAVITY_COEFF * stride];
    c->k_cond = gf[WEDM_G_K_COND * stride];
    c->tuf = gf[WEDM_G_TUF * stride];
    c->a_surf = gf[WEDM_G_A_SURF * stride];
    c->s_area = gf[WEDM_G_S_AREA * stride];
    c->joule_geom = gf[WEDM_G_JOULE_GEOM * stride];
    c->n_seg = gi[WEDM_GI_N_SEG * stride];
    c->zone_start = gi[WEDM_GI_ZONE_START * stride];
    c->az_start = gi[WEDM_GI_AZ_START * stride];
    c->az_end = gi[WEDM_GI_AZ_END * stride];
    c->zone_end = c->az_end;
    c->contact_bottom = gi[WEDM_GI_CONTACT_BOTTOM * stride];
    c->contact_top = gi[WEDM_GI_CONTACT_TOP * stride];
}

#define F64(row) s->f64[(int64_t)(row) * stride + e]
#define I32(row) s->i32[(int64_t)(row) * stride + e]
#define I8(row) s->i8[(int64_t)(row) * stride + e]
#define STAT(row) s->stats[(int64_t)(row) * stride + e]

static void gather_env(const wedm_state_ptrs* s, int64_t e, wedm_oracle_env* v) {
    int64_t stride = s->stride;
    v->workpiece_position = F64(WEDM_F_WORKPIECE_POS); v->wire_position = F64(WEDM_F_WIRE_POS);
    v->wire_velocity = F64(WEDM_F_WIRE_VEL); v->prev_accel = F64(WEDM_F_PREV_ACCEL);
    v->debris_volume = F64(WEDM_F_DEBRIS_VOLUME); v->debris_density = F64(WEDM_F_DEBRIS_DENSITY);
    v->flow_rate = F64(WEDM_F_FLOW); v->diel_last_gap = F64(WEDM_F_LAST_GAP);
    v->diel_last_density = F64(WEDM_F_LAST_DENSITY); v->wire_last_flow = F64(WEDM_F_WIRE_LAST_FLOW);
    v->voltage = F64(WEDM_F_VOLTAGE); v->current = F64(WEDM_F_CURRENT); v->spark_y = F64(WEDM_F_SPARK_Y);
    v->last_crater_volume = F64(WEDM_F_LAST_CRATER); v->cavity_volume = F64(WEDM_F_CAVITY);
    v->target_delta = F64(WEDM_F_TARGET_DELTA); v->target_voltage = F64(WEDM_F_TARGET_VOLTAGE);
    v->on_time = F64(WEDM_F_ON_TIME); v->off_time = F64(WEDM_F_OFF_TIME);
    v->target_position = F64(WEDM_F_TARGET_POS); v->wire_unwinding_velocity = F64(WEDM_F_UNWIND_VEL);
    v->h_base = (float)F64(WEDM_F_H_BASE); v->h_zone = (float)F64(WEDM_F_H_ZONE);
    v->tmax = (float)F64(WEDM_F_TMAX);
    v->volt_acc = F64(WEDM_F_VOLT_ACC); v->volt_sum = F64(WEDM_F_VOLT_SUM);
    v->time = I32(WEDM_I_TIME); v->time_since_servo = I32(WEDM_I_SINCE_SERVO);
    v->time_since_open_voltage = I32(WEDM_I_SINCE_OPEN_V);
    v->time_since_spark_ignition = I32(WEDM_I_SINCE_IGNITION);
    v->time_since_spark_end = I32(WEDM_I_SINCE_SPARK_END);
    v->spark_dur = I32(WEDM_I_SPARK_DUR); v->random_short_remaining = I32(WEDM_I_RANDOM_SHORT_REM);
    v->debris_short_remaining = I32(WEDM_I_DEBRIS_SHORT_REM);
    v->time_in_critical_temp = I32(WEDM_I_TIME_CRITICAL);
    v->current_mode = I32(WEDM_I_CURRENT_MODE) < 0 ? 0 : I32(WEDM_I_CURRENT_MODE); /* -1: None over a stale current cache */
    v->rng.episode = (uint32_t)I32(WEDM_I_EPISODE);
    v->rng.seed = (uint64_t)(uint32_t)I32(WEDM_I_KEY_LO) | ((uint64_t)(uint32_t)I32(WEDM_I_KEY_HI) << 32);
    v->spark_count = I32(WEDM_I_SPARK_COUNT);
    if (s->stats) {
        v->crater_stat_sum = STAT(WEDM_S_CRATER_SUM); v->crater_stat_sumsq = STAT(WEDM_S_CRATER_SUMSQ);
        v->crater_stat_min = STAT(WEDM_S_CRATER_MIN); v->crater_stat_max = STAT(WEDM_S_CRATER_MAX);
    }
    v->spark_state = I8(WEDM_B_SPARK_STATE); v->is_short_circuit = I8(WEDM_B_IS_SHORT);
    v->is_wire_broken = I8(WEDM_B_WIRE_BROKEN); v->is_target_reached = I8(WEDM_B_TARGET_REACHED);
    v->error = I8(WEDM_B_ERROR);
    v->mode_cached = I8(WEDM_B_MODE_CACHED);
    v->dielectric_temperature = v->c.dielectric_temperature;
    for (int i = 0; i < v->c.n_seg; ++i) v->T[i] = s->T[WEDM_T_INDEX(i, stride, e)];  /* quad-interleaved block, include/wedm_hip.h */
}

static void scatter_env(const wedm_state_ptrs* s, int64_t e, const wedm_oracle_env* v, int done) {
    int64_t stride = s->stride;
    F64(WEDM_F_WORKPIECE_POS) = v->workpiece_position; F64(WEDM_F_WIRE_POS) = v->wire_position;
    F64(WEDM_F_WIRE_VEL) = v->wire_velocity; F64(WEDM_F_PREV_ACCEL) = v->prev_accel;
    F64(WEDM_F_DEBRIS_VOLUME) = v->debris_volume; F64(WEDM_F_DEBRIS_DENSITY) = v->debris_density;
    F64(WEDM_F_FLOW) = v->flow_rate; F64(WEDM_F_LAST_GAP) = v->diel_last_gap;
    F64(WEDM_F_LAST_DENSITY) = v->diel_last_density; F64(WEDM_F_WIRE_LAST_FLOW) = v->wire_last_flow;
    F64(WEDM_F_VOLTAGE) = v->voltage; F64(WEDM_F_CURRENT) = v->current; F64(WEDM_F_SPARK_Y) = v->spark_y;
    F64(WEDM_F_LAST_CRATER) = v->last_crater_volume; F64(WEDM_F_CAVITY) = v->cavity_volume;
    F64(WEDM_F_TARGET_DELTA) = v->target_delta; F64(WEDM_F_TARGET_VOLTAGE) = v->target_voltage;
    F64(WEDM_F_ON_TIME) = v->on_time; F64(WEDM_F_OFF_TIME) = v->off_time;
    F64(WEDM_F_TARGET_POS) = v->target_position; F64(WEDM_F_UNWIND_VEL) = v->wire_unwinding_velocity;
    F64(WEDM_F_H_BASE) = (double)v->h_base; F64(WEDM_F_H_ZONE) = (double)v->h_zone;
    F64(WEDM_F_TMAX) = (double)v->tmax;
    F64(WEDM_F_VOLT_ACC) = v->volt_acc; F64(WEDM_F_VOLT_SUM) = v->volt_sum;
    I32(WEDM_I_TIME) = v->time; I32(WEDM_I_SINCE_SERVO) = v->time_since_servo;
    I32(WEDM_I_SINCE_OPEN_V) = v->time_since_open_voltage;
    I32(WEDM_I_SINCE_IGNITION) = v->time_since_spark_ignition;
    I32(WEDM_I_SINCE_SPARK_END) = v->time_since_spark_end;
    I32(WEDM_I_SPARK_DUR) = v->spark_dur; I32(WEDM_I_RANDOM_SHORT_REM) = v->random_short_remaining;
    I32(WEDM_I_DEBRIS_SHORT_REM) = v->debris_short_remaining;
    I32(WEDM_I_TIME_CRITICAL) = v->time_in_critical_temp;
    I32(WEDM_I_CURRENT_MODE) = (v->current_mode == 0 && v->mode_cached) ? -1 : v->current_mode; /* include/wedm_hip.h */
    I32(WEDM_I_SPARK_COUNT) = v->spark_count;
    if (s->stats) {
        STAT(WEDM_S_CRATER_SUM) = v->crater_stat_sum; STAT(WEDM_S_CRATER_SUMSQ) = v->crater_stat_sumsq;
        STAT(WEDM_S_CRATER_MIN) = v->crater_stat_min; STAT(WEDM_S_CRATER_MAX) = v->crater_stat_max;
    }
    I8(WEDM_B_SPARK_STATE) = (int8_t)v->spark_state; I8(WEDM_B_IS_SHORT) = (int8_t)v->is_short_circuit;
    I8(WEDM_B_WIRE_BROKEN) = (int8_t)v->is_wire_broken; I8(WEDM_B_TARGET_REACHED) = (int8_t)v->is_target_reached;
    I8(WEDM_B_DONE) = (int8_t)done; I8(WEDM_B_CTRL_STEP) = (int8_t)v->last_ctrl_step;
    I8(WEDM_B_ERROR) = (int8_t)(v->error & 1);
    I8(WEDM_B_MODE_CACHED) = (int8_t)(v->mode_cached & 1);
    for (int i = 0; i < v->c.n_seg; ++i) s->T[WEDM_T_INDEX(i, stride, e)] = v->T[i];
}

/* observation columns written at control steps (build-defined; the reference's
 * _get_obs is a TODO, wire_edm.py:181-183) */
static void write_obs(const wedm_params* p, const wedm_state_ptrs* s, int64_t e, const wedm_oracle_env* v) {
    if (!s->obs || p->obs_dim < 8) return;
    int64_t stride = s->stride;
    float* o = s->obs + e;
    o[0 * stride] = (float)(v->workpiece_position - v->wire_position);
    o[1 * stride] = (float)v->wire_velocity;
    o[2 * stride] = (float)v->voltage;
    o[3 * stride] = (float)v->current;
    o[4 * stride] = (float)v->spark_state;
    o[5 * stride] = (float)v->debris_density;
    o[6 * stride] = (float)v->flow_rate;
    o[7 * stride] = v->tmax;
}

int32_t wedm_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* wedm_reset for one environment: every row at its constructor / reset() value (wire_edm.py:106-114 on a
 * fresh environment), Philox episode bumped or the stream re-keyed. */
static void reset_env_rows(const wedm_params* p, const wedm_state_ptrs* s, int64_t e, uint64_t seed, int32_t reseed) {
    int64_t stride = s->stride;
    /* wedm_params.reset_semantics 1: the reference's own reset() (wire_edm.py:106-114) re-initialises EDMState only; the
     * rows that mirror what its module objects hold survive (see wedm_oracle_reset_reference) */
    const int keep_modules = p->reset_semantics != 0 && !(reseed & WEDM_RESET_FRESH);
    const uint32_t module_f64 = (1u << WEDM_F_PREV_ACCEL) | (1u << WEDM_F_DEBRIS_VOLUME) | (1u << WEDM_F_FLOW) |
                                (1u << WEDM_F_LAST_GAP) | (1u << WEDM_F_LAST_DENSITY) | (1u << WEDM_F_WIRE_LAST_FLOW) |
                                (1u << WEDM_F_H_BASE) | (1u << WEDM_F_H_ZONE);
    const uint32_t module_i32 = (1u << WEDM_I_RANDOM_SHORT_REM) | (1u << WEDM_I_DEBRIS_SHORT_REM) | (1u << WEDM_I_SPARK_COUNT);
    const uint32_t module_i8 = 1u << WEDM_B_MODE_CACHED;
    for (int f = 0; f < WEDM_F64_COUNT; ++f)
        if (!(keep_modules && ((module_f64 >> f) & 1u))) F64(f) = 0.0;
    int32_t episode = I32(WEDM_I_EPISODE), klo = I32(WEDM_I_KEY_LO), khi = I32(WEDM_I_KEY_HI);
    for (int f = 0; f < WEDM_I32_COUNT; ++f)
        if (!(keep_modules && ((module_i32 >> f) & 1u))) I32(f) = 0;
    for (int f = 0; f < WEDM_I8_COUNT; ++f)
        if (!(keep_modules && ((module_i8 >> f) & 1u))) I8(f) = 0;
    if (s->stats && !keep_modules) {
        STAT(WEDM_S_CRATER_SUM) = 0.0; STAT(WEDM_S_CRATER_SUMSQ) = 0.0;
        STAT(WEDM_S_CRATER_MIN) = INFINITY; STAT(WEDM_S_CRATER_MAX) = -INFINITY;
    }
    if (reseed & WEDM_RESET_RESEED) {
        I32(WEDM_I_EPISODE) = 0;
        I32(WEDM_I_KEY_LO) = (int32_t)(uint32_t)seed;
        I32(WEDM_I_KEY_HI) = (int32_t)(uint32_t)(seed >> 32);
    } else {
        I32(WEDM_I_EPISODE) = episode + 1;
        I32(WEDM_I_KEY_LO) = klo;
        I32(WEDM_I_KEY_HI) = khi;
    }
    F64(WEDM_F_WORKPIECE_POS) = p->initial_gap;
    F64(WEDM_F_TARGET_POS) = p->target_cutting_distance;
    F64(WEDM_F_UNWIND_VEL) = 0.2;
    F64(WEDM_F_SPARK_Y) = NAN;
    if (!keep_modules) {
        F64(WEDM_F_LAST_GAP) = -1.0;
        F64(WEDM_F_LAST_DENSITY) = -1.0;
    } else if (I8(WEDM_B_MODE_CACHED)) {
        I32(WEDM_I_CURRENT_MODE) = -1; /* None over a current cache that names a mode (include/wedm_hip.h) */
    }
    F64(WEDM_F_TMAX) = (double)(float)p->spool_T;
    if (s->reward) s->reward[e] = 0.0f;
}

int32_t wedm_oracle_reset_batch(const wedm_params* p, const wedm_state_ptrs* s, int32_t num_envs,
                                const uint8_t* mask, uint64_t seed, int32_t reseed) {
    if (!p || !s || !s->f64 || !s->i32 || !s->i8 || !s->T || num_envs <= 0) return WEDM_ERR_BAD_ARG;
    for (int64_t e = 0; e < num_envs; ++e) {
        if (mask && !mask[e]) continue;
        reset_env_rows(p, s, e, seed, reseed);
    }
    return WEDM_OK;
}

int32_t wedm_oracle_step_batch(const wedm_params* p, const wedm_state_ptrs* s, const wedm_geom_ptrs* g,
                               const wedm_action_ptrs* a, int32_t num_envs, int32_t n_seg_max, int32_t n_substeps,
                               int32_t math_mode, int32_t stencil_mode, int32_t n_threads) {
    if (!p || !s || !a || num_envs <= 0 || n_substeps < 0 || n_seg_max < 1) return WEDM_ERR_BAD_ARG;
    if (n_substeps == 0) return WEDM_OK;
    if (p->per_env_geometry && (!g || !g->f64 || !g->i32)) return WEDM_ERR_BAD_ARG;
    int64_t stride = s->stride;
    int bad = 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel num_threads(n_threads)
#endif
    {
        wedm_oracle_env* v = (wedm_oracle_env*)malloc(sizeof(wedm_oracle_env));
        memset(v, 0, sizeof(*v));
        consts_from_params(p, &v->c);
        v->math_mode = math_mode;
        v->stencil_mode = stencil_mode;
        v->rng.mode = WEDM_ORACLE_RNG_PHILOX;
        v->disable_ignition = p->disable_ignition;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t e = 0; e < num_envs; ++e) {
            if (I8(WEDM_B_DONE)) {
                if (!p->autoreset && !p->keep_stepping_terminated) {  /* frozen until the caller resets it: it earns nothing in this launch */
                    if (p->reward_mode && s->reward) s->reward[e] = 0.0f;
                    continue;
                }
            }
            if (I8(WEDM_B_DONE) && p->autoreset) {
                /* wedm_params.autoreset: next-step autoreset inside the call = wedm_reset(mask = DONE, reseed = 0)
                 * for this environment (all n_seg_max wire rows at the spool temperature, observation zeroed) */
                reset_env_rows(p, s, e, 0, 0);
                for (int i = 0; i < 4 * WEDM_T_QUADS(n_seg_max); ++i) s->T[WEDM_T_INDEX(i, stride, e)] = (float)p->spool_T;
                if (s->obs)
                    for (int q = 0; q < p->obs_dim; ++q) s->obs[(int64_t)q * stride + e] = 0.0f;
            }
            if (p->per_env_geometry) apply_geometry(g, stride, e, &v->c);
            if (v->c.n_seg > WEDM_ORACLE_MAX_SEG || v->c.n_seg < 1) { bad = 1; continue; }
            gather_env(s, e, v);
            v->crater_log = s->crater_log ? s->crater_log + e : NULL;
            v->crater_log_capacity = s->crater_log_capacity;
            v->crater_log_stride = stride;
            v->rng.env_id = p->env_id_offset + (uint32_t)e;
            wedm_oracle_action act = {a->servo[e], a->target_voltage[e], a->on_time[e], a->off_time[e],
                                      a->current_mode[e]};
            int done = 0;
            const double wp0 = v->workpiece_position;
            const uint32_t t0 = (uint32_t)v->time;  /* low word of state.time at the start of the launch */
            /* a terminated environment is frozen; with wedm_params.keep_stepping_terminated it is stepped on as the
             * reference's step() would be (wire_edm.py:116-157 has no guard), DONE = `terminated` of the last step */
            for (int k = 0; k < n_substeps && (!done || p->keep_stepping_terminated); ++k) {
                done = wedm_oracle_step(v, &act);
                if (v->last_ctrl_step) write_obs(p, s, e, v);
            }
            /* wedm_params.reward_mode 1: the launch's progress reward (the reference's is a TODO, wire_edm.py:185-187) */
            if (p->reward_mode && s->reward)
                s->reward[e] = (float)(v->workpiece_position - wp0) -
                               (float)p->reward_break_penalty * (v->is_wire_broken ? 1.0f : 0.0f);
            /* the clock's high word: the low word wrapped in this launch iff it ended below where it started
             * (a launch advances an environment by less than 2^32 us) */
            if ((uint32_t)v->time < t0) I32(WEDM_I_TIME_HI) += 1;
            scatter_env(s, e, v, done);
        }
        free(v);
    }
    return bad ? WEDM_ERR_BAD_ARG : WEDM_OK;
}

/* layout cross-check for the ctypes mirror in oracle/oracle.py */
int64_t wedm_oracle_sizeof(int32_t which) {
    switch (which) {
        case 0: return (int64_t)sizeof(wedm_oracle_config);
        case 1: return (int64_t)sizeof(wedm_oracle_consts);
        case 2: return (int64_t)sizeof(wedm_oracle_rng);
        case 3: return (int64_t)sizeof(wedm_oracle_env);
        case 4: return (int64_t)sizeof(wedm_oracle_action);
        case 5: return (int64_t)sizeof(wedm_params);
        default: return -1;
    }
}
