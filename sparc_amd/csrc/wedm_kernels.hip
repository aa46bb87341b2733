// wedm_kernels.hip — gfx950 kernels + the C-ABI of include/wedm_hip.h.
//
// Kernels
//   wedm_step_global : one lane per environment, wire temperature walked in place in
//                      global memory (layout T[seg][env] -> every row access is a
//                      256-B coalesced wave transaction).  One HBM/L2 pass per
//                      microsecond; used for n_substeps == 1 (the reference's step()).
//   wedm_step_lds    : same lanes, but the wave first stages its 64 wire columns into
//                      LDS ([seg][lane], conflict-free: lane l always hits bank l%32 in
//                      its half), runs n_substeps microseconds out of LDS + registers,
//                      and writes everything back once.  No barriers: a lane only ever
//                      touches its own LDS column.
//   wedm_reset_kernel: WireEDMEnv.reset for a masked subset.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math (see
// __graft_entry__.build()).  -ffp-contract=off is part of the numerics contract.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "wedm_device.h"

using namespace wedm;

struct KArgs {
    wedm_params p;
    wedm_state_ptrs s;
    wedm_geom_ptrs g;
    wedm_action_ptrs a;
    Tables tb;
    int32_t num_envs;
    int32_t n_substeps;
    int32_t n_seg_max;
};

// ------------------------------------------------------------ T accessors
struct GlobalT {
    float* base;     // &T[0][e]
    int64_t stride;  // elements between consecutive segments
    __device__ __forceinline__ float ld(int i) const { return base[(int64_t)i * stride]; }
    __device__ __forceinline__ void st(int i, float v) const { base[(int64_t)i * stride] = v; }
};
struct LdsT {
    float* base;  // &lds[wave region][0][lane]
    __device__ __forceinline__ float ld(int i) const { return base[i * 64]; }
    __device__ __forceinline__ void st(int i, float v) const { base[i * 64] = v; }
};

// One in-place pass of wire.py:58-123 over the lane's wire.  Tiles of 8 cells: the 8
// "next" temperatures are loaded before any of the tile's stores, so every cell sees
// OLD neighbours (explicit Euler) with one load + one store per cell.
template <class TA>
__device__ __forceinline__ float stencil_pass(const TA& T, const Geom& g, const Coef& c, float spool, float tref,
                                              float alpha, float tdiel) {
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
                float tn = stencil_cell(i, n, tm1, tc, nx[u], g, c, tref, alpha, tdiel);
                T.st(i, tn);
                tmax = tn > tmax ? tn : tmax;
                tm1 = tc;
                tc = nx[u];
            }
        }
    }
    return tmax;
}

template <class TA>
__device__ __forceinline__ void run_substeps(const KArgs& k, const Geom& g, int64_t e, uint32_t gid, Env& s,
                                             const TA& T) {
    const float spool = (float)k.p.spool_T, tref = (float)k.p.temp_ref, alpha = (float)k.p.alpha_rho;
    const float tdiel = (float)k.p.dielectric_temperature;
    for (int it = 0; it < k.n_substeps; ++it) {
        if (s.done) break;
        Coef c = scalar_prelude(k.p, g, k.tb, k.a, e, gid, s);
        float tmax = stencil_pass(T, g, c, spool, tref, alpha, tdiel);
        scalar_epilogue(k.p, s, tmax);
        if (s.ctrl) write_obs(k.p, k.s, e, s);
    }
}

__global__ void __launch_bounds__(256) wedm_step_global(const KArgs k) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k.num_envs) return;
    Env s;
    load_env(k.s, e, s);
    if (s.done) return;
    s.ipk = peak_current(k.p, k.tb, s.mode);
    Geom g;
    load_geom(k.p, k.g, k.s.stride, e, g);
    GlobalT T{k.s.T + e, k.s.stride};
    run_substeps(k, g, e, k.p.env_id_offset + (uint32_t)e, s, T);
    store_env(k.s, e, s);
}

__global__ void __launch_bounds__(64) wedm_step_lds(const KArgs k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    if (e >= k.num_envs) return;
    Env s;
    load_env(k.s, e, s);
    if (s.done) return;
    s.ipk = peak_current(k.p, k.tb, s.mode);
    Geom g;
    load_geom(k.p, k.g, k.s.stride, e, g);
    LdsT T{lds + lane};
    const float* src = k.s.T + e;
    const int64_t stride = k.s.stride;
    for (int i = 0; i < g.n_seg; ++i) T.st(i, src[(int64_t)i * stride]);
    run_substeps(k, g, e, k.p.env_id_offset + (uint32_t)e, s, T);
    float* dst = k.s.T + e;
    for (int i = 0; i < g.n_seg; ++i) dst[(int64_t)i * stride] = T.ld(i);
    store_env(k.s, e, s);
}

__global__ void __launch_bounds__(256)
wedm_reset_kernel(const wedm_params p, const wedm_state_ptrs s, int32_t num_envs, int32_t n_seg_max,
                  const uint8_t* mask, uint32_t key_lo, uint32_t key_hi, int32_t reseed) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= num_envs) return;
    if (mask && !mask[e]) return;
    const int64_t stride = s.stride;
    int32_t episode = *WEDM_ROW(s.i32, WEDM_I_EPISODE);
    int32_t klo = *WEDM_ROW(s.i32, WEDM_I_KEY_LO), khi = *WEDM_ROW(s.i32, WEDM_I_KEY_HI);
    for (int f = 0; f < WEDM_F64_COUNT; ++f) *WEDM_ROW(s.f64, f) = 0.0;
    for (int f = 0; f < WEDM_I32_COUNT; ++f) *WEDM_ROW(s.i32, f) = 0;
    for (int f = 0; f < WEDM_I8_COUNT; ++f) *WEDM_ROW(s.i8, f) = 0;
    if (reseed) {
        *WEDM_ROW(s.i32, WEDM_I_EPISODE) = 0;
        *WEDM_ROW(s.i32, WEDM_I_KEY_LO) = (int32_t)key_lo;
        *WEDM_ROW(s.i32, WEDM_I_KEY_HI) = (int32_t)key_hi;
    } else {
        *WEDM_ROW(s.i32, WEDM_I_EPISODE) = episode + 1;
        *WEDM_ROW(s.i32, WEDM_I_KEY_LO) = klo;
        *WEDM_ROW(s.i32, WEDM_I_KEY_HI) = khi;
    }
    *WEDM_ROW(s.f64, WEDM_F_WORKPIECE_POS) = p.initial_gap;            // wire_edm.py:111
    *WEDM_ROW(s.f64, WEDM_F_TARGET_POS) = p.target_cutting_distance;   // wire_edm.py:112
    *WEDM_ROW(s.f64, WEDM_F_UNWIND_VEL) = 0.2;                         // state.py:55
    *WEDM_ROW(s.f64, WEDM_F_SPARK_Y) = __builtin_nan("");              // [0, None, 0]
    *WEDM_ROW(s.f64, WEDM_F_LAST_GAP) = -1.0;                          // dielectric.py:78
    *WEDM_ROW(s.f64, WEDM_F_LAST_DENSITY) = -1.0;                      // dielectric.py:79
    const float spool = (float)p.spool_T;
    *WEDM_ROW(s.f64, WEDM_F_TMAX) = (double)spool;
    for (int i = 0; i < n_seg_max; ++i) s.T[(int64_t)i * stride + e] = spool;  // wire.py:264-269
    if (s.obs)
        for (int c = 0; c < p.obs_dim; ++c) s.obs[(int64_t)c * stride + e] = 0.0f;
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
        case 6: {  // a = seed-as-double bits are not needed: x = time, y = env id, key fixed
            U2 u = philox_pair(0x12345678u, 0x9abcdef0u, (uint32_t)x, 3u, (uint32_t)y, 1u);
            r = u.a + 2.0 * u.b;  // both words observable: u.a, u.b in [0,1)
            break;
        }
        case 7: r = philox_std_normal(0x12345678u, 0x9abcdef0u, (uint32_t)x, 3u, (uint32_t)y); break;
        default: break;
    }
    out[i] = r;
}

// =================================================================== C-ABI
struct wedm_ctx {
    wedm_params p;
    int32_t num_envs = 0, n_seg_max = 0;
    int device = -1;
    bool bound = false, geom_bound = false;
    wedm_state_ptrs s{};
    wedm_geom_ptrs g{};
    void* tables_dev = nullptr;
    Tables tb{};
    int32_t variant = 0;
    int lds_limit = 0;
    std::string err;
    std::string last_kernel;
};

static int32_t fail(wedm_ctx* ctx, int32_t code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}
static int32_t hip_fail(wedm_ctx* ctx, hipError_t e, const char* what) {
    return fail(ctx, WEDM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

static thread_local std::string g_create_error;

extern "C" {

int32_t wedm_abi_version(void) { return WEDM_ABI_VERSION; }
int64_t wedm_sizeof_params(void) { return (int64_t)sizeof(wedm_params); }

const char* wedm_last_error(wedm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
const char* wedm_last_kernel(wedm_ctx* ctx) { return ctx ? ctx->last_kernel.c_str() : ""; }

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
        hipFree(ctx->tables_dev);
        delete ctx;
        return WEDM_ERR_HIP;
    }
    const double* d = (const double*)ctx->tables_dev;
    ctx->tb.mode_current = d;
    ctx->tb.crater_mean = d + n;
    ctx->tb.crater_std = d + 2 * n;
    ctx->tb.crater_depth = d + 3 * n;
    ctx->tb.crater_valid = (const int32_t*)(d + 4 * n);
    *out = ctx;
    return WEDM_OK;
}

int32_t wedm_destroy(wedm_ctx* ctx) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (ctx->tables_dev) hipFree(ctx->tables_dev);
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
    return WEDM_OK;
}

int32_t wedm_bind_geometry(wedm_ctx* ctx, const wedm_geom_ptrs* geom) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (!geom || !geom->f64 || !geom->i32) return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_bind_geometry: null geometry block");
    ctx->g = *geom;
    ctx->geom_bound = true;
    return WEDM_OK;
}

int32_t wedm_set_kernel(wedm_ctx* ctx, int32_t variant) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (variant < 0 || variant > 2) return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_set_kernel: variant must be 0, 1 or 2");
    ctx->variant = variant;
    return WEDM_OK;
}

int32_t wedm_reset(wedm_ctx* ctx, const uint8_t* mask, uint64_t seed, int32_t reseed, void* stream) {
    if (!ctx) return WEDM_ERR_BAD_ARG;
    if (!ctx->bound) return fail(ctx, WEDM_ERR_NOT_BOUND, "wedm_reset: call wedm_bind_state first");
    const int block = 256;
    const int grid = (ctx->num_envs + block - 1) / block;
    hipLaunchKernelGGL(wedm_reset_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, ctx->p, ctx->s,
                       ctx->num_envs, ctx->n_seg_max, mask, (uint32_t)seed, (uint32_t)(seed >> 32), reseed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(ctx, e, "wedm_reset launch");
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

    KArgs k;
    k.p = ctx->p;
    k.s = ctx->s;
    k.g = ctx->g;
    k.a = *action;
    k.tb = ctx->tb;
    k.num_envs = ctx->num_envs;
    k.n_substeps = n_substeps;
    k.n_seg_max = ctx->n_seg_max;

    const size_t lds_bytes = (size_t)ctx->n_seg_max * 64 * sizeof(float);
    int variant = ctx->variant;
    if (variant == 0) variant = (n_substeps > 1 && lds_bytes <= (size_t)ctx->lds_limit) ? 2 : 1;
    if (variant == 2 && lds_bytes > (size_t)ctx->lds_limit)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: n_seg_max * 256 B exceeds the LDS a workgroup can take");

    char name[128];
    if (variant == 1) {
        const int block = 256;
        const int grid = (ctx->num_envs + block - 1) / block;
        hipLaunchKernelGGL(wedm_step_global, dim3(grid), dim3(block), 0, (hipStream_t)stream, k);
        std::snprintf(name, sizeof(name), "wedm_step_global<<<%d,%d>>> n_sub=%d", grid, block, n_substeps);
    } else {
        const int grid = (ctx->num_envs + 63) / 64;
        hipError_t ea = hipFuncSetAttribute((const void*)wedm_step_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds_bytes);
        if (ea != hipSuccess) return hip_fail(ctx, ea, "hipFuncSetAttribute(wedm_step_lds)");
        hipLaunchKernelGGL(wedm_step_lds, dim3(grid), dim3(64), lds_bytes, (hipStream_t)stream, k);
        std::snprintf(name, sizeof(name), "wedm_step_lds<<<%d,64,%zuB>>> n_sub=%d", grid, lds_bytes, n_substeps);
    }
    ctx->last_kernel = name;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(ctx, e, "wedm_step launch");
    return WEDM_OK;
}

int32_t wedm_debug_math(int32_t kind, const double* a, const double* b, double* out, int32_t n, void* stream) {
    if (!a || !out || n <= 0 || kind < 0 || kind > 7) return WEDM_ERR_BAD_ARG;
    hipLaunchKernelGGL(wedm_debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, kind, a, b,
                       out, n);
    return hipGetLastError() == hipSuccess ? WEDM_OK : WEDM_ERR_HIP;
}

}  // extern "C"
