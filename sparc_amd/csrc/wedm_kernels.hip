// wedm_kernels.hip — the translation unit(s) of libwedm_hip.so: includes the kernel families, instantiates them (one family per
// WEDM_PART, compiled in parallel), and holds the host side of the C-ABI of include/wedm_hip.h (launch plan, wedm_create ...).
//
// Device code, by file (DESIGN.md section 4 has the table with what binds each kernel):
//   wedm_device.h         per-lane physics of one microsecond: Env, prelude (quiet / general), epilogue (monitor + motion),
//                         Philox, the portable exp / log / cube, stencil_cell
//   wedm_common.h         build switches, WalkTable, KArgs, trace point, wire accessors / copy_wire, tile_staged / quad_staged
//   wedm_k_global_split.h wedm_step_global (in place in global memory; stencil_mode 1, injected variates, very long wires),
//                         wedm_step_split (single microseconds where the stream kernel does not fit)
//   wedm_k_stream.h       wedm_step_stream<L>: single microseconds (the reference's step() cadence), uniform geometry
//   wedm_k_lanes.h        wedm_step_lanes<L>: any geometry, cell by cell (kernel 10; stencil_mode 1)
//   wedm_lanes2.h         wedm_step_lanes_pk<L>: any geometry, packed float32 walk (kernel 2: BASELINE config 5), and its served form
//   wedm_k_fused.h        wedm_step_fused<L>: uniform geometry, wire chunks in LDS, wave-uniform tile table
//   wedm_k_packed.h       wedm_step_packed<L>: the same with two virtual chunks per lane in float2 registers
//   wedm_served.h         wedm_step_served<L>: the packed walk on three waves of a block, the scalar physics of the block's
//                         environments on the fourth, one microsecond ahead (kernel 9: large batches of long wires)
//   wedm_k_regs.h         wedm_step_regs<128, L> (the headline: the wire in the registers of two lanes per environment),
//                         wedm_step_regs_wide<16, L> (4 / 8 / 16 lanes of a DPP row per environment: small batches)
// The wire block is quad-interleaved, T[seg >> 2][env][seg & 3] (include/wedm_hip.h, ABI v4): a lane that owns a run of
// segments of one environment moves it with global_load / store_dwordx4, a wavefront still touches contiguous 1-KB runs.
// Kernels exist in several instantiations (signal trace point, FROZEN_OK for autoreset handles, N1 / EXTRA for tile tables
// with one-change tiles or short tails): code that costs the other launches 1-2 % by its mere presence lives in its own
// instantiation, chosen per handle in plan_launch().
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

#include "wedm_common.h"
#include "wedm_k_global_split.h"
#include "wedm_k_lanes.h"
#include "wedm_k_fused.h"
#include "wedm_k_stream.h"
#include "wedm_k_regs.h"
#include "wedm_k_packed.h"
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
#define WEDM_INST_LANES_PK_F64(L, tr) template __global__ void wedm_step_lanes_pk<L, tr, true>(const KArgs);
#define WEDM_EXT_LANES_PK_F64(L, tr) extern template __global__ void wedm_step_lanes_pk<L, tr, true>(const KArgs);
// stencil_mode 1 on the stream kernel: the single-microsecond instantiation <L, false, 64, ONE = true, F64 = true>
#define WEDM_STREAM_F64_LIST(X) X(1) X(2) X(4) X(8) X(16)
#define WEDM_INST_STREAM_F64(L) template __global__ void wedm_step_stream<L, false, 64, true, true>(const KArgs);
#define WEDM_EXT_STREAM_F64(L) extern template __global__ void wedm_step_stream<L, false, 64, true, true>(const KArgs);
#define WEDM_LANES_SERVED_LIST(X) X(4) X(8) X(16)
#define WEDM_INST_LANES_SERVED(L) template __global__ void wedm_step_lanes_served<L>(const KArgs);
#define WEDM_EXT_LANES_SERVED(L) extern template __global__ void wedm_step_lanes_served<L>(const KArgs);
#define WEDM_INST_REGS_SERVED template __global__ void wedm_step_regs_served<128>(const KArgs);
#define WEDM_EXT_REGS_SERVED extern template __global__ void wedm_step_regs_served<128>(const KArgs);
// stencil_mode 1 on the register kernel: <128, L, TRACE, F64 = true>
#define WEDM_REGS_F64_LIST(X) X(1, false) X(1, true) X(2, false) X(2, true)
#define WEDM_INST_REGS_F64(L, tr) template __global__ void wedm_step_regs<128, L, tr, true>(const KArgs);
#define WEDM_EXT_REGS_F64(L, tr) extern template __global__ void wedm_step_regs<128, L, tr, true>(const KArgs);
// stencil_mode 1 on the wide register kernel: <16, L, CUT, TRACE, F64 = true, MINB> (a traced launch runs the CUT form, as in
// float32; MINB = 1 for a batch of one wave per SIMD, 2 beyond)
#define WEDM_WIDE_F64_LIST1(X, mb) X(4, false, false, mb) X(4, true, false, mb) X(4, true, true, mb) X(8, false, false, mb) X(8, true, false, mb) \
                                   X(8, true, true, mb) X(16, false, false, mb) X(16, true, false, mb) X(16, true, true, mb)
#define WEDM_WIDE_F64_LIST(X) WEDM_WIDE_F64_LIST1(X, 1) WEDM_WIDE_F64_LIST1(X, 2)
#define WEDM_INST_WIDE_F64(L, cut, tr, mb) template __global__ void wedm_step_regs_wide<16, L, cut, tr, true, mb>(const KArgs);
#define WEDM_EXT_WIDE_F64(L, cut, tr, mb) extern template __global__ void wedm_step_regs_wide<16, L, cut, tr, true, mb>(const KArgs);
#define WEDM_INST_SERVED(L, ex) template __global__ void wedm_step_served<L, ex>(const KArgs);
#define WEDM_EXT_SERVED(L, ex) extern template __global__ void wedm_step_served<L, ex>(const KArgs);
#if defined(WEDM_PART) && WEDM_PART == 1
WEDM_PACKED_LIST(WEDM_INST_PACKED)
#elif defined(WEDM_PART) && WEDM_PART == 2
WEDM_FUSED_LIST(WEDM_INST_FUSED)
WEDM_FUSED_F64_LIST(WEDM_INST_FUSED_F64)
#elif defined(WEDM_PART) && WEDM_PART == 3
WEDM_SERVED_LIST(WEDM_INST_SERVED)
WEDM_INST_REGS_SERVED
WEDM_LANES_PK_LIST(WEDM_INST_LANES_PK)
WEDM_LANES_SERVED_LIST(WEDM_INST_LANES_SERVED)
#elif defined(WEDM_PART) && WEDM_PART == 4
WEDM_REGS_F64_LIST(WEDM_INST_REGS_F64)
WEDM_WIDE_F64_LIST(WEDM_INST_WIDE_F64)
WEDM_LANES_PK_LIST(WEDM_INST_LANES_PK_F64)
WEDM_STREAM_F64_LIST(WEDM_INST_STREAM_F64)
#else
#if defined(WEDM_PART)
WEDM_PACKED_LIST(WEDM_EXT_PACKED)
WEDM_FUSED_LIST(WEDM_EXT_FUSED)
WEDM_FUSED_F64_LIST(WEDM_EXT_FUSED_F64)
WEDM_SERVED_LIST(WEDM_EXT_SERVED)
WEDM_EXT_REGS_SERVED
WEDM_REGS_F64_LIST(WEDM_EXT_REGS_F64)
WEDM_WIDE_F64_LIST(WEDM_EXT_WIDE_F64)
WEDM_LANES_PK_LIST(WEDM_EXT_LANES_PK)
WEDM_LANES_PK_LIST(WEDM_EXT_LANES_PK_F64)
WEDM_STREAM_F64_LIST(WEDM_EXT_STREAM_F64)
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
    uint32_t walk4_kind_s[5] = {0u, 0u, 0u, 0u, 0u};  // tiles with several flag changes (the stream kernel's register walk has no code for them)
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
template <bool TR, int CMAX, bool ONE = false, bool F64 = false> static const void* pick_stream(int L) {
    switch (L) {
        case 1: return (const void*)wedm_step_stream<1, TR, CMAX, ONE, F64>;
        case 2: return (const void*)wedm_step_stream<2, TR, CMAX, ONE, F64>;
        case 4: return (const void*)wedm_step_stream<4, TR, CMAX, ONE, F64>;
        case 8: return (const void*)wedm_step_stream<8, TR, CMAX, ONE, F64>;
        default: return (const void*)wedm_step_stream<16, TR, CMAX, ONE, F64>;
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

template <bool TR, bool F64 = false> static const void* pick_lanes_pk(int L) {
    switch (L) {
        case 1: return (const void*)wedm_step_lanes_pk<1, TR, F64>;
        case 2: return (const void*)wedm_step_lanes_pk<2, TR, F64>;
        case 4: return (const void*)wedm_step_lanes_pk<4, TR, F64>;
        case 8: return (const void*)wedm_step_lanes_pk<8, TR, F64>;
        default: return (const void*)wedm_step_lanes_pk<16, TR, F64>;
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
    // ... and its packed form (wedm_step_lanes_pk, either typing of the stencil): two virtual chunks of ceil(n_seg_max / 2L) cells per lane
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
    // Numba's typing of the stencil: the register kernels (uniform geometry; at most 128 / 512 segments), the fused tile walk
    // (uniform geometry), the predicated LDS kernel (any geometry), or in place in global memory; no packed LDS form, no served
    // form, no stream / split kernel
    if (f64 && variant != 0 && variant != 1 && variant != 2 && variant != 3 && variant != 6 && variant != 7 && variant != 8 && variant != 10)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: stencil_mode 1 (float64 stencil expressions) runs on kernels 1, 2 (10), 3, 6 (single microseconds without a trace sample), 7 and 8 only");
    // the stream kernel in that typing: its single-microsecond instantiation only (chunks of at most 64 cells, no trace sample)
    const bool stream_f64_ok = stream_ok && single && !tr && WEDM_STREAM_REGWALK && ctx->walk4_C[lanes_index(slanes)] <= 64;
    if (f64 && variant == 6 && !stream_f64_ok)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: under stencil_mode 1 the stream kernel runs launches of one microsecond without a trace sample, chunks of at most 64 cells");
    // (by itself where the float32 launch takes it too and every tile of the table has register-walk code)
    if (f64 && variant == 0 && stream_f64_ok && stream_auto && ctx->walk4_kind_s[lanes_index(slanes)] == 0u) variant = 6;
    // kernel 2 is the packed form where it applies (no injected variates; under stencil_mode 1 the same walk with float64-typed
    // cells); kernel 10 names the cell-by-cell form explicitly (A/B timing, tests)
    const bool use_pk = !ctx->replay && lanes_pk_ok;  // (both typings of the stencil)
    // kernel 8 (wide register kernel): 4, 8 or 16 lanes per environment (the fewest that hold the wire), 32 cells each in
    // registers; uniform geometry, either typing of the stencil, at most 512 segments.  Chosen by itself for a batch
    // that one round of blocks covers at one wave per
    // SIMD: such a launch is one wave's dependent chain per microsecond whatever the kernel, and this one's is the
    // shortest (measured, 4 096 x 400 and 16 384 x 128: DESIGN.md 4.1b)
    const int wl_min = P.n_seg <= 128 ? 4 : P.n_seg <= 256 ? 8 : 16;
    const int wl = (ctx->lanes == 4 || ctx->lanes == 8 || ctx->lanes == 16) ? ctx->lanes : wl_min;
    const bool wide_ok = uniform && P.n_seg >= 9 && P.n_seg <= 512 && !ctx->replay &&
                         (ctx->lanes == 0 || (wl == ctx->lanes && wl >= wl_min));
    // (stencil_mode 1, a short wire in a tiny batch: 256 waves of this kernel over 4 lanes against 1 024 of the tile walk over 16 --
    // 2.81 against 2.47 ms at 4 096 x 128, profiles/r4/plan_sweep_f64.txt)
    const bool f64_tiny = f64 && P.n_seg <= 128 && ctx->num_envs <= 4096;
    if (variant == 0 && !single && wide_ok && !f64_tiny && ctx->lanes == 0 &&
        (int64_t)ctx->num_envs * wl <= (int64_t)WEDM_WIDE_AUTO_MAX_LANES)
        variant = 8;
    if (variant == 8 && !wide_ok)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: wide register kernel needs uniform geometry, 9 to 512 segments and lanes 0, 4, 8 or 16 with 32 cells per lane covering the wire");
    // kernel 7 (register kernel): one or two lanes per environment with the wire in their registers; wires of at most 128
    // segments, uniform geometry, either typing of the stencil; a launch with a trace sample runs its TRACE instantiation
    const bool regs_ok = uniform && ctx->walk_regs_ok && ctx->n_seg_max <= 128 && !ctx->replay;
    if (variant == 0) {
        // fused launches of a batch that gives most CUs a block of the register kernel (measured, 128 segments, two lanes
        // per environment against the best LDS kernel: 8 192 environments 2.8e9 vs 3.5e9, 16 384: 5.5e9 vs 6.1e9,
        // 24 576: 8.3e9 vs 7.4e9, 32 768: 1.10e10 vs 9.9e9, 65 536: 1.67e10 vs 1.44e10, 131 072: 1.76e10 vs 1.50e10;
        // up to 16 384 environments the wide register kernel above has taken the launch: 8.1e9 there)
        // (stencil_mode 1: single microseconds too -- 34 us against the cell-by-cell LDS kernel's 44 at 65 536 x 128)
        if ((!single || f64) && regs_ok && ctx->lanes == 0 && ctx->num_envs >= 20480) variant = 7;
        // stencil_mode 1, longer wires, any batch: the wide register kernel (two blocks per CU beyond one wave per SIMD) -- 18 - 20
        // float64 operations per cell leave the LDS round trips of the tile walk nothing to hide behind (32 768 x 400: 1.93e9
        // against the fused kernel's 1.49e9; 8 192 x 400: 1.5e9 against 1.2e9 already at one block per CU)
        if (variant == 0 && f64 && !single && wide_ok && !f64_tiny && ctx->lanes == 0) variant = 8;
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
        if (f64) variant = (!single && fused_ok) ? 3 : ((lanes_ok || use_pk) ? 2 : 1);
        else if (single) variant = (stream_ok && stream_auto) ? 6 : 5;
        else if (packed_ok && (ctx->auto_prefers_packed || !fused_ok)) variant = 4;
        else if (fused_ok) variant = 3;
        else variant = (lanes_ok || use_pk) ? 2 : 1;
    }
    if (variant == 9 && !served_ok)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: served kernel needs uniform geometry, the float32 stencil, lanes 4 or 8, two chunks that fit in LDS and freeze_terminated");
    if (variant == 9 && tr) variant = packed_ok ? 4 : fused_ok ? 3 : (lanes_ok || use_pk) ? 2 : 1;
    // kernel 12 (served register kernel): the register kernel's conditions + what the served scalar wave does not do
    if (variant == 12 && (!regs_ok || f64 || tr || P.keep_stepping_terminated))
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: served register kernel needs uniform geometry, at most 128 segments, the float32 stencil, no trace sample in the launch and freeze_terminated");
    if (variant == 7 && !regs_ok)
        return fail(ctx, WEDM_ERR_UNSUPPORTED, "wedm_step: register kernel needs uniform geometry and at most 128 segments");
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
        fn = f64 ? (tr ? (rl == 1 ? (const void*)wedm_step_regs<128, 1, true, true> : (const void*)wedm_step_regs<128, 2, true, true>)
                       : (rl == 1 ? (const void*)wedm_step_regs<128, 1, false, true> : (const void*)wedm_step_regs<128, 2, false, true>))
           : tr ? (rl == 1 ? (const void*)wedm_step_regs<128, 1, true> : (const void*)wedm_step_regs<128, 2, true>)
                : (rl == 1 ? (const void*)wedm_step_regs<128, 1> : (const void*)wedm_step_regs<128, 2>);
        std::snprintf(out.name, sizeof(out.name), "wedm_step_regs<%d>%s<<<%d,256>>>", rl, f64 ? "[f64 stencil]" : "", grid);
    } else if (variant == 8) {
        grid = (ctx->num_envs + 256 / wl - 1) / (256 / wl);
#define WEDM_PICK_WIDE(...) (wl == 4 ? (const void*)wedm_step_regs_wide<16, 4, __VA_ARGS__> : wl == 8 ? (const void*)wedm_step_regs_wide<16, 8, __VA_ARGS__> \
                                                                                                  : (const void*)wedm_step_regs_wide<16, 16, __VA_ARGS__>)
        const bool cutw = (P.n_seg & 7) != 0;
        // (stencil_mode 1: a batch of more than one wave per SIMD runs the two-blocks-per-CU instantiation -- 32 768 x 400: 1.93e9
        // against 1.52e9; 4 096 x 400: 1.33e9 against 1.45e9)
        const bool two = (int64_t)ctx->num_envs * wl > (int64_t)WEDM_WIDE_AUTO_MAX_LANES;
        fn = f64 ? (two ? (tr ? WEDM_PICK_WIDE(true, true, true, 2) : cutw ? WEDM_PICK_WIDE(true, false, true, 2) : WEDM_PICK_WIDE(false, false, true, 2))
                        : (tr ? WEDM_PICK_WIDE(true, true, true, 1) : cutw ? WEDM_PICK_WIDE(true, false, true, 1) : WEDM_PICK_WIDE(false, false, true, 1)))
                 : (tr ? WEDM_PICK_WIDE(true, true) : cutw ? WEDM_PICK_WIDE(true) : WEDM_PICK_WIDE(false));
#undef WEDM_PICK_WIDE
        std::snprintf(out.name, sizeof(out.name), "wedm_step_regs_wide<%d>%s<<<%d,256>>>", wl, f64 ? "[f64 stencil]" : "", grid);
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
        fn = f64 ? pick_stream<false, 64, true, true>(slanes)
           : one ? pick_stream<false, 64, true>(slanes)
           : ctx->walk4_C[sli] <= 64 ? (tr ? pick_stream<true, 64>(slanes) : pick_stream<false, 64>(slanes))
                                     : (tr ? pick_stream<true, 104>(slanes) : pick_stream<false, 104>(slanes));
        std::snprintf(out.name, sizeof(out.name), "wedm_step_stream<%d>%s<<<%d,256,%zuB>>>", slanes, f64 ? "[f64 stencil]" : "", grid, fl);
    } else if (lanes_sv_ok && variant == 11) {  // (by name only: at 16 384 environments x <= 450 segments it measures 2.39e9 against the packed form's 2.48e9 - 2.62e9)
        grid = (ctx->num_envs + 192 / svgl - 1) / (192 / svgl);
        fl = (2 * (size_t)((ctx->n_seg_max + 2 * svgl - 1) / (2 * svgl)) + 2) * 768 + (svgl == 4 ? sizeof(ServedBox<48>) : svgl == 8 ? sizeof(ServedBox<24>) : sizeof(ServedBox<12>));
        fn = svgl == 4 ? (const void*)wedm_step_lanes_served<4> : svgl == 8 ? (const void*)wedm_step_lanes_served<8> : (const void*)wedm_step_lanes_served<16>;
        std::snprintf(out.name, sizeof(out.name), "wedm_step_lanes_served<%d><<<%d,256,%zuB>>>", svgl, grid, fl);
    } else if ((variant == 2 || variant == 11) && use_pk) {
        grid = (ctx->num_envs + 256 / pklanes - 1) / (256 / pklanes);
        fl = (2 * (size_t)((ctx->n_seg_max + 2 * pklanes - 1) / (2 * pklanes)) + 2) * 1024;
        fn = f64 ? (tr ? pick_lanes_pk<true, true>(pklanes) : pick_lanes_pk<false, true>(pklanes))
                 : (tr ? pick_lanes_pk<true>(pklanes) : pick_lanes_pk<false>(pklanes));
        std::snprintf(out.name, sizeof(out.name), "wedm_step_lanes_pk<%d>%s<<<%d,256,%zuB>>>", pklanes, f64 ? "[f64 stencil]" : "", grid, fl);
    } else if (variant == 2 || variant == 10) {
        grid = (ctx->num_envs + 256 / glanes - 1) / (256 / glanes);
        fl = (size_t)((ctx->n_seg_max + glanes - 1) / glanes) * 1024;
        fn = f64 ? (tr ? pick_lanes<true, true>(glanes) : pick_lanes<false, true>(glanes))
                 : (tr ? pick_lanes<true, false>(glanes) : pick_lanes<false, false>(glanes));
        std::snprintf(out.name, sizeof(out.name), "wedm_step_lanes<%d>%s<<<%d,256,%zuB>>>", glanes, f64 ? "[f64 stencil]" : "", grid, fl);
    } else if (variant == 12) {
        grid = (ctx->num_envs + 63) / 64;
        fl = sizeof(ServedBox<64>);
        out.walk = ctx->walk_dev + 11;  // four chunks of 32 cells
        out.block = 192;                // two walker waves + the scalar wave
        fn = (const void*)wedm_step_regs_served<128>;
        std::snprintf(out.name, sizeof(out.name), "wedm_step_regs_served<<<%d,192,%zuB>>>", grid, fl);
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
    if (variant != 9 && variant != 12) out.block = 256;
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
            ctx->walk4_kind_s[i] = host_tabs[5 + i].kind_s_mask;
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
    if (variant < 0 || variant > 12) return fail(ctx, WEDM_ERR_BAD_ARG, "wedm_set_kernel: variant must be 0..12");
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

