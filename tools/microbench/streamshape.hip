// Microbenchmark (diagnostic): the MEMORY SHAPE of the single-microsecond stream kernel without its physics --
// L = 2 lanes per environment, 256-thread blocks, every lane requests the state rows a microsecond reads and its
// 64-cell chunk of the wire, then stores what a microsecond changes.  Variants:
//   A  state rows field-major (one row per load), wire T[seg][env]           (ABI v3: 42 + 64 loads, 25 + 64 stores)
//   B  state as A, wire quad-interleaved T[seg/4][env][4] (dwordx4)           (42 + 16 loads, 25 + 16 stores)
//   C  wire as B, state rows packed 16 B per environment (f64 pairs, i32 quads, i8 octet)  (16 + 16 loads, 10 + 16 stores)
//   wire-only / state-only splits of A and B/C.
//   hipcc --offload-arch=gfx950 -O3 -o streamshape streamshape.hip && ./streamshape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int NF = 26, NI = 14, NB = 8;      // rows per block (i8 padded to 8)
constexpr int LF = 21, LI = 14, LB = 5;      // rows a microsecond reads
constexpr int SF = 12, SI = 7, SB = 6;       // rows a microsecond writes

struct Ptrs { double* f64; int32_t* i32; int8_t* i8; float* T; int stride; int n_seg; };

// WIRE_FIRST: the wire words are requested before the state rows.  INDEP: the wire stores do not wait for the state.
// DELAY: dependent f32 operations between the loads and the stores (stands in for the microsecond's arithmetic).
template <bool WIRE, bool STATE, bool QUAD, bool WIRE_FIRST = false, bool INDEP = false, int DELAY = 0, int BLOCK = 256>
__global__ void __launch_bounds__(BLOCK, (BLOCK >= 512 ? 1 : 512 / BLOCK)) k_rows(Ptrs p) {
    extern __shared__ float dyn_lds[];
    if (p.n_seg < 0) dyn_lds[threadIdx.x] = 0.0f;  // keeps the dynamic allocation alive
    const int tid = threadIdx.x, el = tid >> 1, c = tid & 1;
    const int e = blockIdx.x * (BLOCK / 2) + el;
    const int stride = p.stride;
    double f[LF]; int32_t iv[LI]; int32_t bv[LB];
    float w[64];
    const int cbase = c * 64;
    if (WIRE && WIRE_FIRST) {
        const float4* T4 = (const float4*)p.T;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 v = T4[(size_t)(cbase / 4 + q) * stride + e];
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    }
    if (STATE) {
#pragma unroll
        for (int r = 0; r < LF; ++r) f[r] = p.f64[(size_t)r * stride + e];
#pragma unroll
        for (int r = 0; r < LI; ++r) iv[r] = p.i32[(size_t)r * stride + e];
#pragma unroll
        for (int r = 0; r < LB; ++r) bv[r] = p.i8[(size_t)r * stride + e];
    }
    if (WIRE && !WIRE_FIRST) {
        if (QUAD) {
            const float4* T4 = (const float4*)p.T;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float4 v = T4[(size_t)(cbase / 4 + q) * stride + e];
                w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 64; ++j) w[j] = p.T[(size_t)(cbase + j) * stride + e];
        }
    }
    double acc = 0.0; int32_t ia = 0;
    if (STATE) {
#pragma unroll
        for (int r = 0; r < LF; ++r) acc += f[r];
#pragma unroll
        for (int r = 0; r < LI; ++r) ia += iv[r];
#pragma unroll
        for (int r = 0; r < LB; ++r) ia += bv[r];
    }
    float add = INDEP ? 0.5f : (float)(acc * 1e-30) + (float)ia * 1e-30f + 0.5f;
    if (DELAY) {
        float z = add;
#pragma unroll 1
        for (int it = 0; it < DELAY; ++it) { z = z * 1.000001f + 1e-9f; asm volatile("" : "+v"(z)); }
        add = z;
    }
    if (WIRE) {
        if (QUAD) {
            float4* T4 = (float4*)p.T;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float4 v;
                v.x = w[4 * q] * 1.0001f + add; v.y = w[4 * q + 1] * 1.0001f + add;
                v.z = w[4 * q + 2] * 1.0001f + add; v.w = w[4 * q + 3] * 1.0001f + add;
                T4[(size_t)(cbase / 4 + q) * stride + e] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 64; ++j) p.T[(size_t)(cbase + j) * stride + e] = w[j] * 1.0001f + add;
        }
    }
    if (STATE && c == 0) {
#pragma unroll
        for (int r = 0; r < SF; ++r) p.f64[(size_t)r * stride + e] = f[r] + 1.0;
#pragma unroll
        for (int r = 0; r < SI; ++r) p.i32[(size_t)r * stride + e] = iv[r] + 1;
#pragma unroll
        for (int r = 0; r < SB; ++r) p.i8[(size_t)r * stride + e] = (int8_t)(bv[r < LB ? r : 0] + 1);
    }
    if (!WIRE && !STATE) p.T[e] = add;
}

// C: packed state, quad wire
template <bool WIRE, bool STATE, bool ONE_LANE>
__global__ void __launch_bounds__(256, 2) k_packed(Ptrs p) {
    const int tid = threadIdx.x, el = tid >> 1, c = tid & 1;
    const int e = blockIdx.x * 128 + el;
    const int stride = p.stride;
    double2 f[(LF + 1) / 2]; int4 iv[(LI + 3) / 4]; int2 bv;
    if (STATE && (!ONE_LANE || c == 0)) {
        const double2* F2 = (const double2*)p.f64;
        const int4* I4 = (const int4*)p.i32;
        const int2* B8 = (const int2*)p.i8;
#pragma unroll
        for (int r = 0; r < (LF + 1) / 2; ++r) f[r] = F2[(size_t)r * stride + e];
#pragma unroll
        for (int r = 0; r < (LI + 3) / 4; ++r) iv[r] = I4[(size_t)r * stride + e];
        bv = B8[e];
    }
    float w[64];
    const int cbase = c * 64;
    if (WIRE) {
        const float4* T4 = (const float4*)p.T;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 v = T4[(size_t)(cbase / 4 + q) * stride + e];
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    }
    double acc = 0.0; int32_t ia = 0;
    if (STATE) {
        if (ONE_LANE) {  // partner lane copies through DPP-able shuffles
#pragma unroll
            for (int r = 0; r < (LF + 1) / 2; ++r) { f[r].x = __shfl(f[r].x, tid & ~1, 64); f[r].y = __shfl(f[r].y, tid & ~1, 64); }
#pragma unroll
            for (int r = 0; r < (LI + 3) / 4; ++r) { iv[r].x = __shfl(iv[r].x, tid & ~1, 64); iv[r].y = __shfl(iv[r].y, tid & ~1, 64);
                                                     iv[r].z = __shfl(iv[r].z, tid & ~1, 64); iv[r].w = __shfl(iv[r].w, tid & ~1, 64); }
            bv.x = __shfl(bv.x, tid & ~1, 64); bv.y = __shfl(bv.y, tid & ~1, 64);
        }
#pragma unroll
        for (int r = 0; r < (LF + 1) / 2; ++r) acc += f[r].x + f[r].y;
#pragma unroll
        for (int r = 0; r < (LI + 3) / 4; ++r) ia += iv[r].x + iv[r].y + iv[r].z + iv[r].w;
        ia += bv.x + bv.y;
    }
    const float add = (float)(acc * 1e-30) + (float)ia * 1e-30f + 0.5f;
    if (WIRE) {
        float4* T4 = (float4*)p.T;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float4 v;
            v.x = w[4 * q] * 1.0001f + add; v.y = w[4 * q + 1] * 1.0001f + add;
            v.z = w[4 * q + 2] * 1.0001f + add; v.w = w[4 * q + 3] * 1.0001f + add;
            T4[(size_t)(cbase / 4 + q) * stride + e] = v;
        }
    }
    if (STATE && c == 0) {
        double2* F2 = (double2*)p.f64;
        int4* I4 = (int4*)p.i32;
        int2* B8 = (int2*)p.i8;
#pragma unroll
        for (int r = 0; r < SF / 2; ++r) { double2 v = f[r]; v.x += 1.0; v.y += 1.0; F2[(size_t)r * stride + e] = v; }
#pragma unroll
        for (int r = 0; r < (SI + 3) / 4; ++r) { int4 v = iv[r]; v.x += 1; I4[(size_t)r * stride + e] = v; }
        bv.x += 1; B8[e] = bv;
    }
    if (!WIRE && !STATE) p.T[e] = add;
}

template <class F> float time_it(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int N = 65536, S = 128;
    Ptrs p; p.stride = N; p.n_seg = S;
    CK(hipMalloc(&p.f64, (size_t)NF * N * 8)); CK(hipMalloc(&p.i32, (size_t)16 * N * 4)); CK(hipMalloc(&p.i8, (size_t)NB * N));
    CK(hipMalloc(&p.T, (size_t)S * N * 4));
    CK(hipMemset(p.f64, 0, (size_t)NF * N * 8)); CK(hipMemset(p.i32, 0, (size_t)16 * N * 4)); CK(hipMemset(p.i8, 0, (size_t)NB * N));
    CK(hipMemset(p.T, 0, (size_t)S * N * 4));
    const dim3 g(N / 128), b(256);
    CK(hipFuncSetAttribute((const void*)k_rows<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_rows<true, true, true, false, false, 0, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto rep = [&](const char* nm, float ms) { printf("  %-66s %8.2f us\n", nm, ms * 1e3); };
    printf("N=%d S=%d, L=2 lanes per environment, %d blocks x 256\n", N, S, N / 128);
    rep("A  rows: 40 state loads + 64 wire dword loads, 25 + 64 stores", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, false>), g, b, 0, 0, p); }, 300));
    rep("A  wire only (64 dword loads, 64 stores)", time_it([&] { hipLaunchKernelGGL((k_rows<true, false, false>), g, b, 0, 0, p); }, 300));
    rep("A  state only (40 loads, 25 stores)", time_it([&] { hipLaunchKernelGGL((k_rows<false, true, false>), g, b, 0, 0, p); }, 300));
    rep("B  rows state + quad wire (40 + 16 x4 loads, 25 + 16 x4 stores)", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true>), g, b, 0, 0, p); }, 300));
    rep("B  quad wire only (16 x4 loads, 16 x4 stores)", time_it([&] { hipLaunchKernelGGL((k_rows<true, false, true>), g, b, 0, 0, p); }, 300));
    rep("C  packed state + quad wire (16 + 16 x4 loads, 10 + 16 stores)", time_it([&] { hipLaunchKernelGGL((k_packed<true, true, false>), g, b, 0, 0, p); }, 300));
    rep("C  packed state only", time_it([&] { hipLaunchKernelGGL((k_packed<false, true, false>), g, b, 0, 0, p); }, 300));
    rep("C' packed state loaded by one lane per env + quad wire", time_it([&] { hipLaunchKernelGGL((k_packed<true, true, true>), g, b, 0, 0, p); }, 300));
    rep("B  wire words requested BEFORE the state rows", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, true>), g, b, 0, 0, p); }, 300));
    rep("B  wire stores independent of the state loads", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, false, true>), g, b, 0, 0, p); }, 300));
    rep("B  + 1000 dependent f32 ops between loads and stores", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, false, false, 500>), g, b, 0, 0, p); }, 300));
    rep("B  + 4000 dependent f32 ops between loads and stores", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, false, false, 2000>), g, b, 0, 0, p); }, 300));
    rep("B  + 8000 dependent f32 ops between loads and stores", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, false, false, 4000>), g, b, 0, 0, p); }, 300));
    rep("B  blocks of 64 threads (2048 blocks)", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, false, false, 0, 64>), dim3(N / 32), dim3(64), 0, 0, p); }, 300));
    rep("B  blocks of 128 threads (1024 blocks)", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, false, false, 0, 128>), dim3(N / 64), dim3(128), 0, 0, p); }, 300));
    rep("B  blocks of 512 threads (256 blocks)", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, false, false, 0, 512>), dim3(N / 256), dim3(512), 0, 0, p); }, 300));
    rep("B  blocks of 1024 threads (128 blocks)", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, false, false, 0, 1024>), dim3(N / 512), dim3(1024), 0, 0, p); }, 300));
    rep("B  blocks of 256 threads + 65 KB dynamic LDS each", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true>), g, b, 65 * 1024, 0, p); }, 300));
    rep("B  blocks of 512 threads + 130 KB dynamic LDS each", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, false, false, 0, 512>), dim3(N / 256), dim3(512), 130 * 1024, 0, p); }, 300));
    rep("B  + delay 500, blocks of 512 threads", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, true, false, false, 500, 512>), dim3(N / 256), dim3(512), 0, 0, p); }, 300));
    rep("A  again", time_it([&] { hipLaunchKernelGGL((k_rows<true, true, false>), g, b, 0, 0, p); }, 300));
    rep("C  packed state only, again", time_it([&] { hipLaunchKernelGGL((k_packed<false, true, false>), g, b, 0, 0, p); }, 300));
    rep("empty (no loads)", time_it([&] { hipLaunchKernelGGL((k_rows<false, false, false>), g, b, 0, 0, p); }, 300));
    return 0;
}
