// Microbenchmark (diagnostic): in-place read-modify-write of T[seg][env] (float32, env-minor) with the access
// shapes the single-microsecond kernels can use.  Answers: what does one dword per lane cost against 16 B per lane?
//   hipcc --offload-arch=gfx950 -O3 -o rowstream rowstream.hip && ./rowstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// A: one env per lane, ROWS rows per lane, all loads in flight, then stores (dword)
template <int ROWS>
__global__ void __launch_bounds__(256) k_dword(float* T, int stride, int n_seg) {
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c = threadIdx.x >> 6;               // 4 chunks of the wire per block
    const int per = n_seg / 4;
    for (int r0 = c * per; r0 < (c + 1) * per; r0 += ROWS) {
        float v[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) v[u] = T[(size_t)(r0 + u) * stride + e];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) T[(size_t)(r0 + u) * stride + e] = v[u] * 1.0001f + 0.5f;
    }
}
// B: four envs per lane (dwordx4), 256 envs per wave-row
template <int ROWS>
__global__ void __launch_bounds__(256) k_x4(float* T, int stride, int n_seg) {
    const int q = blockIdx.x * 64 + (threadIdx.x & 63);   // env quad
    const int c = threadIdx.x >> 6;
    const int per = n_seg / 4;
    float4* T4 = reinterpret_cast<float4*>(T);
    const int s4 = stride / 4;
    for (int r0 = c * per; r0 < (c + 1) * per; r0 += ROWS) {
        float4 v[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) v[u] = T4[(size_t)(r0 + u) * s4 + q];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            float4 w = v[u];
            w.x = w.x * 1.0001f + 0.5f; w.y = w.y * 1.0001f + 0.5f; w.z = w.z * 1.0001f + 0.5f; w.w = w.w * 1.0001f + 0.5f;
            T4[(size_t)(r0 + u) * s4 + q] = w;
        }
    }
}
// C: two envs per lane (dwordx2)
template <int ROWS>
__global__ void __launch_bounds__(256) k_x2(float* T, int stride, int n_seg) {
    const int q = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c = threadIdx.x >> 6;
    const int per = n_seg / 4;
    float2* T2 = reinterpret_cast<float2*>(T);
    const int s2 = stride / 2;
    for (int r0 = c * per; r0 < (c + 1) * per; r0 += ROWS) {
        float2 v[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) v[u] = T2[(size_t)(r0 + u) * s2 + q];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            float2 w = v[u];
            w.x = w.x * 1.0001f + 0.5f; w.y = w.y * 1.0001f + 0.5f;
            T2[(size_t)(r0 + u) * s2 + q] = w;
        }
    }
}

template <class F> float time_it(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int n_env = cfg == 0 ? 65536 : 32768, n_seg = cfg == 0 ? 128 : 384;  // chunks of n_seg / 4 rows must be multiples of every ROWS used below
        float* T; CK(hipMalloc(&T, (size_t)n_env * n_seg * 4));
        CK(hipMemset(T, 0, (size_t)n_env * n_seg * 4));
        if ((n_seg / 4) % 32 != 0 || n_env % 256 != 0) { printf("bad shape\n"); return 1; }
        const double mb = 2.0 * n_env * n_seg * 4 / 1e6;
        printf("N=%d S=%d: %.1f MB read+write per pass\n", n_env, n_seg, mb);
        auto rep = [&](const char* nm, float ms) { printf("  %-34s %8.2f us  %6.2f TB/s\n", nm, ms * 1e3, mb / 1e6 / (ms * 1e-3)); };
        rep("dword, 8 rows in flight", time_it([&] { hipLaunchKernelGGL(k_dword<8>, dim3(n_env / 64), dim3(256), 0, 0, T, n_env, n_seg); }, 200));
        rep("dword, 16 rows in flight", time_it([&] { hipLaunchKernelGGL(k_dword<16>, dim3(n_env / 64), dim3(256), 0, 0, T, n_env, n_seg); }, 200));
        rep("dword, 32 rows in flight", time_it([&] { hipLaunchKernelGGL(k_dword<32>, dim3(n_env / 64), dim3(256), 0, 0, T, n_env, n_seg); }, 200));
        rep("dwordx2, 8 rows in flight", time_it([&] { hipLaunchKernelGGL(k_x2<8>, dim3(n_env / 128), dim3(256), 0, 0, T, n_env, n_seg); }, 200));
        rep("dwordx2, 16 rows in flight", time_it([&] { hipLaunchKernelGGL(k_x2<16>, dim3(n_env / 128), dim3(256), 0, 0, T, n_env, n_seg); }, 200));
        rep("dwordx4, 4 rows in flight", time_it([&] { hipLaunchKernelGGL(k_x4<4>, dim3(n_env / 256), dim3(256), 0, 0, T, n_env, n_seg); }, 200));
        rep("dwordx4, 8 rows in flight", time_it([&] { hipLaunchKernelGGL(k_x4<8>, dim3(n_env / 256), dim3(256), 0, 0, T, n_env, n_seg); }, 200));
        rep("dwordx4, 16 rows in flight", time_it([&] { hipLaunchKernelGGL(k_x4<16>, dim3(n_env / 256), dim3(256), 0, 0, T, n_env, n_seg); }, 200));
        rep("empty-ish launch (1 block)", time_it([&] { hipLaunchKernelGGL(k_dword<8>, dim3(1), dim3(256), 0, 0, T, n_env, n_seg); }, 200));
        CK(hipFree(T));
    }
    return 0;
}
