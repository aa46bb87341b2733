// f64_issue.hip — issue cost (cycles per wave64 instruction, one wave per SIMD and two) of the float64 operations the
// Numba-typed stencil is made of, on gfx950: v_add_f64, v_mul_f64, v_fma_f64, v_cvt_f32_f64, v_cvt_f64_f32, against v_add_f32
// and v_pk_fma_f32.  Independent instructions (8 accumulators), s_memtime around 512 x 64 of them (64 per loop trip: ~100 us, so that the blocks of a launch overlap).
//   hipcc --offload-arch=gfx950 -O2 -o build/microbench/f64_issue tools/microbench/f64_issue.hip && build/microbench/f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(256) issue(unsigned long long* out, double seed) {
    double a[8], b = seed * 1.0000001, c = seed * 0.999999;
    float f[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i; f[i] = (float)(seed + i); p[i] = f2{f[i], f[i] + 1.0f}; }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < 512; ++r) {
#define OP(i)                                                                                                      \
    if (KIND == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
    if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                      \
    if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));                           \
    if (KIND == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));                                   \
    if (KIND == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));                                   \
    if (KIND == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));                         \
    if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7])); \
    if (KIND == 7) asm volatile("v_cvt_f32_f64 %0, %1\n\tv_cvt_f64_f32 %1, %0" : "+v"(f[i]), "+v"(a[i]));
        REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + f[i] + p[i].x + p[i].y;
    if (s == 12345.678) out[1023] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8192);
    const char* names[8] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_add_f32", "v_pk_fma_f32", "cvt pair (dependent)"};
    for (int waves = 1; waves <= 2; ++waves) {
        for (int k = 0; k < 8; ++k) {
            for (int pass = 0; pass < 2; ++pass) {  // (the second pass is reported: clocks up, code cached)
            hipMemset(d, 0, 8192);
            const int blocks = 256 * waves;  // 256-thread blocks: 4 waves = one per SIMD; two blocks per CU = two per SIMD
            switch (k) {
                case 0: issue<0><<<blocks, 256>>>(d, 1.5); break;
                case 1: issue<1><<<blocks, 256>>>(d, 1.5); break;
                case 2: issue<2><<<blocks, 256>>>(d, 1.5); break;
                case 3: issue<3><<<blocks, 256>>>(d, 1.5); break;
                case 4: issue<4><<<blocks, 256>>>(d, 1.5); break;
                case 5: issue<5><<<blocks, 256>>>(d, 1.5); break;
                case 6: issue<6><<<blocks, 256>>>(d, 1.5); break;
                default: issue<7><<<blocks, 256>>>(d, 1.5); break;
            }
            hipDeviceSynchronize();
            }
            unsigned long long h[4];
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            const double n = 512.0 * 64.0 * (k == 7 ? 2 : 1);
            printf("%d wave(s)/SIMD  %-24s %6.2f shader cycles per instruction of one wave\n", waves, names[k], h[0] / n);
        }
    }
    return 0;
}
