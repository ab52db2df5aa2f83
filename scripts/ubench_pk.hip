// ubench_pk.hip -- packed-fp32 (v_pk_mul_f32 / v_pk_add_f32) issue rate against plain fp32 VALU ops at
// the FPS kernel's occupancy (512 threads on a CU = 2 waves per SIMD). Development aid.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_pk.hip -o build_lab/ubench_pk
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(float *sink, unsigned long long *ticks, int iters)
{
    const int t = threadIdx.x;
    float a[16];
    float2v p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = t * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) { p[i].x = a[2 * i]; p[i].y = a[2 * i + 1]; }
    float b = 1.0001f;
    float2v bb = {1.0001f, 0.9999f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {           // 16 independent v_mul_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if (KIND == 1) {    // 8 independent v_pk_mul_f32 (same 16 multiplies)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(bb));
        } else if (KIND == 2) {    // dependent chain: 16 v_mul_f32 on one register
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));
        } else if (KIND == 3) {    // dependent chain: 16 v_pk_mul_f32 on one pair
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[0]) : "v"(bb));
        } else if (KIND == 4) {    // 8 independent v_pk_add_f32 with negated, broadcast-low second source
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(bb));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0 && blockIdx.x == 0) *ticks = t1 - t0;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    sink[blockIdx.x * 1024 + t] = s;
}

template <int KIND> static void run(const char *name, int T, int ops_per_iter)
{
    float *sink; unsigned long long *ticks;
    CK(hipMalloc(&sink, 256 * 1024 * 4)); CK(hipMalloc(&ticks, 8));
    const int iters = 2048;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(T), 0, 0, sink, ticks, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(T), 0, 0, sink, ticks, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h; CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
    printf("%-44s T=%4d: %.2f cycles per instruction per wave (%.2f per SIMD-instruction)\n", name, T,
           (double)h / iters / ops_per_iter, (double)h / iters / ops_per_iter / (T / 256.0));
    CK(hipFree(sink)); CK(hipFree(ticks));
}

int main()
{
    for (int T : {256, 512, 1024}) {
        run<0>("16 independent v_mul_f32", T, 16);
        run<1>("8 independent v_pk_mul_f32", T, 8);
        run<4>("8 independent v_pk_add_f32 (neg, bcast lo)", T, 8);
        run<2>("16 dependent v_mul_f32", T, 16);
        run<3>("16 dependent v_pk_mul_f32", T, 16);
    }
    return 0;
}
