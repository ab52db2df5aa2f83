// ubench_mfma.hip -- does fp32 MFMA (v_mfma_f32_32x32x2_f32) overlap with VALU work of the OTHER wave on
// the same SIMD? Wave w < 4 (one per SIMD) runs a dependent MFMA chain; wave w + 4 on the same SIMD is
// idle / runs independent v_max_f32 / runs its own MFMA chain. Development aid for csrc/sa_mlp.hip.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_mfma.hip -o build_lab/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float *sink, unsigned long long *ticks, int iters)
{
    const int t = threadIdx.x, w = t >> 6;
    f32x16 acc = {0};
    float a = t * 0.001f, b = 1.0f + t * 1e-6f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = t + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4 || MODE == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    } else if (MODE == 1) {
        for (int it = 0; it < iters * 16; ++it) {       // 16 independent v_max per trip; ~same duration as the MFMA wave
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    if (t == 256 && blockIdx.x == 0) ticks[1] = t1 - t0;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i] + v[i];
    sink[blockIdx.x * 512 + t] = s;
}

template <int MODE> static void run(const char *name)
{
    float *sink; unsigned long long *ticks;
    CK(hipMalloc(&sink, 256 * 512 * 4)); CK(hipMalloc(&ticks, 16));
    const int iters = 512;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, sink, ticks, iters); CK(hipDeviceSynchronize()); }
    // wall time of a long launch: the sustained shader clock under this load = ticks / time
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int long_iters = iters * 16;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, sink, ticks, long_iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost));
    const double tot = (double)(h[0] > h[1] ? h[0] : h[1]);
    printf("%-52s MFMA wave: %.1f ticks per MFMA;  other wave total %.0f ticks (%.2f per v_max); kernel %.3f ms -> %.2f G ticks/s\n", name,
           (double)h[0] / long_iters / 16, (double)h[1], (double)h[1] / long_iters / 256, ms, tot / ms / 1e6);
    CK(hipFree(sink)); CK(hipFree(ticks));
}

int main()
{
    run<0>("second wave idle");
    run<1>("second wave: independent v_max_f32 stream");
    run<2>("second wave: its own MFMA chain");
    return 0;
}
