// ubench_mfma_bf16.hip -- issue rate of v_mfma_f32_32x32x16_bf16 in the shapes csrc/sa_mlp.hip uses: a chain
// of dependent MFMAs on one accumulator, two interleaved accumulators, one or two waves per SIMD.
// Development aid.  hipcc --offload-arch=gfx950 -O3 -std=c++17 [-mllvm -amdgpu-mfma-vgpr-form=1] scripts/ubench_mfma_bf16.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: one accumulator, 6 dependent MFMAs per trip; 1: two accumulators alternating; 2: four accumulators
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float *sink, unsigned long long *ticks, int iters)
{
    const int t = threadIdx.x;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    u32x4 w[3], x[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { w[i] = u32x4{0x3f803f80u + t, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i}; x[i] = u32x4{0x3f803f80u, 0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u + t}; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define M(A, WL, XL) acc[A] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[WL]), __builtin_bit_cast(bf16x8, x[XL]), acc[A], 0, 0, 0)
        if (MODE == 0) { M(0, 0, 2); M(0, 1, 1); M(0, 2, 0); M(0, 0, 1); M(0, 1, 0); M(0, 0, 0); }
        if (MODE == 1) { M(0, 0, 2); M(1, 1, 1); M(0, 2, 0); M(1, 0, 1); M(0, 1, 0); M(1, 0, 0); }
        if (MODE == 2) { M(0, 0, 2); M(1, 1, 1); M(2, 2, 0); M(3, 0, 1); M(0, 1, 0); M(1, 0, 0); }
#undef M
        asm volatile("" : "+v"(w[0]), "+v"(x[0]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    float s = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    sink[blockIdx.x * 64 * WAVES + t] = s;
}

template <int MODE, int WAVES> static void run(const char *name)
{
    float *sink; unsigned long long *ticks;
    CK(hipMalloc(&sink, 256 * 64 * WAVES * 4)); CK(hipMalloc(&ticks, 16));
    const int iters = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, sink, ticks, iters); CK(hipDeviceSynchronize()); }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, sink, ticks, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h; CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
    const double mfmas_per_simd = (double)iters * 6 * (WAVES / 4.0);
    printf("%-44s %5.1f ticks per MFMA per wave; %5.1f ns*2.4 cycles per MFMA per SIMD; %.1f TFLOP/s bf16\n", name, (double)h / iters / 6,
           ms * 1e6 * 2.4 / mfmas_per_simd, 256.0 * WAVES * iters * 6 * 32768 / ms / 1e9);
    CK(hipFree(sink)); CK(hipFree(ticks));
}

int main()
{
    run<0, 4>("1 wave/SIMD, one accumulator (chain)");
    run<1, 4>("1 wave/SIMD, two accumulators");
    run<2, 4>("1 wave/SIMD, four accumulators");
    run<0, 8>("2 waves/SIMD, one accumulator each");
    run<1, 8>("2 waves/SIMD, two accumulators each");
    return 0;
}
