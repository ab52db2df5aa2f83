// ubench.hip -- latency/throughput microbenchmarks that size the FPS round (development aid).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench.hip -o build_lab/ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;

// each kernel: one block per CU-ish (grid=32), T threads; lane 0 of wave 0 records s_memtime delta
template <int KIND>
__global__ void k(int T, float *sink, unsigned long long *ticks, int iters)
{
    __shared__ float lds[4096];
    __shared__ int slot[64];
    const int t = threadIdx.x;
    lds[t] = t; lds[t + 1024] = t * 2.f;
    __syncthreads();
    float a = t * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 2.f, f = 3.f, g = 4.f, h = 5.f;
    int iv = t, cur = t & 1023;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {            // dependent VALU chain: 8 dependent mul-adds (separate mul, add)
#pragma unroll
            for (int u = 0; u < 8; ++u) { a = __fmul_rn(a, b); a = __fadd_rn(a, c); }
        } else if (KIND == 1) {     // independent VALU: 8 chains x 2 ops
            a = __fmul_rn(a, b); c = __fmul_rn(c, b); d = __fmul_rn(d, b); e = __fmul_rn(e, b);
            f = __fmul_rn(f, b); g = __fmul_rn(g, b); h = __fmul_rn(h, b); iv += 3;
            a = __fadd_rn(a, 1.f); c = __fadd_rn(c, 1.f); d = __fadd_rn(d, 1.f); e = __fadd_rn(e, 1.f);
            f = __fadd_rn(f, 1.f); g = __fadd_rn(g, 1.f); h = __fadd_rn(h, 1.f); iv ^= 5;
        } else if (KIND == 2) {     // LDS dependent broadcast read chain (address from previous value)
            const float v = lds[cur];
            cur = ((int)v + 1) & 1023;
        } else if (KIND == 3) {     // barrier only
            __syncthreads();
        } else if (KIND == 4) {     // lane0 LDS write + barrier + broadcast read (dependent)
            if ((t & 63) == 0) slot[t >> 6] = cur + i;
            __syncthreads();
            cur = slot[(cur + i) & 3] & 1023;
        } else if (KIND == 5) {     // 4-step DPP butterfly max (compiler form) + readfirstlane
            int v = iv;
            v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
            v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
            v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
            v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
            iv = __builtin_amdgcn_readfirstlane(v) + t;
        } else if (KIND == 6) {     // readlane -> SGPR -> VALU dependent
            iv = __builtin_amdgcn_readlane(iv, 17) + t;
        } else if (KIND == 7) {     // ballot + ctz + readlane chain
            const unsigned long long m = __ballot(iv > i);
            const int l = m ? __builtin_ctzll(m) : 0;
            iv = __builtin_amdgcn_readlane(iv, l) + t + 1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0 && blockIdx.x == 0) *ticks = t1 - t0;
    sink[blockIdx.x * 1024 + t] = a + c + d + e + f + g + h + iv + cur;
}

template <int KIND>
static void bench(const char *name, int T, int opsPerIter)
{
    float *sink; unsigned long long *ticks;
    CK(hipMalloc(&sink, 32 * 1024 * 4)); CK(hipMalloc(&ticks, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(32), dim3(T), 0, 0, T, sink, ticks, ITERS);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(32), dim3(T), 0, 0, T, sink, ticks, ITERS * 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h; CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
    const double ns = ms * 1e6 / (ITERS * 16.0);
    printf("%-44s T=%4d : %8.2f ns/iter  %8.1f ticks/iter  (%.2f ns per op; tick=%.3f ns)\n", name, T, ns,
           (double)h / (ITERS * 16.0), ns / opsPerIter, ns / ((double)h / (ITERS * 16.0)));
    CK(hipFree(sink)); CK(hipFree(ticks));
}

int main()
{
    for (int T : {64, 256, 512, 1024}) {
        bench<0>("dependent VALU chain (16 ops)", T, 16);
        bench<1>("independent VALU (16 ops)", T, 16);
    }
    for (int T : {64, 256, 1024}) bench<2>("LDS dependent broadcast read", T, 1);
    for (int T : {64, 256, 512, 1024}) bench<3>("s_barrier only", T, 1);
    for (int T : {256, 512, 1024}) bench<4>("lane0 ds_write + barrier + ds_read", T, 1);
    for (int T : {64, 256, 1024}) bench<5>("4-step DPP max + readfirstlane", T, 1);
    for (int T : {64, 256}) bench<6>("readlane -> VALU", T, 1);
    for (int T : {64, 256}) bench<7>("ballot+ctz+readlane", T, 1);
    return 0;
}
