// fps_prof.hip -- s_memtime phase profile of the shipped FPS round (same instruction sequence as
// fps_body.h, with time stamps between the phases). Development aid: the stamps themselves cost
// ~30 cycles each, so read the numbers as proportions.
#include "../pointnet2_amd/csrc/fps_body.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
using namespace pn2;

template <int T, int P, bool PROF>
__global__ __launch_bounds__(T) void k(int n, int m, int Q, const float *__restrict__ xyz, int *__restrict__ out, unsigned long long *prof)
{
    constexpr int W = T / 64, NS = T * P;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *partial = reinterpret_cast<unsigned long long *>(smem);
    float4 *lds_rank = reinterpret_cast<float4 *>(smem + 256);
    const float *__restrict__ src = xyz + (size_t)blockIdx.x * n * 3;
    int *__restrict__ dst = out + (size_t)blockIdx.x * m;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    float x[P], y[P], z[P], md[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = t * P + p; const int kk0 = (r % Q) * 512 + r / Q; const bool valid = (r < 512 * Q) && (kk0 < n);
        const int kk = valid ? kk0 : 0;
        x[p] = valid ? src[kk * 3] : 0.f; y[p] = valid ? src[kk * 3 + 1] : 0.f; z[p] = valid ? src[kk * 3 + 2] : 0.f;
        md[p] = valid ? 1e38f : 0.f;
        lds_rank[NS - 1 - r] = make_float4(x[p], y[p], z[p], __int_as_float(kk));
    }
    __syncthreads();
    float sx, sy, sz; { const float4 s = lds_rank[NS - 1]; sx = s.x; sy = s.y; sz = s.z; }
    if (t == 0) dst[0] = 0;
    const unsigned low0 = (unsigned)(NS - 1 - t * P);
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 1; j < m; ++j) {
        unsigned long long t0, t1, t2, t3, t4, t5, t6;
        if (PROF) STAMP(t0);
        double kd[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float d = sqdist(x[p], y[p], z[p], sx, sy, sz);
            md[p] = vmin_f32(d, md[p]);
            kd[p] = __hiloint2double(__float_as_int(md[p]), (int)(low0 - (unsigned)p));
        }
        if (PROF) { asm volatile("" : "+v"(kd[0]), "+v"(kd[P - 1])); STAMP(t1); }
#pragma unroll
        for (int st = 1; st < P; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < P; i += 2 * st) asm("v_max_f64 %0, %1, %2" : "=v"(kd[i]) : "v"(kd[i]), "v"(kd[i + st]));
        double bestd = kd[0];
        if (PROF) { asm volatile("" : "+v"(bestd)); STAMP(t2); }
        unsigned long long *slot = partial + (j & 1) * W;
        double wd = wave_max_f64_lane63(bestd);
        if (PROF) { asm volatile("" : "+v"(wd)); STAMP(t3); }
        if (lane == 63) reinterpret_cast<double *>(slot)[w] = wd;
        __syncthreads();
        if (PROF) STAMP(t4);
        const double *dslot = reinterpret_cast<const double *>(slot);
        double key[W];
#pragma unroll
        for (int i = 0; i < W; ++i) key[i] = dslot[i];
#pragma unroll
        for (int st = 1; st < W; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < W; i += 2 * st) asm("v_max_f64 %0, %1, %2" : "=v"(key[i]) : "v"(key[i]), "v"(key[i + st]));
        unsigned win = (unsigned)__double2loint(key[0]);
        if (PROF) { asm volatile("" : "+v"(win)); STAMP(t5); }
        const float4 s = lds_rank[win];
        sx = s.x; sy = s.y; sz = s.z;
        if (t == 0) dst[j] = __float_as_int(s.w);
        if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sx), "+v"(sy), "+v"(sz)); STAMP(t6);
            acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4; acc[5] += t6 - t5; }
    }
    if (PROF && blockIdx.x == 0 && lane == 0 && (w == 0 || w == W - 1))
        for (int i = 0; i < 6; ++i) prof[(w ? 6 : 0) + i] = acc[i];
}
template <int T, int P> static void go(int b, int n, int m, const float *d_xyz, int *d_out, unsigned long long *d_prof)
{
    const int Q = (n + 511) / 512; const size_t lds = 256 + 16 * (size_t)T * P;
    for (int prof = 0; prof < 2; ++prof) {
        auto kern = prof ? k<T, P, true> : k<T, P, false>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, 0, n, m, Q, d_xyz, d_out, d_prof); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, 0, n, m, Q, d_xyz, d_out, d_prof);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("T=%d P=%d prof=%d: %.1f ns/round\n", T, P, prof, ms * 1e6f / 3 / (m - 1));
        if (prof) {
            unsigned long long h[12]; CK(hipMemcpy(h, d_prof, sizeof h, hipMemcpyDeviceToHost));
            const char *nm[6] = {"distance update", "lane tournament", "wave DPP ladder", "write+barrier", "key read+tournament", "winner read"};
            for (int wv = 0; wv < 2; ++wv) { printf("  wave %s:", wv ? "last" : "0   "); for (int i = 0; i < 6; ++i) printf("  %s=%.0f", nm[i], (double)h[wv * 6 + i] / (m - 1)); printf("  (cycles/round)\n"); }
        }
    }
}
int main()
{
    const int b = 32, n = 4096, m = 1024;
    std::vector<float> h((size_t)b * n * 3); uint32_t s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
    float *d_xyz; int *d_out; unsigned long long *d_prof;
    CK(hipMalloc(&d_xyz, h.size() * 4)); CK(hipMalloc(&d_out, (size_t)b * m * 4)); CK(hipMalloc(&d_prof, 128));
    CK(hipMemcpy(d_xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    go<512, 8>(b, n, m, d_xyz, d_out, d_prof);
    go<256, 16>(b, n, m, d_xyz, d_out, d_prof);
    go<1024, 4>(b, n, m, d_xyz, d_out, d_prof);
    return 0;
}
