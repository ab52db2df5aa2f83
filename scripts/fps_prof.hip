// fps_prof.hip -- s_memtime phase profile of the shipped FPS round structure (development aid).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
__device__ __forceinline__ float vmin_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define DPP_MAX(v, ctrl) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl " bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ int wave_max_fast(int v)
{
    DPP_MAX(v, "quad_perm:[1,0,3,2] row_mask:0xf"); DPP_MAX(v, "quad_perm:[2,3,0,1] row_mask:0xf");
    DPP_MAX(v, "row_half_mirror row_mask:0xf"); DPP_MAX(v, "row_mirror row_mask:0xf");
    DPP_MAX(v, "row_bcast:15 row_mask:0xa"); DPP_MAX(v, "row_bcast:31 row_mask:0xc");
    return __builtin_amdgcn_readlane(v, 63);
}
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int T, int P, bool PROF>
__global__ __launch_bounds__(T) void k(int n, int m, int Q, const float *__restrict__ xyz, int *__restrict__ out, unsigned long long *prof)
{
    constexpr int W = T / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int2 *partial = reinterpret_cast<int2 *>(smem);
    float4 *lds_xyz = reinterpret_cast<float4 *>(smem + 256);
    const float *__restrict__ src = xyz + (size_t)blockIdx.x * n * 3;
    int *__restrict__ dst = out + (size_t)blockIdx.x * m;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int kk = t; kk < n; kk += T) lds_xyz[kk] = make_float4(src[kk * 3], src[kk * 3 + 1], src[kk * 3 + 2], 0.f);
    __syncthreads();
    float x[P], y[P], z[P], md[P]; int kidx[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = t * P + p; const int kk = (r % Q) * 512 + r / Q; const bool valid = (r < 512 * Q) && (kk < n);
        kidx[p] = valid ? kk : 0; const float4 v = lds_xyz[valid ? kk : 0];
        x[p] = valid ? v.x : 0.f; y[p] = valid ? v.y : 0.f; z[p] = valid ? v.z : 0.f; md[p] = valid ? 1e38f : -1.0f;
    }
    int cur = 0; if (t == 0) dst[0] = 0;
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 1; j < m; ++j) {
        unsigned long long t0, t1, t2, t3, t4, t5, t6;
        if (PROF) STAMP(t0);
        const float4 s = lds_xyz[cur];
        float sx = s.x, sy = s.y, sz = s.z;
        if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sx), "+v"(sy), "+v"(sz)); STAMP(t1); }
        int bv = INT_MIN, bk = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float d = sqdist(x[p], y[p], z[p], sx, sy, sz); md[p] = vmin_f32(d, md[p]);
            const int iv = __float_as_int(md[p]); if (iv > bv) { bv = iv; bk = kidx[p]; }
        }
        if (PROF) { asm volatile("" : "+v"(bv), "+v"(bk)); STAMP(t2); }
        const int wm = wave_max_fast(bv);
        const int wl = __builtin_ctzll(__ballot(bv == wm));
        int wk = __builtin_amdgcn_readlane(bk, wl);
        if (PROF) { asm volatile("" : "+s"(wk)); STAMP(t3); }
        int2 *slot = partial + (j & 1) * W;
        if (lane == 0) slot[w] = make_int2(wm, wk);
        __syncthreads();
        if (PROF) STAMP(t4);
        int bm = slot[0].x; cur = slot[0].y;
#pragma unroll
        for (int i = 1; i < W; ++i) { const int2 q = slot[i]; if (q.x > bm) { bm = q.x; cur = q.y; } }
        if (PROF) { asm volatile("" : "+v"(cur)); STAMP(t5); }
        if (t == 0) dst[j] = cur;
        if (PROF) { STAMP(t6); acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4; acc[5] += t6 - t5; }
    }
    if (PROF && blockIdx.x == 0 && lane == 0 && (w == 0 || w == W - 1))
        for (int i = 0; i < 6; ++i) prof[(w ? 6 : 0) + i] = acc[i];
}
template <int T, int P> static void go(int b, int n, int m, const float *d_xyz, int *d_out, unsigned long long *d_prof)
{
    const int Q = (n + 511) / 512; const size_t lds = 256 + 16 * (size_t)n;
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
            const char *nm[6] = {"read s", "compute+local argmax", "wave reduce", "write+barrier", "select", "store+loop"};
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
