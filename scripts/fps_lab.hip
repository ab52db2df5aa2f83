// fps_lab.hip -- development microbenchmark for the FPS inner loop (NOT product code).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/fps_lab.hip -o gpurun_out/fps_lab
// Runs variants of the per-round synchronisation chain on B clouds and prints ns/round.
// Correctness of each variant is checked against variant 0 (the shipped v1 structure).
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
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int row16_max(int v)
{
    v = max(v, dpp_mov<0xB1>(v)); v = max(v, dpp_mov<0x4E>(v)); v = max(v, dpp_mov<0x141>(v)); v = max(v, dpp_mov<0x140>(v));
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
    v = row16_max(v);
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(r0, r1), max(r2, r3));
}
// asm DPP: single-instruction v_max_i32 with DPP source (2 wait states folded in)
#define DPP_MAX(v, ctrl) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ int wave_max_asm(int v)
{
    DPP_MAX(v, "quad_perm:[1,0,3,2]");
    DPP_MAX(v, "quad_perm:[2,3,0,1]");
    DPP_MAX(v, "row_half_mirror");
    DPP_MAX(v, "row_mirror");
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// MODE 0: shipped v1 structure. MODE 1: asm wave reduce + W-specialised block reduce by broadcast reads.
// MODE 2: MODE 1 + partial carries xyz (no second LDS lookup).
// ABL (ablation bit mask): 1 = skip distance update, 2 = skip wave reduce (fake), 4 = skip barrier+block reduce
template <int T, int P, int MODE, int ABL>
__global__ __launch_bounds__(T) void fps_lab_kernel(int n, int m, int Q, const float *__restrict__ xyz, int *__restrict__ out)
{
    constexpr int W = T / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4 *partial4 = reinterpret_cast<int4 *>(smem);                // [2][W] {val,x,y,z}  (MODE 2) / int2 view (MODE 0/1)
    int *partialk = reinterpret_cast<int *>(smem + 512);            // [2][W]
    float4 *lds_xyz = reinterpret_cast<float4 *>(smem + 1024);
    const float *__restrict__ src = xyz + (size_t)blockIdx.x * n * 3;
    int *__restrict__ dst = out + (size_t)blockIdx.x * m;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int k = t; k < n; k += T) lds_xyz[k] = make_float4(src[k * 3], src[k * 3 + 1], src[k * 3 + 2], 0.f);
    __syncthreads();
    typedef float vecP __attribute__((ext_vector_type(P)));
    vecP x, y, z;
    float md[P];
    int kidx[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = t * P + p;
        const int k = (r % Q) * 512 + r / Q;
        const bool valid = (r < 512 * Q) && (k < n);
        kidx[p] = valid ? k : 0;
        const float4 v = lds_xyz[valid ? k : 0];
        x[p] = valid ? v.x : 0.f; y[p] = valid ? v.y : 0.f; z[p] = valid ? v.z : 0.f;
        md[p] = valid ? 1e38f : -1.0f;
    }
    int cur = 0;
    if (t == 0) dst[0] = 0;
    float sx, sy, sz;
    { const float4 s = lds_xyz[0]; sx = s.x; sy = s.y; sz = s.z; }
    for (int j = 1; j < m; ++j) {
        if (MODE < 2) { const float4 s = lds_xyz[cur]; sx = s.x; sy = s.y; sz = s.z; }
        int bv = INT_MIN, bk = 0, bp = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (!(ABL & 1)) { const float d = sqdist(x[p], y[p], z[p], sx, sy, sz); md[p] = __builtin_fminf(d, md[p]); }
            const int iv = __float_as_int(md[p]);
            if (iv > bv) { bv = iv; bk = kidx[p]; bp = p; }
        }
        int wm, wl;
        if (ABL & 2) { wm = __builtin_amdgcn_readfirstlane(bv); wl = 0; }
        else {
            wm = (MODE == 0) ? wave_max(bv) : wave_max_asm(bv);
            wl = __builtin_ctzll(__ballot(bv == wm));
        }
        const int wk = __builtin_amdgcn_readlane(bk, wl);
        if (MODE == 2) {
            const int wp = __builtin_amdgcn_readlane(bp, wl);
            const float cx = x[wp], cy = y[wp], cz = z[wp];     // uniform dynamic index -> v_movrels
            if (lane == wl) {
                partial4[(j & 1) * W + w] = make_int4(wm, __float_as_int(cx), __float_as_int(cy), __float_as_int(cz));
                partialk[(j & 1) * W + w] = wk;
            }
        } else {
            if (lane == 0) reinterpret_cast<int2 *>(partial4)[(j & 1) * W + w] = make_int2(wm, wk);
        }
        if (ABL & 4) { cur = wk; if (MODE == 2) { sx += 1e-9f; } if (t == 0) dst[j] = cur; continue; }
        __syncthreads();
        if (MODE == 0) {
            int2 pp = make_int2(INT_MIN, 0);
            if (lane < W) pp = reinterpret_cast<int2 *>(partial4)[(j & 1) * W + lane];
            const int bm = (W <= 16) ? __builtin_amdgcn_readfirstlane(row16_max(pp.x)) : wave_max(pp.x);
            cur = __builtin_amdgcn_readlane(pp.y, __builtin_ctzll(__ballot(pp.x == bm)));
        } else if (MODE == 1) {
            // broadcast-read all W partials, select in VALU on uniform data
            const int2 *pp = reinterpret_cast<int2 *>(partial4) + (j & 1) * W;
            int bm = pp[0].x; cur = pp[0].y;
#pragma unroll
            for (int i = 1; i < W; ++i) { const int2 q = pp[i]; if (q.x > bm) { bm = q.x; cur = q.y; } }
        } else {
            const int4 *pp = partial4 + (j & 1) * W;
            const int *pk = partialk + (j & 1) * W;
            int4 b4 = pp[0]; cur = pk[0];
#pragma unroll
            for (int i = 1; i < W; ++i) { const int4 q = pp[i]; const int qk = pk[i]; if (q.x > b4.x) { b4 = q; cur = qk; } }
            sx = __int_as_float(b4.y); sy = __int_as_float(b4.z); sz = __int_as_float(b4.w);
        }
        if (t == 0) dst[j] = cur;
    }
}


// MODE 3 (separate kernel): row-level DPP max, then ds_max_u64 on a rotating LDS slot.
// key = (value bits << 32) | (0xFFFFFFFF - rank); cloud kept in LDS in RANK order as (x,y,z,k).
__device__ __forceinline__ float vmin_f32(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int T, int P, int ABL>
__global__ __launch_bounds__(T) void fps_atomic_kernel(int n, int m, int Q, const float *__restrict__ xyz, int *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(smem);   // [3] (+pad)
    float4 *lds_rank = reinterpret_cast<float4 *>(smem + 64);                   // [T*P] rank order (x,y,z,k)
    const float *__restrict__ src = xyz + (size_t)blockIdx.x * n * 3;
    int *__restrict__ dst = out + (size_t)blockIdx.x * m;
    const int t = threadIdx.x;
    float x[P], y[P], z[P], md[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = t * P + p;
        const int k = (r % Q) * 512 + r / Q;
        const bool valid = (r < 512 * Q) && (k < n);
        const int kk = valid ? k : 0;
        x[p] = valid ? src[kk * 3] : 0.f; y[p] = valid ? src[kk * 3 + 1] : 0.f; z[p] = valid ? src[kk * 3 + 2] : 0.f;
        md[p] = valid ? 1e38f : 0.0f;
        lds_rank[r] = make_float4(x[p], y[p], z[p], __int_as_float(kk));
    }
    if (t < 3) slot[t] = 0ull;
    __syncthreads();
    if (t == 0) dst[0] = 0;
    float4 s = lds_rank[0];
    const unsigned rank0 = 0xFFFFFFFFu - (unsigned)(t * P);
    for (int j = 1; j < m; ++j) {
        int bv = -1, bp = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (!(ABL & 1)) { const float d = sqdist(x[p], y[p], z[p], s.x, s.y, s.z); md[p] = vmin_f32(d, md[p]); }
            const int iv = __float_as_int(md[p]);
            if (iv > bv) { bv = iv; bp = p; }
        }
        int rm = bv;
        DPP_MAX(rm, "quad_perm:[1,0,3,2]");
        DPP_MAX(rm, "quad_perm:[2,3,0,1]");
        DPP_MAX(rm, "row_half_mirror");
        DPP_MAX(rm, "row_mirror");
        unsigned long long *sl = slot + (j % 3);
        if (bv == rm) {
            const unsigned long long key = ((unsigned long long)(unsigned)bv << 32) | (unsigned long long)(rank0 - (unsigned)bp);
            __hip_atomic_fetch_max(sl, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (t == 0) slot[(j + 1) % 3] = 0ull;
        __syncthreads();
        const unsigned long long win = *sl;
        const unsigned rank = 0xFFFFFFFFu - (unsigned)win;
        s = lds_rank[rank];
        if (t == 0) dst[j] = __float_as_int(s.w);
    }
}
template <int T, int P, int ABL>
static void report_atomic(const char *name, int b, int n, int m, const float *d_xyz, int *d_out, const std::vector<int> &ref)
{
    const int Q = (n + 511) / 512;
    const size_t lds = 64 + 16 * (size_t)T * P;
    auto kern = fps_atomic_kernel<T, P, ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, 0, n, m, Q, d_xyz, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, 0, n, m, Q, d_xyz, d_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<int> got((size_t)b * m);
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    printf("%-34s T=%4d P=%2d mode=3 abl=%d : %8.1f ns/round  %s\n", name, T, P, ABL, ms * 1e6f / 5 / (m - 1),
           ABL ? "(ablation)" : (got == ref ? "OK" : "MISMATCH"));
}


// MODE 4: wave max into ALL lanes without SGPR (4 DPP + permlane16/32 swap), ballot->exec winner write,
// tree select of the W partials from broadcast reads, cloud copy in LDS (original order, w = k unused).
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int wave_max_all(int v)
{
    DPP_MAX(v, "quad_perm:[1,0,3,2]");
    DPP_MAX(v, "quad_perm:[2,3,0,1]");
    DPP_MAX(v, "row_half_mirror");
    DPP_MAX(v, "row_mirror");
    v2u r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = max((int)r.x, (int)r.y);
    v2u q = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)q.x, (int)q.y);
}
template <int T, int P, int ABL>
__global__ __launch_bounds__(T) void fps_m4_kernel(int n, int m, int Q, const float *__restrict__ xyz, int *__restrict__ out)
{
    constexpr int W = T / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int2 *partial = reinterpret_cast<int2 *>(smem);                 // [2][W]
    float4 *lds_xyz = reinterpret_cast<float4 *>(smem + 256);
    const float *__restrict__ src = xyz + (size_t)blockIdx.x * n * 3;
    int *__restrict__ dst = out + (size_t)blockIdx.x * m;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int k = t; k < n; k += T) lds_xyz[k] = make_float4(src[k * 3], src[k * 3 + 1], src[k * 3 + 2], 0.f);
    __syncthreads();
    float x[P], y[P], z[P], md[P];
    int kidx[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = t * P + p;
        const int k = (r % Q) * 512 + r / Q;
        const bool valid = (r < 512 * Q) && (k < n);
        kidx[p] = valid ? k : 0;
        const float4 v = lds_xyz[valid ? k : 0];
        x[p] = valid ? v.x : 0.f; y[p] = valid ? v.y : 0.f; z[p] = valid ? v.z : 0.f;
        md[p] = valid ? 1e38f : -1.0f;
    }
    int cur = 0;
    if (t == 0) dst[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float4 s = lds_xyz[cur];
        int bv = INT_MIN, bk = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (!(ABL & 1)) { const float d = sqdist(x[p], y[p], z[p], s.x, s.y, s.z); md[p] = vmin_f32(d, md[p]); }
            const int iv = __float_as_int(md[p]);
            if (iv > bv) { bv = iv; bk = kidx[p]; }
        }
        const int wm = wave_max_all(bv);
        const unsigned long long hit = __ballot(bv == wm);
        const int wl = __builtin_ctzll(hit);
        int2 *slot = partial + (j & 1) * W;
        if (lane == wl) slot[w] = make_int2(bv, bk);
        __syncthreads();
        // tree select over W partials (broadcast reads), ties -> lower wave
        int2 q[W];
#pragma unroll
        for (int i = 0; i < W; ++i) q[i] = slot[i];
#pragma unroll
        for (int st = 1; st < W; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < W; i += 2 * st)
                if (q[i + st].x > q[i].x) q[i] = q[i + st];
        cur = q[0].y;
        if (t == 0) dst[j] = cur;
    }
}
template <int T, int P, int ABL>
static void report_m4(const char *name, int b, int n, int m, const float *d_xyz, int *d_out, const std::vector<int> &ref)
{
    const int Q = (n + 511) / 512;
    const size_t lds = 256 + 16 * (size_t)n;
    auto kern = fps_m4_kernel<T, P, ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, 0, n, m, Q, d_xyz, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, 0, n, m, Q, d_xyz, d_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<int> got((size_t)b * m);
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    printf("%-34s T=%4d P=%2d mode=4 abl=%d : %8.1f ns/round  %s\n", name, T, P, ABL, ms * 1e6f / 5 / (m - 1),
           ABL ? "(ablation)" : (got == ref ? "OK" : "MISMATCH"));
}


// MODE 5: value-only local max (v_max tree), wave max in all lanes, P ballots + scalar resolution of
// (lowest lane, lowest slot); cloud kept in LDS in RANK order as (x,y,z,k); partial = {value, rank}.
template <int T, int P, int ABL>
__global__ __launch_bounds__(T) void fps_m5_kernel(int n, int m, int Q, const float *__restrict__ xyz, int *__restrict__ out)
{
    constexpr int W = T / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int2 *partial = reinterpret_cast<int2 *>(smem);                 // [2][W]
    float4 *lds_rank = reinterpret_cast<float4 *>(smem + 256);      // [T*P]
    const float *__restrict__ src = xyz + (size_t)blockIdx.x * n * 3;
    int *__restrict__ dst = out + (size_t)blockIdx.x * m;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    float x[P], y[P], z[P], md[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = t * P + p;
        const int k = (r % Q) * 512 + r / Q;
        const bool valid = (r < 512 * Q) && (k < n);
        const int kk = valid ? k : 0;
        x[p] = valid ? src[kk * 3] : 0.f; y[p] = valid ? src[kk * 3 + 1] : 0.f; z[p] = valid ? src[kk * 3 + 2] : 0.f;
        md[p] = valid ? 1e38f : -1.0f;
        lds_rank[r] = make_float4(x[p], y[p], z[p], __int_as_float(kk));
    }
    __syncthreads();
    if (t == 0) dst[0] = 0;
    float4 s = lds_rank[0];
    for (int j = 1; j < m; ++j) {
        int iv[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (!(ABL & 1)) { const float d = sqdist(x[p], y[p], z[p], s.x, s.y, s.z); md[p] = vmin_f32(d, md[p]); }
            iv[p] = __float_as_int(md[p]);
        }
        int lm[P];
#pragma unroll
        for (int p = 0; p < P; ++p) lm[p] = iv[p];
#pragma unroll
        for (int st = 1; st < P; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < P; i += 2 * st) lm[i] = max(lm[i], lm[i + st]);
        const int wm = wave_max_all(lm[0]);
        unsigned long long mk[P];
        unsigned long long any = 0ull;
#pragma unroll
        for (int p = 0; p < P; ++p) { mk[p] = __ballot(iv[p] == wm); any |= mk[p]; }
        const int wl = __builtin_ctzll(any);
        int ps = P - 1;
#pragma unroll
        for (int p = P - 2; p >= 0; --p) ps = ((mk[p] >> wl) & 1ull) ? p : ps;
        const int rank = (w * 64 + wl) * P + ps;
        int2 *slot = partial + (j & 1) * W;
        if (lane == 0) slot[w] = make_int2(wm, rank);
        __syncthreads();
        int2 q[W];
#pragma unroll
        for (int i = 0; i < W; ++i) q[i] = slot[i];
#pragma unroll
        for (int st = 1; st < W; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < W; i += 2 * st)
                if (q[i + st].x > q[i].x) q[i] = q[i + st];
        s = lds_rank[q[0].y];
        if (t == 0) dst[j] = __float_as_int(s.w);
    }
}
template <int T, int P, int ABL>
static void report_m5(const char *name, int b, int n, int m, const float *d_xyz, int *d_out, const std::vector<int> &ref)
{
    const int Q = (n + 511) / 512;
    const size_t lds = 256 + 16 * (size_t)T * P;
    auto kern = fps_m5_kernel<T, P, ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, 0, n, m, Q, d_xyz, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, 0, n, m, Q, d_xyz, d_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<int> got((size_t)b * m);
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    printf("%-34s T=%4d P=%2d mode=5 abl=%d : %8.1f ns/round  %s\n", name, T, P, ABL, ms * 1e6f / 5 / (m - 1),
           ABL ? "(ablation)" : (got == ref ? "OK" : "MISMATCH"));
}

template <int T, int P, int MODE, int ABL>
static float run(int b, int n, int m, const float *d_xyz, int *d_out, int reps)
{
    const int Q = (n + 511) / 512;
    const size_t lds = 1024 + 16 * (size_t)n;
    auto kern = fps_lab_kernel<T, P, MODE, ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, 0, n, m, Q, d_xyz, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, 0, n, m, Q, d_xyz, d_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e6f / reps / (m - 1);   // ns per round
}

template <int T, int P, int MODE, int ABL>
static void report(const char *name, int b, int n, int m, const float *d_xyz, int *d_out, const std::vector<int> &ref)
{
    const float ns = run<T, P, MODE, ABL>(b, n, m, d_xyz, d_out, 5);
    std::vector<int> got((size_t)b * m);
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    bool ok = ref.empty() ? true : (got == ref);
    printf("%-34s T=%4d P=%2d mode=%d abl=%d : %8.1f ns/round  %s\n", name, T, P, MODE, ABL, ns, ABL ? "(ablation)" : (ok ? "OK" : "MISMATCH"));
}

int main(int argc, char **argv)
{
    const int b = 32, n = argc > 1 ? atoi(argv[1]) : 4096, m = argc > 2 ? atoi(argv[2]) : 1024;
    std::vector<float> h((size_t)b * n * 3);
    uint32_t s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
    float *d_xyz; int *d_out;
    CK(hipMalloc(&d_xyz, h.size() * 4)); CK(hipMalloc(&d_out, (size_t)b * m * 4));
    CK(hipMemcpy(d_xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> ref;
    if (n > 4096) run<1024, 8, 0, 0>(b, n, m, d_xyz, d_out, 1); else run<1024, 4, 0, 0>(b, n, m, d_xyz, d_out, 1);
    ref.resize((size_t)b * m);
    CK(hipMemcpy(ref.data(), d_out, ref.size() * 4, hipMemcpyDeviceToHost));
    if (n == 4096) {
        report<1024, 4, 0, 0>("v1 shipped", b, n, m, d_xyz, d_out, ref);
        report<512, 8, 1, 0>("asm-dpp + bcast select", b, n, m, d_xyz, d_out, ref);
        report_m4<512, 8, 0>("m4 all-lane max+tree", b, n, m, d_xyz, d_out, ref);
    } else if (n == 8192) {
        report<1024, 8, 0, 0>("v1 shipped", b, n, m, d_xyz, d_out, ref);
        report<256, 32, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
        report<512, 16, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
        report<1024, 8, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
        report_m4<512, 16, 0>("m4", b, n, m, d_xyz, d_out, ref);
        report_m4<1024, 8, 0>("m4", b, n, m, d_xyz, d_out, ref);
    } else if (n == 2048) {
        report<1024, 2, 0, 0>("v1 shipped", b, n, m, d_xyz, d_out, ref);
        report<256, 8, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
        report<512, 4, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
        report_m4<256, 8, 0>("m4", b, n, m, d_xyz, d_out, ref);
        report_m4<512, 4, 0>("m4", b, n, m, d_xyz, d_out, ref);
    } else if (n == 1024) {
        report<1024, 1, 0, 0>("v1 shipped", b, n, m, d_xyz, d_out, ref);
        report<256, 4, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
        report<512, 2, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
        report_m4<256, 4, 0>("m4", b, n, m, d_xyz, d_out, ref);
        report_m4<512, 2, 0>("m4", b, n, m, d_xyz, d_out, ref);
        report_m4<128, 8, 0>("m4", b, n, m, d_xyz, d_out, ref);
        report<128, 8, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
    } else if (n == 512) {
        report<512, 1, 0, 0>("v1 shipped", b, n, m, d_xyz, d_out, ref);
        report<256, 2, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
        report<128, 4, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
        report<64, 8, 1, 0>("mode1", b, n, m, d_xyz, d_out, ref);
        report_m4<256, 2, 0>("m4", b, n, m, d_xyz, d_out, ref);
        report_m4<128, 4, 0>("m4", b, n, m, d_xyz, d_out, ref);
        report_m4<64, 8, 0>("m4", b, n, m, d_xyz, d_out, ref);
    }
    return 0;
}
