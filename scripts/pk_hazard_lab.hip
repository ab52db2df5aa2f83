// pk_hazard_lab.hip -- do v_pk_*_f32 results survive a neighbour wave that issues MFMAs? scripts/fps_concurrency_lab.hip
// showed that exactly the FPS variants whose distance update uses v_pk_add_f32 / v_pk_mul_f32 pick different samples when an
// MFMA kernel shares the GPU. This probe evaluates the same dependent chain (subtract, square, add, add, min) packed and
// scalar in ONE kernel and counts the lanes where the two disagree, alone and beside the MFMA kernel, for several numbers
// of wait states between dependent packed instructions. Development aid.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void ag_mfma(float *sink, int iters)
{
    f32x16 acc = {0};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {7, 6, 5, 4, 3, 2, 1, 0};
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (acc[0] == 123.456f) sink[0] = acc[1];
}
__global__ void ag_valu(float *sink, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.9999f + 1e-4f; }
    if (a == 123.456f) sink[0] = a + b;
}

#define NOPSTR_0 ""
#define NOPSTR_1 "s_nop 0\n\t"
#define NOPSTR_2 "s_nop 1\n\t"
#define NOPSTR_3 "s_nop 2\n\t"
#define NOPSTR_5 "s_nop 4\n\t"
#define NOPSTR_8 "s_nop 7\n\t"
#define NOPSTR_16 "s_nop 7\n\ts_nop 7\n\t"

// MODE 100: compiler-generated packed code (no asm); 101: scalar asm only (control)
template <int MODE>
__device__ __forceinline__ f2 chain(f2 x, f2 y, f2 z, f2 s, f2 t)
{
    if (MODE == 100) {
        f2 dx = x - s.x, dy = y - s.y, dz = z - t.x;
        f2 r = dx * dx; f2 q = dy * dy; f2 u = dz * dz;
        r = r + q; r = r + u;
        return r;
    }
    f2 dx, dy, dz;
#define CHAIN(N)                                                                                                         \
    asm volatile("v_pk_add_f32 %0, %3, %6 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                  \
                 "v_pk_add_f32 %1, %4, %6 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"                    \
                 "v_pk_add_f32 %2, %5, %7 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" N                                \
                 "v_pk_mul_f32 %0, %0, %0\n\t"                                                                           \
                 "v_pk_mul_f32 %1, %1, %1\n\t"                                                                           \
                 "v_pk_mul_f32 %2, %2, %2\n\t" N                                                                          \
                 "v_pk_add_f32 %0, %0, %1\n\t" N                                                                          \
                 "v_pk_add_f32 %0, %0, %2\n\t" N                                                                          \
                 : "=&v"(dx), "=&v"(dy), "=&v"(dz) : "v"(x), "v"(y), "v"(z), "v"(s), "v"(t))
    if (MODE == 200) {          // every broadcast from a LOW half (what hipcc itself emits)
        f2 sy2 = {s.y, 0.f};
        asm volatile("v_pk_add_f32 %0, %3, %6 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_add_f32 %1, %4, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_add_f32 %2, %5, %7 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_mul_f32 %0, %0, %0\n\tv_pk_mul_f32 %1, %1, %1\n\tv_pk_mul_f32 %2, %2, %2\n\t"
                     "v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\t"
                     : "=&v"(dx), "=&v"(dy), "=&v"(dz) : "v"(x), "v"(y), "v"(z), "v"(s), "v"(t), "v"(sy2));
        return dx;
    }
    if (MODE == 201) {          // every broadcast from a HIGH half
        f2 sx2 = {0.f, s.x}, sz2 = {0.f, t.x};
        asm volatile("v_pk_add_f32 %0, %3, %8 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_add_f32 %1, %4, %6 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_add_f32 %2, %5, %9 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_mul_f32 %0, %0, %0\n\tv_pk_mul_f32 %1, %1, %1\n\tv_pk_mul_f32 %2, %2, %2\n\t"
                     "v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\t"
                     : "=&v"(dx), "=&v"(dy), "=&v"(dz) : "v"(x), "v"(y), "v"(z), "v"(s), "v"(t), "v"(sx2), "v"(sz2));
        return dx;
    }
    if (MODE == 202) {          // low-half broadcasts WITHOUT the neg modifiers (the sample negated beforehand)
        f2 nx = {-s.x, 0.f}, ny = {-s.y, 0.f}, nz = {-t.x, 0.f};
        asm volatile("v_pk_add_f32 %0, %3, %6 op_sel_hi:[1,0]\n\t"
                     "v_pk_add_f32 %1, %4, %7 op_sel_hi:[1,0]\n\t"
                     "v_pk_add_f32 %2, %5, %8 op_sel_hi:[1,0]\n\t"
                     "v_pk_mul_f32 %0, %0, %0\n\tv_pk_mul_f32 %1, %1, %1\n\tv_pk_mul_f32 %2, %2, %2\n\t"
                     "v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\t"
                     : "=&v"(dx), "=&v"(dy), "=&v"(dz) : "v"(x), "v"(y), "v"(z), "v"(nx), "v"(ny), "v"(nz));
        return dx;
    }
    if (MODE == 0) CHAIN(NOPSTR_0);
    if (MODE == 1) CHAIN(NOPSTR_1);
    if (MODE == 2) CHAIN(NOPSTR_2);
    if (MODE == 3) CHAIN(NOPSTR_3);
    if (MODE == 5) CHAIN(NOPSTR_5);
    if (MODE == 8) CHAIN(NOPSTR_8);
    if (MODE == 16) CHAIN(NOPSTR_16);
    return dx;
}

__device__ __forceinline__ float sc(float x, float y, float z, float sx, float sy, float sz)
{
    const float dx = __fsub_rn(x, sx), dy = __fsub_rn(y, sy), dz = __fsub_rn(z, sz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long *bad, int iters)
{
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    f2 x = {rnd(), rnd()}, y = {rnd(), rnd()}, z = {rnd(), rnd()};
    float m0 = 1e38f, m1 = 1e38f, n0 = 1e38f, n1 = 1e38f;
    unsigned long long cnt = 0;
    for (int i = 0; i < iters; ++i) {
        f2 sxy = {rnd(), rnd()}, szk = {rnd(), 0.f};
        const f2 r = chain<MODE>(x, y, z, sxy, szk);
        float a0, a1;
        asm volatile("v_min_f32 %0, %1, %2" : "=v"(a0) : "v"(r.x), "v"(m0));
        asm volatile("v_min_f32 %0, %1, %2" : "=v"(a1) : "v"(r.y), "v"(m1));
        m0 = a0; m1 = a1;
        const float c0 = sc(x.x, y.x, z.x, sxy.x, sxy.y, szk.x), c1 = sc(x.y, y.y, z.y, sxy.x, sxy.y, szk.x);
        n0 = fminf(c0, n0); n1 = fminf(c1, n1);
        cnt += (__float_as_uint(m0) != __float_as_uint(n0)) + (__float_as_uint(m1) != __float_as_uint(n1));
        if ((i & 63) == 63) { m0 = m1 = n0 = n1 = 1e38f; }
    }
    if (cnt) atomicAdd(bad, cnt);
}

template <int MODE>
static void run(const char *name, unsigned long long *d_bad, float *sink, hipStream_t sv, hipStream_t sa)
{
    for (int ag = 0; ag < 3; ++ag) {
        CK(hipMemset(d_bad, 0, 8));
        for (int r = 0; r < 20; ++r) {
            if (ag == 1) hipLaunchKernelGGL(ag_mfma, dim3(2048), dim3(256), 0, sa, sink, 4000);
            if (ag == 2) hipLaunchKernelGGL(ag_valu, dim3(2048), dim3(256), 0, sa, sink, 20000);
            hipLaunchKernelGGL(probe<MODE>, dim3(64), dim3(256), 0, sv, d_bad, 2000);
            if (ag == 1) hipLaunchKernelGGL(ag_mfma, dim3(2048), dim3(256), 0, sa, sink, 4000);
        }
        CK(hipDeviceSynchronize());
        unsigned long long h; CK(hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost));
        printf("%-28s %-10s: %llu lanes disagree with the scalar evaluation (of %llu)\n", name, ag == 0 ? "alone" : ag == 1 ? "beside mfma" : "beside valu", h,
               20ull * 64 * 256 * 2000 * 2);
    }
}

int main()
{
    unsigned long long *d_bad; float *sink;
    CK(hipMalloc(&d_bad, 8)); CK(hipMalloc(&sink, 64));
    hipStream_t sv, sa; CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sa));
    run<100>("compiler-generated v_pk", d_bad, sink, sv, sa);
    run<200>("asm, low-half broadcasts", d_bad, sink, sv, sa);
    run<201>("asm, high-half broadcasts", d_bad, sink, sv, sa);
    run<202>("asm, low-half, no neg", d_bad, sink, sv, sa);
    run<0>("asm, no wait states", d_bad, sink, sv, sa);
    run<1>("asm, 1 wait state", d_bad, sink, sv, sa);
    return 0;
}
