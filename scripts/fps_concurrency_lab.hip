// fps_concurrency_lab.hip -- what kind of neighbour disturbs an FPS chain? The product FPS kernels (fps.hip included verbatim)
// run on one stream while a synthetic aggressor runs on another; the indices are compared with the run that had the GPU
// to itself. Development aid for the multi-stream mismatch: scripts/multistream_probe.py showed
// that only the MLP / three_nn jobs disturb the FPS-based jobs, never another geometry job.
#include "../pointnet2_amd/csrc/fps.hip"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ---- aggressors: ~100-200 us each, grid of 1024 x 256 threads ---------------------------------------------------------
__global__ void ag_valu(float *sink, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.9999f + 1e-4f; }
    if (a == 123.456f) sink[0] = a + b;
}
__global__ void ag_lds(float *sink, int iters, unsigned pattern)
{
    extern __shared__ unsigned lds[];
    const int n = 16384;                                          // 64 KB of dynamic LDS, every word written
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) acc += lds[(threadIdx.x * 17 + i * 64) & (n - 1)];
    if (acc == 0x12345u) sink[0] = 1.0f;
}
__global__ __launch_bounds__(256) void ag_vgpr(float *sink, int iters, unsigned pattern)
{
    // ~200 live VGPRs holding `pattern` (whoever takes these registers next finds it there)
    float v[192];
#pragma unroll
    for (int i = 0; i < 192; ++i) v[i] = __uint_as_float(pattern + (i == 191 ? threadIdx.x & 1 : 0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 192; ++i) asm volatile("v_mov_b32 %0, %0" : "+v"(v[i]));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 192; ++i) s += v[i];
    if (s == 123.456f) sink[0] = s;
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void ag_mfma(float *sink, int iters)
{
    f32x16 acc = {0};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {7, 6, 5, 4, 3, 2, 1, 0};
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (acc[0] == 123.456f) sink[0] = acc[1];
}
__global__ void ag_setprio(float *sink, int iters)
{
    __builtin_amdgcn_s_setprio(3);
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.9999f + 1e-4f; }
    if (a == 123.456f) sink[0] = a + b;
}
__global__ void ag_mem(float *buf, size_t n, int iters)
{
    // streams through a 256 MB buffer (L2 / HBM pressure)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0;
    for (int it = 0; it < iters; ++it) { s += buf[i % n]; i += (size_t)gridDim.x * blockDim.x; }
    if (s == 123.456f) buf[0] = s;
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 60;
    float *sink; CK(hipMalloc(&sink, 64));
    float *big; const size_t bign = 64u << 20; CK(hipMalloc(&big, bign * 4)); CK(hipMemset(big, 0, bign * 4));
    hipStream_t sv, sa; CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sa));
    struct Ag { const char *name; std::function<void()> launch; };
    std::vector<Ag> ags = {
        {"none", [&]() {}},
        {"valu spin", [&]() { hipLaunchKernelGGL(ag_valu, dim3(2048), dim3(256), 0, sa, sink, 20000); }},
        {"valu spin prio3", [&]() { hipLaunchKernelGGL(ag_setprio, dim3(2048), dim3(256), 0, sa, sink, 20000); }},
        {"lds 64K zeros", [&]() { hipFuncSetAttribute((const void *)ag_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); hipLaunchKernelGGL(ag_lds, dim3(2048), dim3(256), 65536, sa, sink, 4000, 0u); }},
        {"lds 64K ones", [&]() { hipLaunchKernelGGL(ag_lds, dim3(2048), dim3(256), 65536, sa, sink, 4000, 0xFFFFFFFFu); }},
        {"lds 64K 1e38", [&]() { hipLaunchKernelGGL(ag_lds, dim3(2048), dim3(256), 65536, sa, sink, 4000, 0x7e967699u); }},
        {"vgpr zeros", [&]() { hipLaunchKernelGGL(ag_vgpr, dim3(2048), dim3(256), 0, sa, sink, 100, 0u); }},
        {"vgpr ones", [&]() { hipLaunchKernelGGL(ag_vgpr, dim3(2048), dim3(256), 0, sa, sink, 100, 0xFFFFFFFFu); }},
        {"vgpr 1e38", [&]() { hipLaunchKernelGGL(ag_vgpr, dim3(2048), dim3(256), 0, sa, sink, 100, 0x7e967699u); }},
        {"mfma", [&]() { hipLaunchKernelGGL(ag_mfma, dim3(2048), dim3(256), 0, sa, sink, 4000); }},
        {"mem stream", [&]() { hipLaunchKernelGGL(ag_mem, dim3(4096), dim3(256), 0, sa, big, bign, 64); }},
    };
    struct Vc { const char *name; int n, m, T, P; };     // T = 0: the library's own choice (pruned tier where it applies)
    std::vector<Vc> vcs = {{"256x4 n=1024", 1024, 512, 256, 4}, {"512x2 n=1024", 1024, 512, 512, 2}, {"256x8 n=2048", 2048, 512, 256, 8},
                           {"512x8 n=4096", 4096, 512, 512, 8}, {"256x16 n=4096", 4096, 512, 256, 16}, {"pruned n=4096", 4096, 512, 0, 0},
                           {"1024x4 n=4096", 4096, 256, 1024, 4}};
    const int b = 32;
    for (const Vc &vc : vcs) {
        std::vector<float> h((size_t)b * vc.n * 3);
        uint32_t s = 777u;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
        float *d_xyz; int *d_out;
        CK(hipMalloc(&d_xyz, h.size() * 4)); CK(hipMalloc(&d_out, (size_t)b * vc.m * 4));
        CK(hipMemcpy(d_xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        auto run = [&]() { return vc.T ? pn2_farthest_point_sample_ex(vc.T, vc.P, b, vc.n, vc.m, d_xyz, d_out, sv)
                                       : pn2_farthest_point_sample(b, vc.n, vc.m, d_xyz, nullptr, d_out, sv); };
        std::vector<int> ref((size_t)b * vc.m), got((size_t)b * vc.m);
        if (run()) { printf("%s: launch refused\n", vc.name); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), d_out, ref.size() * 4, hipMemcpyDeviceToHost));
        printf("%-16s:", vc.name);
        for (const Ag &ag : ags) {
            long bad = 0; int badruns = 0;
            for (int r = 0; r < reps; ++r) {
                ag.launch();
                CK(hipMemsetAsync(d_out, 0xff, got.size() * 4, sv));
                run();
                ag.launch();
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
                long d = 0;
                for (size_t i = 0; i < got.size(); ++i) d += got[i] != ref[i];
                bad += d; badruns += d != 0;
            }
            printf("  %s %ld/%d", ag.name, bad, badruns);
        }
        printf("\n");
        CK(hipFree(d_xyz)); CK(hipFree(d_out));
    }
    return 0;
}
