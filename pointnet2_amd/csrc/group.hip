// group.hip -- gather_point / group_point and their gradients for gfx950.
//
// Replaces gatherpointKernel, scatteraddpointKernel (reference
// tf_ops/sampling/tf_sampling_g.cu:172-192, launched :206-211) and
// group_point_gpu, group_point_grad_gpu (tf_ops/grouping/tf_grouping_g.cu:40-78,
// launched :133-141; CPU twins test/query_ball_point.cpp:52-84).
// Pure copies are bit-exact; the scatter-adds use fp32 hardware atomics, whose
// accumulation order (like the reference's atomicAdd) is not fixed.
//
// Design (DESIGN.md "group"). These are the HBM-bound kernels of the path.
// The reference gives one thread a whole (nsample x c) output tile, so
// neighbouring lanes write nsample*c floats apart. Here the flat OUTPUT index
// is the thread index: consecutive lanes write consecutive addresses
// (16 B/lane when c % 4 == 0, one 12-byte row per lane when c == 3), and the
// gathered source rows (n*c*4 bytes per cloud) are served from L2.
#include "pn2_device.h"

#include <limits.h>

namespace pn2 {

constexpr int kThreads = 256;

static inline unsigned grid_for(long long work, int per_block = kThreads)
{
    long long g = (work + per_block - 1) / per_block;
    const long long cap = 256ll * 32;   // grid-stride beyond 32 blocks per CU
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// out[(i*m+j)*3+:] = inp[(i*n+idx[i*m+j])*3+:]
__global__ __launch_bounds__(kThreads) void gather_point_kernel(long long rows, int n, int m,
                                                                const float *__restrict__ inp,
                                                                const int *__restrict__ idx, float *__restrict__ out)
{
    for (long long r = (long long)blockIdx.x * kThreads + threadIdx.x; r < rows; r += (long long)gridDim.x * kThreads) {
        const long long i = r / m;
        const float *s = inp + (i * n + idx[r]) * 3;
        float *d = out + r * 3;
        const float a = s[0], b = s[1], c = s[2];
        d[0] = a; d[1] = b; d[2] = c;
    }
}

__global__ __launch_bounds__(kThreads) void gather_point_grad_kernel(long long rows, int n, int m,
                                                                     const float *__restrict__ out_g,
                                                                     const int *__restrict__ idx,
                                                                     float *__restrict__ inp_g)
{
    for (long long r = (long long)blockIdx.x * kThreads + threadIdx.x; r < rows; r += (long long)gridDim.x * kThreads) {
        const long long i = r / m;
        float *d = inp_g + (i * n + idx[r]) * 3;
        const float *s = out_g + r * 3;
        atomicAdd(d + 0, s[0]);
        atomicAdd(d + 1, s[1]);
        atomicAdd(d + 2, s[2]);
    }
}

// c == 3: one 12-byte row per lane (global_load/store_dwordx3; a wave writes 768 contiguous bytes)
__global__ __launch_bounds__(kThreads) void group_point_c3_kernel(long long rows, long long rows_per_cloud, int n,
                                                                  const float *__restrict__ points,
                                                                  const int *__restrict__ idx,
                                                                  float *__restrict__ out)
{
    for (long long r = (long long)blockIdx.x * kThreads + threadIdx.x; r < rows; r += (long long)gridDim.x * kThreads) {
        const long long i = r / rows_per_cloud;
        const float *s = points + (i * n + idx[r]) * 3;
        const float a = s[0], b = s[1], c = s[2];
        float *d = out + r * 3;
        d[0] = a; d[1] = b; d[2] = c;
    }
}

// c % 4 == 0: 16 B per lane, c/4 lanes per row
__global__ __launch_bounds__(kThreads) void group_point_v4_kernel(long long chunks, long long rows_per_cloud, int n,
                                                                  int c4, const float4 *__restrict__ points,
                                                                  const int *__restrict__ idx,
                                                                  float4 *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < chunks; e += (long long)gridDim.x * kThreads) {
        const long long r = e / c4;
        const int l = (int)(e - r * c4);
        const long long i = r / rows_per_cloud;
        out[e] = points[(i * n + idx[r]) * c4 + l];
    }
}

// any c: one float per lane
__global__ __launch_bounds__(kThreads) void group_point_s_kernel(long long elems, long long rows_per_cloud, int n,
                                                                 int c, const float *__restrict__ points,
                                                                 const int *__restrict__ idx, float *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < elems; e += (long long)gridDim.x * kThreads) {
        const long long r = e / c;
        const int l = (int)(e - r * c);
        const long long i = r / rows_per_cloud;
        out[e] = points[(i * n + idx[r]) * c + l];
    }
}

__global__ __launch_bounds__(kThreads) void group_point_grad_kernel(long long elems, long long rows_per_cloud, int n,
                                                                    int c, const float *__restrict__ grad_out,
                                                                    const int *__restrict__ idx,
                                                                    float *__restrict__ grad_points)
{
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < elems; e += (long long)gridDim.x * kThreads) {
        const long long r = e / c;
        const int l = (int)(e - r * c);
        const long long i = r / rows_per_cloud;
        atomicAdd(grad_points + (i * n + idx[r]) * c + l, grad_out[e]);
    }
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace pn2

extern "C" int pn2_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    if (b == 0 || m == 0) return PN2_OK;
    if (!inp || !idx || !out) return PN2_E_NULL;
    const long long rows = (long long)b * m;
    hipLaunchKernelGGL(gather_point_kernel, dim3(grid_for(rows)), dim3(kThreads), 0, as_stream(stream), rows, n, m,
                       inp, idx, out);
    return launch_status();
}

extern "C" int pn2_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                                     void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!inp_g) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(inp_g, 0, sizeof(float) * (size_t)b * n * 3, st);   // tf_sampling.cpp:174
    if (e != hipSuccess) return (int)e;
    if (m == 0) return PN2_OK;
    if (!out_g || !idx) return PN2_E_NULL;
    const long long rows = (long long)b * m;
    hipLaunchKernelGGL(gather_point_grad_kernel, dim3(grid_for(rows)), dim3(kThreads), 0, st, rows, n, m, out_g, idx,
                       inp_g);
    return launch_status();
}

extern "C" int pn2_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                               float *out, void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return PN2_E_SHAPE;
    const long long rpc = (long long)m * nsample;
    const long long rows = (long long)b * rpc;
    if (rows == 0) return PN2_OK;
    if (!points || !idx || !out) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    if (c == 3) {
        hipLaunchKernelGGL(group_point_c3_kernel, dim3(grid_for(rows)), dim3(kThreads), 0, st, rows, rpc, n, points,
                           idx, out);
    } else if (c % 4 == 0 && aligned16(points) && aligned16(out)) {
        const long long chunks = rows * (c / 4);
        hipLaunchKernelGGL(group_point_v4_kernel, dim3(grid_for(chunks)), dim3(kThreads), 0, st, chunks, rpc, n, c / 4,
                           reinterpret_cast<const float4 *>(points), idx, reinterpret_cast<float4 *>(out));
    } else {
        const long long elems = rows * c;
        hipLaunchKernelGGL(group_point_s_kernel, dim3(grid_for(elems)), dim3(kThreads), 0, st, elems, rpc, n, c,
                           points, idx, out);
    }
    return launch_status();
}

extern "C" int pn2_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                                    float *grad_points, void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!grad_points) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, st);   // tf_grouping.cpp:204
    if (e != hipSuccess) return (int)e;
    const long long rpc = (long long)m * nsample;
    const long long elems = (long long)b * rpc * c;
    if (elems == 0) return PN2_OK;
    if (!grad_out || !idx) return PN2_E_NULL;
    hipLaunchKernelGGL(group_point_grad_kernel, dim3(grid_for(elems)), dim3(kThreads), 0, st, elems, rpc, n, c,
                       grad_out, idx, grad_points);
    return launch_status();
}
