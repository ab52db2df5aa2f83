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

// ---- bandwidth kernels, second generation ------------------------------------------------------
// The flat kernels above pay one or two 64-bit integer divisions per element (row = e / c4, cloud =
// row / rows_per_cloud) and keep one load in flight per lane: 32-49 % of the HBM peak at the reference
// configurations. Here the grid is (parts per cloud) x (clouds), decoded so that the
// workgroups of one cloud share an XCD (decode_cloud_block: the gathered source rows are re-read
// nsample*m/n times and then come from that XCD's L2); a thread derives (row, chunk) ONCE and advances
// them by adding constants; and U independent idx -> row -> store chains are in flight per lane.
// Outputs larger than the 256 MB Infinity Cache are written with non-temporal stores (measured on MI355X,
// scripts/bw_probe.py: (32,4096,128)->(32,1024,32,128), 608 MB by SURVEY 8(d): 171 us flat kernel, 156 us
// row kernel, 102 us = 5.97 TB/s with nt stores; (32,512,320)->(32,128,128,320): 180 / 166 / 129 us; a
// 144 MB output that stays in the cache prefers plain stores: 25.5 / 22.7 / 25.6 us).
// c == 3 keeps the flat one-row-per-lane kernel: a four-rows-per-lane variant (one 16-byte index load, three
// 16-byte stores) measured SLOWER (7.6 vs 6.6 us at the metric shape; the launch is latency bound: 18 MB
// is 2.3 us of HBM time, less than two kernel-launch floors). Two and four rows per lane with every index load
// issued before the first gather: 6.3 / 6.5 us -- no change either).

typedef float pn2_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(float4 *p, float4 v)      // global_store_dwordx4 ... nt
{
    pn2_v4f w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<pn2_v4f *>(p));
}

// c % 4 == 0. Thread t of a workgroup starts at chunk t of the workgroup's row range (chunk = 16 B of an
// output row, c4 chunks per row) and steps by 256 chunks = (dr rows, dl chunks).
template <int U, bool NT>
__global__ __launch_bounds__(kThreads) void group_rows_v4_kernel(int rows_per_cloud, int n, int c4, int dr, int dl,
                                                                 int rows_per_part, int parts, int b,
                                                                 const float4 *__restrict__ points,
                                                                 const int *__restrict__ idx, float4 *__restrict__ out)
{
    int cloud, part;
    decode_cloud_block(blockIdx.x, parts, b, cloud, part);
    const int rb = part * rows_per_part, re = min(rb + rows_per_part, rows_per_cloud);
    const int *__restrict__ idc = idx + (size_t)cloud * rows_per_cloud;
    const float4 *__restrict__ src = points + (size_t)cloud * n * c4;
    float4 *__restrict__ dst = out + (size_t)cloud * rows_per_cloud * c4;
    int r = rb + (int)threadIdx.x / c4, l = (int)threadIdx.x % c4;
    while (r < re) {
        int rr[U], ll[U], k[U];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rr[u] = r; ll[u] = l;
            l += dl; r += dr;
            if (l >= c4) { l -= c4; ++r; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) k[u] = rr[u] < re ? idc[rr[u]] : 0;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[(unsigned)k[u] * (unsigned)c4 + (unsigned)ll[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rr[u] < re) {
                float4 *o = dst + ((unsigned)rr[u] * (unsigned)c4 + (unsigned)ll[u]);
                if (NT) store_nt(o, v[u]);
                else *o = v[u];
            }
        }
    }
}

// Parts per cloud: enough workgroups to fill the chip several times over (256 CUs x 8 resident workgroups
// of 256 threads), but at least `min_units` units (rows / quads) per workgroup.
static inline int parts_for(int b, int units_per_cloud, int min_units)
{
    int parts = (4096 + b - 1) / b;
    const int most = (units_per_cloud + min_units - 1) / min_units;
    if (parts > most) parts = most;
    return parts < 1 ? 1 : parts;
}

// variant: 0 automatic, 1 first-generation flat kernels, 2 row kernels, 3 row kernels with non-temporal stores
static int group_rows(int b, int n, int c, long long rpc, const float *points, const int *idx, float *out, int variant,
                      hipStream_t st)
{
    const bool fits = rpc <= INT_MAX / 2 && (long long)n * c < (1ll << 31) && rpc * c < (1ll << 31) &&
                      (long long)b * 4096 < INT_MAX && true;
    if (variant != 1 && fits && c % 4 == 0 && aligned16(points) && aligned16(out)) {
        constexpr int U = 4;
        const int c4 = c / 4;
        const int rows_min = (kThreads * U * 2 + c4 - 1) / c4;          // two trips of U chunks per thread
        const int parts = parts_for(b, (int)rpc, rows_min);
        const int rpp = (int)((rpc + parts - 1) / parts);
        const bool nt = variant == 3 || (variant == 0 && (long long)b * rpc * c * 4 > (192ll << 20));
        auto kern = nt ? group_rows_v4_kernel<U, true> : group_rows_v4_kernel<U, false>;
        return launch(kern, dim3((unsigned)parts * b), dim3(kThreads), 0, st, (int)rpc, n, c4, kThreads / c4, kThreads % c4,
                      rpp, parts, b, reinterpret_cast<const float4 *>(points), idx, reinterpret_cast<float4 *>(out));
    }
    return -1000;                                                       // not handled here
}

}  // namespace pn2

extern "C" int pn2_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    if (b == 0 || m == 0) return PN2_OK;
    if (!inp || !idx || !out) return PN2_E_NULL;
    const long long rows = (long long)b * m;
    if (int rc = launch(gather_point_kernel, dim3(grid_for(rows)), dim3(kThreads), 0, as_stream(stream), rows, n, m,
                       inp, idx, out)) return rc;
    return PN2_OK;
}

extern "C" int pn2_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                                     void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!inp_g) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    if (int rc = clear_async(inp_g, sizeof(float) * (size_t)b * n * 3, st)) return rc;   // tf_sampling.cpp:174
    if (m == 0) return PN2_OK;
    if (!out_g || !idx) return PN2_E_NULL;
    const long long rows = (long long)b * m;
    if (int rc = launch(gather_point_grad_kernel, dim3(grid_for(rows)), dim3(kThreads), 0, st, rows, n, m, out_g, idx,
                       inp_g)) return rc;
    return PN2_OK;
}

static int group_point_entry(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out,
                             int variant, void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return PN2_E_SHAPE;
    if (variant < 0 || variant > 3) return PN2_E_ARG;
    const long long rpc = (long long)m * nsample;
    const long long rows = (long long)b * rpc;
    if (rows == 0) return PN2_OK;
    if (!points || !idx || !out) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    {
        const int rc = group_rows(b, n, c, rpc, points, idx, out, variant, st);
        if (rc != -1000) return rc;
    }
    if (c == 3) {
        if (int rc = launch(group_point_c3_kernel, dim3(grid_for(rows)), dim3(kThreads), 0, st, rows, rpc, n, points,
                           idx, out)) return rc;
    } else if (c % 4 == 0 && aligned16(points) && aligned16(out)) {
        const long long chunks = rows * (c / 4);
        if (int rc = launch(group_point_v4_kernel, dim3(grid_for(chunks)), dim3(kThreads), 0, st, chunks, rpc, n, c / 4,
                           reinterpret_cast<const float4 *>(points), idx, reinterpret_cast<float4 *>(out))) return rc;
    } else {
        const long long elems = rows * c;
        if (int rc = launch(group_point_s_kernel, dim3(grid_for(elems)), dim3(kThreads), 0, st, elems, rpc, n, c,
                           points, idx, out)) return rc;
    }
    return PN2_OK;
}

extern "C" int pn2_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                               float *out, void *stream)
{
    return group_point_entry(b, n, c, m, nsample, points, idx, out, 0, stream);
}

// pn2_group_point with the kernel choice as a per-call argument (tests force every kernel; scripts time them):
// 0 automatic, 1 flat first-generation kernels, 2 row kernels, 3 row kernels with non-temporal stores.
extern "C" int pn2_group_point_ex(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                                  float *out, int variant, void *stream)
{
    return group_point_entry(b, n, c, m, nsample, points, idx, out, variant, stream);
}

extern "C" int pn2_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                                    float *grad_points, void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!grad_points) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    if (int rc = clear_async(grad_points, sizeof(float) * (size_t)b * n * c, st)) return rc;   // tf_grouping.cpp:204
    const long long rpc = (long long)m * nsample;
    const long long elems = (long long)b * rpc * c;
    if (elems == 0) return PN2_OK;
    if (!grad_out || !idx) return PN2_E_NULL;
    if (int rc = launch(group_point_grad_kernel, dim3(grid_for(elems)), dim3(kThreads), 0, st, elems, rpc, n, c,
                       grad_out, idx, grad_points)) return rc;
    return PN2_OK;
}
