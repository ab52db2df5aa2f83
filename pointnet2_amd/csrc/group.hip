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

// c == 3, rows_per_cloud % 4 == 0: a lane gathers FOUR rows (one 16-byte load of their indices, four
// 12-byte row loads) and writes them as three 16-byte stores = 48 contiguous bytes per lane, instead of
// one 12-byte row per lane (global_store_dwordx3 rows straddle the 16-byte sectors).
template <bool NT>
__global__ __launch_bounds__(kThreads) void group_rows_c3x4_kernel(int quads_per_cloud, int n, int quads_per_part,
                                                                   int parts, int b, const float *__restrict__ points,
                                                                   const int4 *__restrict__ idx4,
                                                                   float4 *__restrict__ out4)
{
    int cloud, part;
    decode_cloud_block(blockIdx.x, parts, b, cloud, part);
    const int qb = part * quads_per_part, qe = min(qb + quads_per_part, quads_per_cloud);
    const int4 *__restrict__ idc = idx4 + (size_t)cloud * quads_per_cloud;
    const float *__restrict__ src = points + (size_t)cloud * n * 3;
    float4 *__restrict__ dst = out4 + (size_t)cloud * quads_per_cloud * 3;
    for (int q = qb + (int)threadIdx.x; q < qe; q += kThreads) {
        const int4 k = idc[q];
        const float *pa = src + (unsigned)k.x * 3u, *pb = src + (unsigned)k.y * 3u;
        const float *pc = src + (unsigned)k.z * 3u, *pd = src + (unsigned)k.w * 3u;
        const float a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
        const float c0 = pc[0], c1 = pc[1], c2 = pc[2], d0 = pd[0], d1 = pd[1], d2 = pd[2];
        float4 *o = dst + (unsigned)q * 3u;
        const float4 v0 = make_float4(a0, a1, a2, b0), v1 = make_float4(b1, b2, c0, c1), v2 = make_float4(c2, d0, d1, d2);
        if (NT) {
            store_nt(o + 0, v0); store_nt(o + 1, v1); store_nt(o + 2, v2);
        } else {
            o[0] = v0; o[1] = v1; o[2] = v2;
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
    if (variant != 1 && fits && c == 3 && (rpc & 3) == 0 && aligned16(idx) && aligned16(out)) {
        const int quads = (int)(rpc / 4);
        const int parts = parts_for(b, quads, kThreads);
        const int qpp = (quads + parts - 1) / parts;
        const bool nt = variant == 3;
        auto kern = nt ? group_rows_c3x4_kernel<true> : group_rows_c3x4_kernel<false>;
        return launch(kern, dim3((unsigned)parts * b), dim3(kThreads), 0, st, quads, n, qpp, parts, b, points,
                      reinterpret_cast<const int4 *>(idx), reinterpret_cast<float4 *>(out));
    }
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
    {
        const int rc = group_rows(b, n, 3, m, inp, idx, out, 0, as_stream(stream));     // gather = group with nsample 1
        if (rc != -1000) return rc;
    }
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
    hipError_t e = hipMemsetAsync(inp_g, 0, sizeof(float) * (size_t)b * n * 3, st);   // tf_sampling.cpp:174
    if (e != hipSuccess) return (int)e;
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
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, st);   // tf_grouping.cpp:204
    if (e != hipSuccess) return (int)e;
    const long long rpc = (long long)m * nsample;
    const long long elems = (long long)b * rpc * c;
    if (elems == 0) return PN2_OK;
    if (!grad_out || !idx) return PN2_E_NULL;
    if (int rc = launch(group_point_grad_kernel, dim3(grid_for(elems)), dim3(kThreads), 0, st, elems, rpc, n, c,
                       grad_out, idx, grad_points)) return rc;
    return PN2_OK;
}
