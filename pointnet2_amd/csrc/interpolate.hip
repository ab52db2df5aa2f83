// interpolate.hip -- three_nn / three_interpolate / three_interpolate_grad for gfx950.
//
// Replaces the reference's CPU-only loops threenn_cpu, threeinterpolate_cpu,
// threeinterpolate_grad_cpu (tf_ops/3d_interpolation/tf_interpolate.cpp:60-103,
// :107-127, :131-153). In the reference these ops force a device->host->device
// round trip inside every feature-propagation layer.
//   * three_nn is index- and bit-exact: same fp32 squared distance
//     ((dx*dx)+(dy*dy))+(dz*dz), same strict-< cascade scanning the known points
//     in ascending order (order key (d,k)), +inf / index 0 for missing
//     neighbours (the reference's (float)1e40).
//   * three_interpolate is bit-exact too: (p1*w1 + p2*w2) + p3*w3, no FMA.
//   * the gradient accumulates with fp32 atomics (order not fixed).
//
// Design (DESIGN.md "three_nn"). FOUR lanes (a DPP quad) per unknown point: the known points are
// staged through LDS in float4 tiles, lane q of the quad scans the q-th quarter of every tile in
// ascending order with the reference's strict-< insertion (four candidates per trip, one wave-
// uniform skip test), and at the end the quad merges its four triples with two quad_perm exchanges
// under the explicit order key (d, k) -- the key the reference's single ascending scan implies.
// The first version used one lane per point: at the largest FP layer (8 x 8192 unknown points) that
// is one wave per SIMD, a pure latency chain (115 us); the quad split quadruples the waves in flight.
#include "pn2_device.h"

#include <limits.h>
#include <math.h>

namespace pn2 {

constexpr int kNnThreads = 256;             // 64 unknown points x 4 lanes
constexpr int kNnPoints = kNnThreads / 4;
constexpr int kNnTile = 2048;               // known points per LDS tile (32 KiB), a multiple of 16

// strict-< insertion of candidate (d, kk) into the ascending triple (tf_interpolate.cpp:74-89), branch-free
__device__ __forceinline__ void nn_insert(float d, int kk, float &b1, float &b2, float &b3, int &i1, int &i2, int &i3)
{
    const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;
    const float nb3 = c2 ? b2 : (c3 ? d : b3);
    const int ni3 = c2 ? i2 : (c3 ? kk : i3);
    const float nb2 = c1 ? b1 : (c2 ? d : b2);
    const int ni2 = c1 ? i1 : (c2 ? kk : i2);
    b1 = c1 ? d : b1;
    i1 = c1 ? kk : i1;
    b2 = nb2; i2 = ni2; b3 = nb3; i3 = ni3;
}

// the same insertion under the explicit key (d, k): used only to merge the four lanes' triples.
// Empty slots are (+inf, 0); a real candidate with d = +inf is never inserted by the reference
// (inf < 1e40 is false), and (inf, k) < (inf, 0) is false here too.
__device__ __forceinline__ bool nn_less(float d, int k, float b, int i) { return d < b || (d == b && k < i); }
__device__ __forceinline__ void nn_insert_lex(float d, int kk, float &b1, float &b2, float &b3, int &i1, int &i2,
                                              int &i3)
{
    const bool c1 = nn_less(d, kk, b1, i1), c2 = nn_less(d, kk, b2, i2), c3 = nn_less(d, kk, b3, i3);
    const float nb3 = c2 ? b2 : (c3 ? d : b3);
    const int ni3 = c2 ? i2 : (c3 ? kk : i3);
    const float nb2 = c1 ? b1 : (c2 ? d : b2);
    const int ni2 = c1 ? i1 : (c2 ? kk : i2);
    b1 = c1 ? d : b1;
    i1 = c1 ? kk : i1;
    b2 = nb2; i2 = ni2; b3 = nb3; i3 = ni3;
}

template <int CTRL>
__device__ __forceinline__ float quad_xchg_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int quad_xchg_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

// merge the partner lane's triple (quad_perm CTRL) into this lane's
template <int CTRL>
__device__ __forceinline__ void nn_merge(float &b1, float &b2, float &b3, int &i1, int &i2, int &i3)
{
    const float o1 = quad_xchg_f<CTRL>(b1), o2 = quad_xchg_f<CTRL>(b2), o3 = quad_xchg_f<CTRL>(b3);
    const int j1 = quad_xchg_i<CTRL>(i1), j2 = quad_xchg_i<CTRL>(i2), j3 = quad_xchg_i<CTRL>(i3);
    nn_insert_lex(o1, j1, b1, b2, b3, i1, i2, i3);
    nn_insert_lex(o2, j2, b1, b2, b3, i1, i2, i3);
    nn_insert_lex(o3, j3, b1, b2, b3, i1, i2, i3);
}

__global__ __launch_bounds__(kNnThreads) void three_nn_kernel(int n, int m, const float *__restrict__ xyz1,
                                                              const float *__restrict__ xyz2,
                                                              float *__restrict__ dist, int *__restrict__ idx)
{
    __shared__ float4 tile[kNnTile];
    const int bi = blockIdx.y;
    const int sub = threadIdx.x & 3;                       // lane of the quad
    const int j = blockIdx.x * kNnPoints + (threadIdx.x >> 2);
    const bool live = j < n;
    const float *u = xyz1 + ((size_t)bi * n + (live ? j : 0)) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float *__restrict__ known = xyz2 + (size_t)bi * m * 3;

    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;   // (float)1e40, tf_interpolate.cpp:67
    int i1 = 0, i2 = 0, i3 = 0;
    for (int base = 0; base < m; base += kNnTile) {
        const int cnt = min(kNnTile, m - base);
        const int cnt16 = (cnt + 15) & ~15;                // four quarters, each a multiple of 4
        const int quarter = cnt16 >> 2;
        __syncthreads();
        for (int k = threadIdx.x; k < cnt16; k += kNnThreads) {
            if (k < cnt) {
                const float *p = known + (size_t)(base + k) * 3;
                tile[k] = make_float4(p[0], p[1], p[2], 0.0f);
            } else {
                tile[k] = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);   // pad: distance +inf (or NaN), never inserted
            }
        }
        __syncthreads();
        // this lane's quarter of the tile, ascending; four known points per trip
        const int k0 = sub * quarter;
        for (int k = k0; k < k0 + quarter; k += 4) {
            const float4 p0 = tile[k], p1 = tile[k + 1], p2 = tile[k + 2], p3 = tile[k + 3];   // 4 addresses per wave
            // (x2-x1)..., x2 the known point (tf_interpolate.cpp:69-73)
            const float d0 = sqdist(p0.x, p0.y, p0.z, ux, uy, uz);
            const float d1 = sqdist(p1.x, p1.y, p1.z, ux, uy, uz);
            const float d2 = sqdist(p2.x, p2.y, p2.z, ux, uy, uz);
            const float d3 = sqdist(p3.x, p3.y, p3.z, ux, uy, uz);
            const bool better = (d0 < b3) | (d1 < b3) | (d2 < b3) | (d3 < b3);
            if (__any(better)) {
                const int kk = base + k;
                nn_insert(d0, kk, b1, b2, b3, i1, i2, i3);       // ascending k within the lane
                nn_insert(d1, kk + 1, b1, b2, b3, i1, i2, i3);
                nn_insert(d2, kk + 2, b1, b2, b3, i1, i2, i3);
                nn_insert(d3, kk + 3, b1, b2, b3, i1, i2, i3);
            }
        }
    }
    nn_merge<0xB1>(b1, b2, b3, i1, i2, i3);                // quad_perm:[1,0,3,2]
    nn_merge<0x4E>(b1, b2, b3, i1, i2, i3);                // quad_perm:[2,3,0,1]
    if (live && sub == 0) {
        float *od = dist + ((size_t)bi * n + j) * 3;
        int *oi = idx + ((size_t)bi * n + j) * 3;
        od[0] = b1; od[1] = b2; od[2] = b3;
        oi[0] = i1; oi[1] = i2; oi[2] = i3;
    }
}

constexpr int kThreads = 256;

static inline unsigned grid_for(long long work)
{
    long long g = (work + kThreads - 1) / kThreads;
    const long long cap = 256ll * 32;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ float interp3(float p1, float p2, float p3, float w1, float w2, float w3)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(p1, w1), __fmul_rn(p2, w2)), __fmul_rn(p3, w3));   // :122
}

// c % 4 == 0: one float4 of the output row per lane
__global__ __launch_bounds__(kThreads) void three_interpolate_v4_kernel(long long chunks, int m, int n, int c4,
                                                                        const float4 *__restrict__ points,
                                                                        const int *__restrict__ idx,
                                                                        const float *__restrict__ weight,
                                                                        float4 *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < chunks; e += (long long)gridDim.x * kThreads) {
        const long long r = e / c4;            // r = i*n + j
        const int l = (int)(e - r * c4);
        const long long i = r / n;
        const int *q = idx + r * 3;
        const float *wq = weight + r * 3;
        const float w1 = wq[0], w2 = wq[1], w3 = wq[2];
        const float4 *base = points + i * m * c4;
        const float4 a = base[(long long)q[0] * c4 + l];
        const float4 b = base[(long long)q[1] * c4 + l];
        const float4 c = base[(long long)q[2] * c4 + l];
        float4 o;
        o.x = interp3(a.x, b.x, c.x, w1, w2, w3);
        o.y = interp3(a.y, b.y, c.y, w1, w2, w3);
        o.z = interp3(a.z, b.z, c.z, w1, w2, w3);
        o.w = interp3(a.w, b.w, c.w, w1, w2, w3);
        out[e] = o;
    }
}

__global__ __launch_bounds__(kThreads) void three_interpolate_s_kernel(long long elems, int m, int n, int c,
                                                                       const float *__restrict__ points,
                                                                       const int *__restrict__ idx,
                                                                       const float *__restrict__ weight,
                                                                       float *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < elems; e += (long long)gridDim.x * kThreads) {
        const long long r = e / c;
        const int l = (int)(e - r * c);
        const long long i = r / n;
        const int *q = idx + r * 3;
        const float *wq = weight + r * 3;
        const float *base = points + i * m * c;
        out[e] = interp3(base[(long long)q[0] * c + l], base[(long long)q[1] * c + l], base[(long long)q[2] * c + l],
                         wq[0], wq[1], wq[2]);
    }
}

__global__ __launch_bounds__(kThreads) void three_interpolate_grad_kernel(long long elems, int m, int n, int c,
                                                                          const float *__restrict__ grad_out,
                                                                          const int *__restrict__ idx,
                                                                          const float *__restrict__ weight,
                                                                          float *__restrict__ grad_points)
{
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < elems; e += (long long)gridDim.x * kThreads) {
        const long long r = e / c;
        const int l = (int)(e - r * c);
        const long long i = r / n;
        const int *q = idx + r * 3;
        const float *wq = weight + r * 3;
        const float g = grad_out[e];
        float *base = grad_points + i * m * c;
        atomicAdd(base + (long long)q[0] * c + l, __fmul_rn(g, wq[0]));   // :146-148
        atomicAdd(base + (long long)q[1] * c + l, __fmul_rn(g, wq[1]));
        atomicAdd(base + (long long)q[2] * c + l, __fmul_rn(g, wq[2]));
    }
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace pn2

extern "C" int pn2_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
                            void *stream)
{
    using namespace pn2;
    if (b < 0 || n < 0 || m < 0) return PN2_E_SHAPE;
    if (b == 0 || n == 0) return PN2_OK;
    if (!xyz1 || !dist || !idx || (m > 0 && !xyz2)) return PN2_E_NULL;
    if (b > 65535) return PN2_E_TOO_LARGE;
    hipLaunchKernelGGL(three_nn_kernel, dim3((n + kNnPoints - 1) / kNnPoints, b), dim3(kNnThreads), 0,
                       as_stream(stream), n, m, xyz1, xyz2, dist, idx);
    return launch_status();
}

extern "C" int pn2_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx,
                                     const float *weight, float *out, void *stream)
{
    using namespace pn2;
    if (b < 0 || m <= 0 || c <= 0 || n < 0) return PN2_E_SHAPE;
    const long long rows = (long long)b * n;
    if (rows == 0) return PN2_OK;
    if (!points || !idx || !weight || !out) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    if (c % 4 == 0 && aligned16(points) && aligned16(out)) {
        const long long chunks = rows * (c / 4);
        hipLaunchKernelGGL(three_interpolate_v4_kernel, dim3(grid_for(chunks)), dim3(kThreads), 0, st, chunks, m, n,
                           c / 4, reinterpret_cast<const float4 *>(points), idx, weight,
                           reinterpret_cast<float4 *>(out));
    } else {
        const long long elems = rows * c;
        hipLaunchKernelGGL(three_interpolate_s_kernel, dim3(grid_for(elems)), dim3(kThreads), 0, st, elems, m, n, c,
                           points, idx, weight, out);
    }
    return launch_status();
}

extern "C" int pn2_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                          const float *weight, float *grad_points, void *stream)
{
    using namespace pn2;
    if (b < 0 || m <= 0 || c <= 0 || n < 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!grad_points) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * m * c, st);   // tf_interpolate.cpp:258
    if (e != hipSuccess) return (int)e;
    const long long elems = (long long)b * n * c;
    if (elems == 0) return PN2_OK;
    if (!grad_out || !idx || !weight) return PN2_E_NULL;
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(grid_for(elems)), dim3(kThreads), 0, st, elems, m, n, c,
                       grad_out, idx, weight, grad_points);
    return launch_status();
}
