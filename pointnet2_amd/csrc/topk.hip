// topk.hip -- selection_sort (select_top_k) for gfx950.
//
// Replaces selection_sort_gpu / selectionSortLauncher (reference
// tf_ops/grouping/tf_grouping_g.cu:83-123, :129-132; CPU twin
// test/selection_sort.cpp:20-63). The op's contract is the FULL (b,m,n) pair
// of outputs: a copy of each distance row and an iota, on which k rounds of
// "swap the first minimum of the unsorted tail into slot s" have been applied.
// The swaps permute the tail, so the whole row -- not just its first k
// entries -- is reproduced exactly.
//
// Design. The reference gives a row to ONE thread (O(k*n) serial global-memory
// traffic per row). Here a 64-lane wave owns a row held in LDS: each round every
// lane scans a strided slice of the tail keeping its first minimum, the wave
// reduces (value, index) keys with a 64-bit min, and lane 0 swaps. Rows longer
// than the LDS tier fall back to a one-thread-per-row kernel.
#include "pn2_device.h"

#include <limits.h>

namespace pn2 {

constexpr int kSortMaxLdsN = 16384;   // 8 B per element -> 128 KiB

// map a float to a uint whose unsigned order equals the float `<` order;
// +0 and -0 compare equal under `<`, so both map to the same key.
__device__ __forceinline__ unsigned orderable(float f)
{
    if (f == 0.0f) f = 0.0f;   // folds -0 into +0
    const unsigned u = (unsigned)__float_as_int(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(64) void selection_sort_wave_kernel(int n, int k, const float *__restrict__ dist,
                                                                 int *__restrict__ outi, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *val = reinterpret_cast<float *>(smem);
    int *ind = reinterpret_cast<int *>(smem + sizeof(float) * (size_t)n);
    const size_t row = blockIdx.x;
    const int lane = threadIdx.x;
    const float *src = dist + row * n;
    for (int s = lane; s < n; s += 64) { val[s] = src[s]; ind[s] = s; }
    __syncthreads();
    const int rounds = min(k, n);
    for (int s = 0; s < rounds; ++s) {
        // first minimum of val[s..n): key = (orderable value, position), 64-bit min
        unsigned long long best = ~0ull;
        for (int t = s + lane; t < n; t += 64) {
            const unsigned long long key = ((unsigned long long)orderable(val[t]) << 32) | (unsigned)t;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)best, o, 64);
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), o, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            best = other < best ? other : best;
        }
        const int mn = (int)(unsigned)best;
        if (lane == 0 && mn != s) {
            const float tv = val[mn]; val[mn] = val[s]; val[s] = tv;
            const int ti = ind[mn]; ind[mn] = ind[s]; ind[s] = ti;
        }
        __syncthreads();
    }
    for (int s = lane; s < n; s += 64) { out[row * n + s] = val[s]; outi[row * n + s] = ind[s]; }
}

// knn_point without the (b, m, n) tensors. The reference builds the pairwise
// squared-distance matrix in TF (tf_grouping.py:57-65: tile, subtract, square, reduce_sum -- three
// (b,m,n,c)/(b,m,n) tensors, ~0.5 GB each at the metric shape), runs the selection sort above on it and
// slices the first k columns. Here a wave computes its query's distance row straight into LDS
// (((dx*dx)+(dy*dy))+(dz*dz), d = xyz1 - xyz2, no FMA: the values the elementwise graph produces), runs
// the SAME k swap rounds on it -- so ties come out in the reference's swap-dependent order, not in index
// order -- and writes only the k results.
__global__ __launch_bounds__(64) void knn_wave_kernel(int n, int m, int k, const float *__restrict__ xyz1,
                                                      const float *__restrict__ xyz2, float *__restrict__ oval,
                                                      int *__restrict__ oidx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *val = reinterpret_cast<float *>(smem);
    int *ind = reinterpret_cast<int *>(smem + sizeof(float) * (size_t)n);
    const size_t row = blockIdx.x;                               // query number over all clouds
    const size_t cloud = row / m;
    const int lane = threadIdx.x;
    const float *pts = xyz1 + cloud * n * 3;
    const float qx = xyz2[row * 3 + 0], qy = xyz2[row * 3 + 1], qz = xyz2[row * 3 + 2];
    for (int s = lane; s < n; s += 64) {
        val[s] = sqdist(pts[s * 3 + 0], pts[s * 3 + 1], pts[s * 3 + 2], qx, qy, qz);
        ind[s] = s;
    }
    __syncthreads();
    const int rounds = min(k, n);
    for (int s = 0; s < rounds; ++s) {
        unsigned long long best = ~0ull;
        for (int t = s + lane; t < n; t += 64) {
            const unsigned long long key = ((unsigned long long)orderable(val[t]) << 32) | (unsigned)t;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)best, o, 64);
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), o, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            best = other < best ? other : best;
        }
        const int mn = (int)(unsigned)best;
        if (lane == 0 && mn != s) {
            const float tv = val[mn]; val[mn] = val[s]; val[s] = tv;
            const int ti = ind[mn]; ind[mn] = ind[s]; ind[s] = ti;
        }
        __syncthreads();
    }
    for (int s = lane; s < k; s += 64) {                         // k > n: the tail is the untouched row, as in the slice
        oval[row * k + s] = s < n ? val[s] : 0.0f;
        oidx[row * k + s] = s < n ? ind[s] : 0;
    }
}

// fallback for very long rows: the reference's own mapping (one thread per row)
__global__ __launch_bounds__(64) void selection_sort_serial_kernel(long long rows, int n, int k,
                                                                   const float *__restrict__ dist,
                                                                   int *__restrict__ outi, float *__restrict__ out)
{
    const long long row = (long long)blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const float *src = dist + row * n;
    float *v = out + row * n;
    int *vi = outi + row * n;
    for (int s = 0; s < n; ++s) { v[s] = src[s]; vi[s] = s; }
    for (int s = 0; s < k && s < n; ++s) {
        int mn = s;
        float mv = v[s];
        for (int t = s + 1; t < n; ++t) {
            const float c = v[t];
            if (c < mv) { mv = c; mn = t; }
        }
        if (mn != s) {
            v[mn] = v[s]; v[s] = mv;
            const int ti = vi[mn]; vi[mn] = vi[s]; vi[s] = ti;
        }
    }
}

}  // namespace pn2

extern "C" int pn2_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream)
{
    using namespace pn2;
    if (k <= 0) return PN2_E_ARG;                       // tf_grouping.cpp:113
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    const long long rows = (long long)b * m;
    if (rows == 0) return PN2_OK;
    if (!dist || !outi || !out) return PN2_E_NULL;
    if (rows > INT_MAX) return PN2_E_TOO_LARGE;
    hipStream_t st = as_stream(stream);
    if (n <= kSortMaxLdsN) {
        const size_t lds = 8 * (size_t)n;
        auto kern = selection_sort_wave_kernel;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(64), lds, st, n, k, dist, outi, out);
    } else {
        hipLaunchKernelGGL(selection_sort_serial_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, rows, n,
                           k, dist, outi, out);
    }
    return launch_status();
}

extern "C" int pn2_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                             void *stream)
{
    using namespace pn2;
    if (k <= 0) return PN2_E_ARG;                       // tf_grouping.cpp:113
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    const long long rows = (long long)b * m;
    if (rows == 0) return PN2_OK;
    if (!xyz1 || !xyz2 || !val || !idx) return PN2_E_NULL;
    if (rows > INT_MAX || k > n) return PN2_E_TOO_LARGE;   // k > n: tf.slice would fail in the reference as well
    if (n > kSortMaxLdsN) return PN2_E_TOO_LARGE;           // callers keep the matrix + pn2_selection_sort path
    const size_t lds = 8 * (size_t)n;
    auto kern = knn_wave_kernel;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(64), lds, as_stream(stream), n, m, k, xyz1, xyz2, val, idx);
    return launch_status();
}
