// prob_sample.hip -- multinomial sampling by inverse CDF for gfx950.
//
// Replaces cumsumKernel + binarysearchKernel / probsampleLauncher (reference
// tf_ops/sampling/tf_sampling_g.cu:7-104, :198-201). The op is adjacent to the
// set-abstraction path (same .so, unused by the shipped models).
//
// The cumulative sum is fp32, so its ASSOCIATION ORDER is part of the result:
// the reference sums groups of four serially, combines the group totals with a
// Blelloch up-sweep / down-sweep over an 8192-element tile, adds the exclusive
// group prefix, then adds a running total carried between tiles with a
// compensation term (:81-84). The kernel below keeps exactly that association
// (so the sampled indices are bit-identical to the oracle's restatement) while
// laying the work out for a 1024-thread gfx950 workgroup per row: float4 global
// loads, one LDS array for the per-element partials and one for the group totals
// (no padding: LDS on gfx950 has 64 banks and these strides are conflict-light).
#include "pn2_device.h"

#include <limits.h>

namespace pn2 {

constexpr int kPsThreads = 1024;
constexpr int kPsTile = 8192;           // elements per tile (reference BlockSize*4)
constexpr int kPsGroups = kPsTile / 4;

__global__ __launch_bounds__(kPsThreads) void cumsum_kernel(int n, const float *__restrict__ inp,
                                                            float *__restrict__ out)
{
    __shared__ float part[kPsTile];      // running sums inside each group of four
    __shared__ float tot[kPsGroups];     // group totals -> inclusive group prefix
    const float *src = inp + (size_t)blockIdx.x * n;
    float *dst = out + (size_t)blockIdx.x * n;
    const int t = threadIdx.x;
    float runningsum = 0.0f, carry = 0.0f;
    for (int j = 0; j < n; j += kPsTile) {
        const int cnt = min(n - j, kPsTile);
        const int cnt4 = (cnt + 3) & ~3;
        const int groups = cnt4 >> 2;
        for (int g = t; g < groups; g += kPsThreads) {
            const int k = g * 4;
            if (k + 3 < cnt) {
                const float v1 = src[j + k];
                const float v2 = __fadd_rn(src[j + k + 1], v1);
                const float v3o = src[j + k + 2];
                const float v4 = __fadd_rn(__fadd_rn(src[j + k + 3], v3o), v2);
                const float v3 = __fadd_rn(v3o, v2);
                part[k] = v1; part[k + 1] = v2; part[k + 2] = v3; part[k + 3] = v4;
                tot[g] = v4;
            } else {
                float v = 0.0f;
                for (int k2 = k; k2 < cnt; ++k2) { v = __fadd_rn(v, src[j + k2]); part[k2] = v; }
                for (int k2 = cnt; k2 < cnt4; ++k2) part[k2] = v;
                tot[g] = v;
            }
        }
        // up-sweep
        int u = 0;
        for (; (2 << u) <= groups; ++u) {
            __syncthreads();
            for (int k = t; k < (groups >> (u + 1)); k += kPsThreads) {
                const int hi = (((k << 1) + 2) << u) - 1;
                const int lo = (((k << 1) + 1) << u) - 1;
                tot[hi] = __fadd_rn(tot[hi], tot[lo]);
            }
        }
        // down-sweep
        for (--u; u >= 0; --u) {
            __syncthreads();
            for (int k = t; k < ((groups - (1 << u)) >> (u + 1)); k += kPsThreads) {
                const int hi = (((k << 1) + 3) << u) - 1;
                const int lo = (((k << 1) + 2) << u) - 1;
                tot[hi] = __fadd_rn(tot[hi], tot[lo]);
            }
        }
        __syncthreads();
        for (int k = t; k < cnt; k += kPsThreads) {
            float v = part[k];
            if (k >= 4) v = __fadd_rn(v, tot[(k >> 2) - 1]);
            dst[j + k] = __fadd_rn(v, runningsum);
        }
        const float tsum = __fadd_rn(tot[groups - 1], carry);
        const float r2 = __fadd_rn(runningsum, tsum);
        carry = __fsub_rn(tsum, __fsub_rn(r2, runningsum));
        runningsum = r2;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void inverse_cdf_kernel(int n, int m, int base, const float *__restrict__ cdf,
                                                          const float *__restrict__ query, int *__restrict__ result)
{
    const int i = blockIdx.y;
    const float *row = cdf + (size_t)i * n;
    const float total = row[n - 1];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256) {
        const float q = __fmul_rn(query[(size_t)i * m + j], total);
        int r = n - 1;
        for (int k = base; k >= 1; k >>= 1)
            if (r >= k && row[r - k] >= q) r -= k;
        result[(size_t)i * m + j] = r;
    }
}

}  // namespace pn2

extern "C" int pn2_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out,
                               void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    if (b == 0 || m == 0) return PN2_OK;
    if (!inp_p || !inp_r || !temp || !out) return PN2_E_NULL;
    if (b > 65535) return PN2_E_TOO_LARGE;
    hipStream_t st = as_stream(stream);
    if (int rc = launch(cumsum_kernel, dim3(b), dim3(kPsThreads), 0, st, n, inp_p, temp)) return rc;
    int base = 1;
    while (base < n) base <<= 1;
    const int gx = (m + 255) / 256 > 64 ? 64 : (m + 255) / 256;
    if (int rc = launch(inverse_cdf_kernel, dim3(gx, b), dim3(256), 0, st, n, m, base, temp, inp_r, out)) return rc;
    return PN2_OK;
}
