// det_grad.hip -- run-to-run reproducible gradients for gather_point / group_point / three_interpolate
// (SURVEY.md section 8, row f3). gfx950.
//
// The reference GPU gradients are scatter-adds with fp32 atomics (tf_sampling_g.cu:182-190,
// tf_grouping_g.cu:60-78); so are pn2_*_grad in group.hip / interpolate.hip. fp32 addition is not
// associative and the order in which atomics land is not fixed, so two runs of the same training step
// can differ in the last bits. Here every addend is converted to a 64-bit FIXED-POINT integer with one
// power-of-two scale per call and accumulated with 64-bit integer atomics: integer addition is
// associative, so the sum does not depend on the order at all, and the result is converted back once.
//   * scale: 2^k with k = 62 - ceil(log2(entries per cloud)) - e, where 2^e bounds every |addend|
//     (from a max-reduction over grad_out, and over weight for three_interpolate): no overflow for
//     any distribution of indices, 2^-(62-log2(entries)) of the largest addend as resolution -- finer
//     than the 2^-24 an fp32 accumulator keeps;
//   * addend * 2^k is exact in fp64 (a power-of-two scale), llrint makes it an integer; the sum is
//     exact; the conversion back rounds twice (int64 -> fp64 -> fp32), both deterministic;
//   * a sort-based segmented reduction would fix the order instead; it needs a sort of b*m*nsample
//     keys per call and degenerates on crowded targets (the reference's dropout augmentation sends
//     87 % of a cloud's samples to ONE point). Integer atomics have neither problem.
// Non-finite gradients (Inf/NaN) cannot be represented: they are detected by the max-reduction and the
// call then falls back to the fp32-atomic accumulation, which propagates them like the reference does.
#include "pn2_device.h"

#include <limits.h>
#include <math.h>

namespace pn2 {

constexpr int kDetThreads = 256;

static inline unsigned det_grid(long long work)
{
    long long g = (work + kDetThreads - 1) / kDetThreads;
    if (g > 256 * 16) g = 256 * 16;
    return (unsigned)(g > 0 ? g : 1);
}

// ws: [0] max |grad_out| bits, [1] max |weight| bits (0 = no weight), [2..3] unused,
//     then the int64 accumulators
struct DetWs {
    unsigned *head;
    unsigned long long *acc;
};
static inline DetWs det_ws(void *ws) { return {reinterpret_cast<unsigned *>(ws), reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ws) + 16)}; }

// max of |x| as raw bits: for non-negative floats the bit patterns order like the values, Inf sorts
// above every finite value and NaN above Inf, so one unsigned max finds "largest, or non-finite"
__global__ __launch_bounds__(kDetThreads) void det_absmax_kernel(const float *__restrict__ x, long long nelem,
                                                                 unsigned *__restrict__ slot)
{
    unsigned v = 0u;
    for (long long e = (long long)blockIdx.x * kDetThreads + threadIdx.x; e < nelem; e += (long long)gridDim.x * kDetThreads)
        v = max(v, __float_as_uint(x[e]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(slot, v);
}

// 2^e > the value with these bits (finite): e = biased exponent - 126 (denormals: 2^-126)
__device__ __forceinline__ int det_exp_bound(unsigned bits) { return (int)(bits >> 23) - 126; }

// shift k, or INT_MIN when an input is non-finite (fall back to fp32 atomics)
__device__ __forceinline__ int det_shift(const unsigned *head, int logcount)
{
    const unsigned g = head[0], w = head[1] ? head[1] : 0x3f800000u;   // no weight (or all-zero weight): bound 1.0
    if (g >= 0x7f800000u || w >= 0x7f800000u) return INT_MIN;
    return 62 - logcount - (det_exp_bound(g) + det_exp_bound(w));
}

__device__ __forceinline__ unsigned long long det_fixed(float a, int k)
{
    return (unsigned long long)__double2ll_rn(ldexp((double)a, k));   // exact product, one rounding to integer
}

__global__ __launch_bounds__(kDetThreads) void det_gather_grad_kernel(long long rows, int n, int m, int logcount,
                                                                      const float *__restrict__ out_g,
                                                                      const int *__restrict__ idx,
                                                                      const unsigned *__restrict__ head,
                                                                      unsigned long long *__restrict__ acc,
                                                                      float *__restrict__ inp_g)
{
    const int k = det_shift(head, logcount);
    for (long long r = (long long)blockIdx.x * kDetThreads + threadIdx.x; r < rows; r += (long long)gridDim.x * kDetThreads) {
        const long long i = r / m;
        const long long dst = (i * n + idx[r]) * 3;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (k == INT_MIN) atomicAdd(inp_g + dst + l, out_g[r * 3 + l]);
            else atomicAdd(acc + dst + l, det_fixed(out_g[r * 3 + l], k));
        }
    }
}

__global__ __launch_bounds__(kDetThreads) void det_group_grad_kernel(long long elems, long long rows_per_cloud, int n, int c,
                                                                     int logcount, const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx,
                                                                     const unsigned *__restrict__ head,
                                                                     unsigned long long *__restrict__ acc,
                                                                     float *__restrict__ grad_points)
{
    const int k = det_shift(head, logcount);
    for (long long e = (long long)blockIdx.x * kDetThreads + threadIdx.x; e < elems; e += (long long)gridDim.x * kDetThreads) {
        const long long r = e / c;
        const int l = (int)(e - r * c);
        const long long i = r / rows_per_cloud;
        const long long dst = (i * n + idx[r]) * c + l;
        if (k == INT_MIN) atomicAdd(grad_points + dst, grad_out[e]);
        else atomicAdd(acc + dst, det_fixed(grad_out[e], k));
    }
}

__global__ __launch_bounds__(kDetThreads) void det_interp_grad_kernel(long long elems, int m, int n, int c, int logcount,
                                                                      const float *__restrict__ grad_out,
                                                                      const int *__restrict__ idx,
                                                                      const float *__restrict__ weight,
                                                                      const unsigned *__restrict__ head,
                                                                      unsigned long long *__restrict__ acc,
                                                                      float *__restrict__ grad_points)
{
    const int k = det_shift(head, logcount);
    for (long long e = (long long)blockIdx.x * kDetThreads + threadIdx.x; e < elems; e += (long long)gridDim.x * kDetThreads) {
        const long long r = e / c;
        const int l = (int)(e - r * c);
        const long long i = r / n;
        const int *q = idx + r * 3;
        const float *wq = weight + r * 3;
        const float g = grad_out[e];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float a = __fmul_rn(g, wq[t]);                    // the reference's addend, tf_interpolate.cpp:146-148
            const long long dst = (i * m + q[t]) * c + l;
            if (k == INT_MIN) atomicAdd(grad_points + dst, a);
            else atomicAdd(acc + dst, det_fixed(a, k));
        }
    }
}

__global__ __launch_bounds__(kDetThreads) void det_convert_kernel(long long nelem, int logcount,
                                                                  const unsigned *__restrict__ head,
                                                                  const unsigned long long *__restrict__ acc,
                                                                  float *__restrict__ out)
{
    const int k = det_shift(head, logcount);
    if (k == INT_MIN) return;                                       // the fp32 fallback wrote `out` itself
    for (long long e = (long long)blockIdx.x * kDetThreads + threadIdx.x; e < nelem; e += (long long)gridDim.x * kDetThreads)
        out[e] = (float)ldexp((double)(long long)acc[e], -k);
}

static int ceil_log2(long long v)
{
    int l = 0;
    while ((1ll << l) < v) ++l;
    return l;
}

// zero the workspace and the output, reduce max |grad_out| (and |weight|)
static int det_prepare(void *ws, long long acc_elems, float *out, const float *grad, long long grad_elems,
                       const float *weight, long long weight_elems, hipStream_t st)
{
    DetWs w = det_ws(ws);
    if (int rc = clear_async(ws, 16 + sizeof(unsigned long long) * (size_t)acc_elems, st)) return rc;
    if (int rc = clear_async(out, sizeof(float) * (size_t)acc_elems, st)) return rc;
    if (grad_elems > 0)
        if (int rc = launch(det_absmax_kernel, dim3(det_grid(grad_elems)), dim3(kDetThreads), 0, st, grad, grad_elems, w.head)) return rc;
    if (weight && weight_elems > 0)
        if (int rc = launch(det_absmax_kernel, dim3(det_grid(weight_elems)), dim3(kDetThreads), 0, st, weight, weight_elems, w.head + 1)) return rc;
    return PN2_OK;
}

}  // namespace pn2

extern "C" long long pn2_det_grad_ws_bytes(int b, int rows, int c)
{
    if (b <= 0 || rows <= 0 || c <= 0) return 16;
    return 16 + (long long)sizeof(unsigned long long) * b * rows * c;
}

extern "C" int pn2_gather_point_grad_det(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                                         void *ws, void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!inp_g || !ws) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    const long long rows = (long long)b * m, accn = (long long)b * n * 3;
    if (m > 0 && (!out_g || !idx)) return PN2_E_NULL;
    int rc = det_prepare(ws, accn, inp_g, out_g, rows * 3, nullptr, 0, st);
    if (rc || m == 0) return rc;
    DetWs w = det_ws(ws);
    const int logc = ceil_log2(m);
    if (int rc = launch(det_gather_grad_kernel, dim3(det_grid(rows)), dim3(kDetThreads), 0, st, rows, n, m, logc, out_g, idx,
                       w.head, w.acc, inp_g)) return rc;
    if (int rc = launch(det_convert_kernel, dim3(det_grid(accn)), dim3(kDetThreads), 0, st, accn, logc, w.head, w.acc, inp_g)) return rc;
    return PN2_OK;
}

extern "C" int pn2_group_point_grad_det(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                                        float *grad_points, void *ws, void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!grad_points || !ws) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    const long long rpc = (long long)m * nsample, elems = (long long)b * rpc * c, accn = (long long)b * n * c;
    if (elems > 0 && (!grad_out || !idx)) return PN2_E_NULL;
    int rc = det_prepare(ws, accn, grad_points, grad_out, elems, nullptr, 0, st);
    if (rc || elems == 0) return rc;
    DetWs w = det_ws(ws);
    const int logc = ceil_log2(rpc);
    if (int rc = launch(det_group_grad_kernel, dim3(det_grid(elems)), dim3(kDetThreads), 0, st, elems, rpc, n, c, logc,
                       grad_out, idx, w.head, w.acc, grad_points)) return rc;
    if (int rc = launch(det_convert_kernel, dim3(det_grid(accn)), dim3(kDetThreads), 0, st, accn, logc, w.head, w.acc,
                       grad_points)) return rc;
    return PN2_OK;
}

extern "C" int pn2_three_interpolate_grad_det(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                              const float *weight, float *grad_points, void *ws, void *stream)
{
    using namespace pn2;
    if (b < 0 || n < 0 || c <= 0 || m <= 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!grad_points || !ws) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    const long long elems = (long long)b * n * c, accn = (long long)b * m * c;
    if (elems > 0 && (!grad_out || !idx || !weight)) return PN2_E_NULL;
    int rc = det_prepare(ws, accn, grad_points, grad_out, elems, weight, (long long)b * n * 3, st);
    if (rc || elems == 0) return rc;
    DetWs w = det_ws(ws);
    const int logc = ceil_log2((long long)n * 3);
    if (int rc = launch(det_interp_grad_kernel, dim3(det_grid(elems)), dim3(kDetThreads), 0, st, elems, m, n, c, logc, grad_out,
                       idx, weight, w.head, w.acc, grad_points)) return rc;
    if (int rc = launch(det_convert_kernel, dim3(det_grid(accn)), dim3(kDetThreads), 0, st, accn, logc, w.head, w.acc,
                       grad_points)) return rc;
    return PN2_OK;
}
