// sa_mlp_common.h -- helpers shared by the two fused-MLP kernels (sa_mlp.hip: weights resident in LDS;
// sa_mlp_stream.hip: weights streamed through LDS). See sa_mlp.hip for the formulation.
#pragma once
#include "pn2_device.h"

#include <math.h>

namespace pn2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMlpThreads = 256;          // 4 waves: one per SIMD; a second workgroup shares the CU when registers allow
constexpr int kMlpMaxLds = 150 * 1024;

// channel (within a 32-tile) that register v of lane-half h holds / must be fed with
__host__ __device__ __forceinline__ int mlp_chan(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

// packed sizes (floats): weights [t][u][q = 4][lane = 64][r = 4] per layer, bias [t][h = 2][v = 16]
__host__ __device__ __forceinline__ size_t mlp_w_floats(int t_out, int t_in) { return (size_t)t_out * t_in * 1024; }
__host__ __device__ __forceinline__ size_t mlp_b_floats(int t_out) { return (size_t)t_out * 32; }

__device__ __forceinline__ f32x16 mlp_bias(const float *bp, int t, int h)
{
    const float4 *p = reinterpret_cast<const float4 *>(bp + (t * 2 + h) * 16);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    f32x16 r = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    return r;
}

// packed bias [t][h][v] holds channel 32t + mlp_chan(v, h); the inverse for one channel
__device__ __forceinline__ float b3_at(const float *bp, int ch)
{
    const int t = ch >> 5, c = ch & 31;
    const int hh = (c >> 2) & 1, v = 4 * (c >> 3) + (c & 3);
    return bp[(t * 2 + hh) * 16 + v];
}

__device__ __forceinline__ f32x16 mlp_relu(f32x16 x)
{
#pragma unroll
    for (int v = 0; v < 16; ++v) x[v] = fmaxf(x[v], 0.0f);
    return x;
}


// ---- streamed variant (sa_mlp_stream.hip) ------------------------------------------------------------
constexpr int kMlpStagePairs = 4;          // 32x32 weight tile pairs per LDS stage (16 KiB)

constexpr int kS = kMlpStagePairs;

__host__ __device__ __forceinline__ int pad_to_stage(int pairs) { return (pairs + kS - 1) / kS * kS; }

// 16 MFMAs of one tile pair out of the current LDS stage; SWAP: operands exchanged (last layer)
template <bool SWAP>
__device__ __forceinline__ f32x16 stream_pair(const float4 *stage, int slot, int lane, f32x16 act, f32x16 acc)
{
    const float4 *w4 = stage + slot * 256 + lane;
    const float4 a0 = w4[0], a1 = w4[64], a2 = w4[128], a3 = w4[192];
    const float wv[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
#pragma unroll
    for (int v = 0; v < 16; ++v)
        acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(act[v], wv[v], acc, 0, 0, 0)
                   : __builtin_amdgcn_mfma_f32_32x32x2f32(wv[v], act[v], acc, 0, 0, 0);
    return acc;
}

// one 32x32 tile pair of a (kin, nout) row-major weight matrix in the MFMA operand layout (sa_mlp_stream.hip)
float *mlp_pack_pair(float *wp, const float *w, int kin, int nout, int t, int u, const int *krow);

struct MlpStreamConfig { int ti, t1, t2, t3; };
bool mlp_stream_pick(int cin, int c1, int c2, int c3, MlpStreamConfig &cfg);
size_t mlp_stream_w_floats(const MlpStreamConfig &c);
size_t mlp_stream_b_floats(const MlpStreamConfig &c);
void mlp_stream_pack(const MlpStreamConfig &c, int cin, int c1, int c2, int c3, int xyz_first, const float *const *ws,
                     const float *const *bs, float *wpacked, float *bpacked);
int mlp_stream_launch(const MlpStreamConfig &c, int b, int n, int m, int nsample, int cfeat, int c3, const float *xyz,
                      const float *new_xyz, const float *points, const int *idx, const float *wp, const float *bp,
                      float *out, hipStream_t st);

// ---- cooperative variant (coop_mlp.hip): four waves share one 32-sample item and split every layer's output
// tiles; wide stacks over few rows (SA levels beyond (128,128,256), group_all levels, small FP levels) -------
struct MlpCoopConfig { int ti, q1, q2, q3; };          // input tiles; output tiles PER WAVE of the layers (q3 = 0: two layers)
struct CoopParams {
    int n, m, nsample, cf, c1, cout, ti;              // cf: grouped feature channels (SA) / interpolated channels c2 (FP)
    long long rows;                                   // SA: b * m centroids; FP: b * n unknown points
    const float *xyz, *new_xyz;                       // SA (new_xyz == nullptr: group_all, no centroid)
    const float *feat;                                // SA: points (b,n,cf); FP: points2 (b,m,cf)
    const float *skip;                                // FP: points1 (b,n,c1) or nullptr
    const int *idx;                                   // SA: (b,m,nsample) or nullptr (group_all); FP: (b,n,3)
    const float *dist;                                // FP: (b,n,3)
    const float *wp, *bp;
    float *out;
    int split;                                        // set by mlp_coop_launch
};
bool mlp_coop_pick(int cin, int nlayers, const int *widths, MlpCoopConfig &cfg);
bool mlp_coop_has_kernel(const MlpCoopConfig &c, int fp);
size_t mlp_coop_w_floats(const MlpCoopConfig &c);
size_t mlp_coop_b_floats(const MlpCoopConfig &c);
void mlp_coop_pack(const MlpCoopConfig &c, int cin, int nlayers, const int *widths, const int *krow, const float *const *ws,
                   const float *const *bs, float *wpacked, float *bpacked);
int mlp_coop_launch(const MlpCoopConfig &c, int fp, const CoopParams &p, hipStream_t st);

}  // namespace pn2
