// sa_mlp_common.h -- helpers shared by the fused-MLP kernels (sa_mlp.hip: weights resident in LDS; sa_mlp_stream.hip:
// weights streamed through LDS + the per-point layers; coop_mlp.hip: four waves per item + the last-layer GEMM;
// fp_mlp.hip: feature propagation). See sa_mlp.hip for the formulation and the arithmetic.
#pragma once
#include "pn2_device.h"

#include <math.h>

namespace pn2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMlpThreads = 256;          // 4 waves: one per SIMD; a second workgroup shares the CU when registers allow
constexpr int kMlpMaxLds = 150 * 1024;

// channel (within a 32-tile) that register v of lane-half h holds / must be fed with
__host__ __device__ __forceinline__ int mlp_chan(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

// packed sizes (4-byte words): weights [t][u] tile pairs of kPairWords (resident kernel) per layer, bias [t][h = 2][v = 16]
__host__ __device__ __forceinline__ size_t mlp_w_floats(int t_out, int t_in) { return (size_t)t_out * t_in * 1536; }
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

// (the MLP sources are built with -fno-honor-nans: otherwise hipcc puts a canonicalising v_max x, x, x in front of
// every fmaxf on an MFMA result -- a third of the kernel's VALU work -- while v_max_f32 returns the non-NaN
// operand either way. Not inline asm: the MFMA -> VALU read hazard is only handled for instructions the compiler knows.)
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }

__device__ __forceinline__ f32x16 mlp_relu(f32x16 x)
{
#pragma unroll
    for (int v = 0; v < 16; ++v) x[v] = vmax(x[v], 0.0f);
    return x;
}

// ---- fp32 products on the bf16 matrix pipe (resident kernel; see the header of sa_mlp.hip) ---------------------
// An fp32 value is the exact sum of three bf16 values a1 + a2 + a3 (8 significant bits each, round-to-nearest
// residuals: |a - a1 - a2 - a3| <= 2^-24 |a|). A product a.b is then a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1)
// up to terms of relative size 2^-24: SIX v_mfma_f32_32x32x16_bf16 (bf16 products are exact in the fp32
// accumulator) replace EIGHT v_mfma_f32_32x32x2_f32 per 16 contraction indices, at half the cycles each.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kPairWords = 1536;          // one 32x32 tile pair: [e = 2 K16 steps][level = 3][lane = 64][4 words = 8 bf16]

// the three bf16 levels of a 32-channel activation tile: p[e][level] is the MFMA operand of K16 step e
// (slot j of lane-half h <- register 8e + j, i.e. channel mlp_chan(8e + j, h))
struct ActSplit { u32x4 p[2][3]; };

__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32, nearest-even
}

__device__ __forceinline__ ActSplit split_act(const f32x16 &x)
{
    ActSplit s;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float a = x[8 * e + 2 * d], b = x[8 * e + 2 * d + 1];
            const unsigned int p1 = pack_bf16(a, b);
            const float ra = __fsub_rn(a, __uint_as_float(p1 << 16)), rb = __fsub_rn(b, __uint_as_float(p1 & 0xffff0000u));
            const unsigned int p2 = pack_bf16(ra, rb);
            const float sa = __fsub_rn(ra, __uint_as_float(p2 << 16)), sb = __fsub_rn(rb, __uint_as_float(p2 & 0xffff0000u));
            s.p[e][0][d] = p1;
            s.p[e][1][d] = p2;
            s.p[e][2][d] = pack_bf16(sa, sb);
        }
    return s;
}

// acc += sum over the 16 contraction slots of x * w, six bf16 MFMAs, small terms first. SWAP: the activations are
// the MFMA "A" operand (last layer: untransposed result), else the weights are.
template <bool SWAP>
__device__ __forceinline__ f32x16 mma_x6(const u32x4 (&w)[3], const u32x4 (&x)[3], f32x16 acc)
{
#define PN2_MMA(WL, XL)                                                                                              \
    acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x[XL]), __builtin_bit_cast(bf16x8, w[WL]), acc, 0, 0, 0) \
               : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[WL]), __builtin_bit_cast(bf16x8, x[XL]), acc, 0, 0, 0)
#ifdef PN2_LAB_ONE_MFMA                    /* lab builds only (scripts/build_mlp_labs.sh): what everything except the MFMAs costs */
    PN2_MMA(0, 0);
#else
    PN2_MMA(0, 2); PN2_MMA(1, 1); PN2_MMA(2, 0); PN2_MMA(0, 1); PN2_MMA(1, 0); PN2_MMA(0, 0);
#endif
#undef PN2_MMA
    return acc;
}

// host side: the three bf16 levels of one weight
void mlp_split_weight(float w, unsigned short out[3]);
// one 32x32 tile pair of a (kin, nout) row-major weight matrix in the split operand layout (kPairWords words)
float *mlp_pack_pair_x6(float *wp, const float *w, int kin, int nout, int t, int u, const int *krow);


// ---- streamed variant (sa_mlp_stream.hip) ------------------------------------------------------------
constexpr int kMlpStagePairs = 4;          // 32x32 weight tile pairs per LDS stage (24 KiB in three-level form)

constexpr int kS = kMlpStagePairs;

__host__ __device__ constexpr int pad_to_stage(int pairs) { return (pairs + kS - 1) / kS * kS; }

constexpr int kStreamThreads = 512;        // sa_mlp_stream.hip: eight waves share one weight stream
constexpr int kStageVec = kS * kPairWords / 4;          // 16-byte vectors per stage

// 12 MFMAs of one tile pair out of the current LDS stage (three-level bf16 operands, see split_act); SWAP:
// operands exchanged (last layer)
template <bool SWAP>
__device__ __forceinline__ f32x16 stream_pair(const u32x4 *stage, int slot, int lane, const ActSplit &act, f32x16 acc)
{
    const u32x4 *w4 = stage + slot * (kPairWords / 4) + lane;
    u32x4 w0[3] = {w4[0], w4[64], w4[128]};
    asm volatile("" ::: "memory");            // keeps hipcc from hoisting the reads of later pairs up here (register budget)
    const u32x4 w1[3] = {w4[192], w4[256], w4[320]};
    acc = mma_x6<SWAP>(w0, act.p[0], acc);
    acc = mma_x6<SWAP>(w1, act.p[1], acc);
    return acc;
}

struct MlpStreamConfig { int ti, t1, t2, t3; };         // ti: 32-channel tiles of the grouped FEATURES (per-point layer)
bool mlp_stream_pick(int cin, int c1, int c2, int c3, MlpStreamConfig &cfg);
size_t mlp_stream_w_floats(const MlpStreamConfig &c);
size_t mlp_stream_b_floats(const MlpStreamConfig &c);
size_t mlp_stream_ws_bytes(const MlpStreamConfig &c, long long points);
void mlp_stream_pack(const MlpStreamConfig &c, int cin, int c1, int c2, int c3, int xyz_first, const float *const *ws,
                     const float *const *bs, float *wpacked, float *bpacked);
int mlp_stream_launch(const MlpStreamConfig &c, int b, int n, int m, int nsample, int cfeat, int c3, const float *xyz,
                      const float *new_xyz, const float *points, const int *idx, const float *wp, const float *bp,
                      float *out, void *ws, hipStream_t st, int pooling = 0);

int point_layer_launch(int t1, int cfeat, long long rows, int tif, const float *points, const float *wstream,
                       const float *bias, float *pre, int out_stride, int col0, hipStream_t st);

bool point_layer_prefers_few_rows(long long rows, int tiles, int tif);
int point_layer_few_rows_launch(int tiles, int cfeat, long long rows, int tif, const float *points, const float *wstream,
                                const float *bias, float *pre, hipStream_t st);

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
                   const float *const *bs, float *wpacked, float *bpacked, bool krow_is_grouped);
bool mlp_coop_gemm_last(const MlpCoopConfig &c, int fp);
size_t mlp_coop_ws_bytes(const MlpCoopConfig &c, int fp, long long rows, int nsample);
int mlp_coop_launch(const MlpCoopConfig &c, int fp, const CoopParams &p, hipStream_t st, void *ws);

}  // namespace pn2
