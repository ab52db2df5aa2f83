// sa_mlp_stream.hip -- the fused grouped MLP + max-pool (see sa_mlp.hip for the formulation) for layer
// stacks whose weights do not fit in LDS: SA2/SA3-sized stacks such as 131 -> 128 -> 128 -> 256
// (396 KB of three-level weights), reference models/pointnet2_*: pointnet_sa_module(... mlp=[128,128,256] ...).
//
// The first layer is NOT evaluated per (centroid, sample) pair. Its input is [features of point j, xyz_j - c]
// (pointnet_util.py:44-50), so   W1^T x = W1f^T f_j  +  W1x^T (xyz_j - c) :   the feature part depends on the
// POINT only, and a point is a sample of nsample * m / n (16-64) groups. point_layer_kernel computes
// P[j] = W1f^T f_j + b1 once per point (b*n rows instead of b*m*nsample; the 32 consecutive points of a work
// item are contiguous rows: no gather) into a caller-provided scratch; the grouped kernel starts layer 1 from
// the gathered P[j] (128 floats per sample where the features would be up to 384) and adds the xyz part with ONE
// K16 step per output tile (those few weights stay in LDS for the whole launch, outside the stream). For cls_msg's second level (323 input channels) that removes 504 of 1104 MFMAs
// per 32 samples and the splitting of 11 input tiles. (A reassociation of the layer-1 sum, within fp32 rounding.
// The xyz part is deliberately NOT precomputed per point and per centroid: W1x^T xyz_j - W1x^T c would cancel
// catastrophically for clouds far from the origin, the reference subtracts coordinates first.)
//
// The weights are STREAMED: the packed array is the exact sequence of 32x32 tile pairs one work item
// (32 samples through the three layers) consumes, in the three-level bf16 operand layout of sa_mlp.hip
// (kPairWords words a pair), cut into stages of kMlpStagePairs pairs (24 KiB).
// The eight waves of a workgroup (one workgroup per CU: ONE stream per CU) walk their items in lockstep;
// while they run the MFMAs of stage s out of one LDS buffer, every thread holds its 48 bytes of stage
// s + 1 in registers (global loads issued a stage earlier, i.e. ~3000 cycles of MFMA work ago), writes
// them to the other buffer and one s_barrier flips the buffers. All workgroups stream the same bytes,
// so the source is the L2.
// Differences to the resident kernel, all forced by the register budget (256 VGPRs at 2 waves/SIMD):
//   * layer 2 walks the INPUT tiles in the outer loop (only one 32-channel tile of inputs is alive in
//     its three-level form, all output accumulators are);
//   * the last layer's 16 registers of a tile are max-reduced right away (they hold 16 samples of one
//     channel, see sa_mlp.hip), so the running maximum over a centroid's sample groups is T3 registers.
#include "sa_mlp_common.h"

#include <stdlib.h>
#include <string.h>

namespace pn2 {

constexpr int kXyzVec = 3 * 64;            // 16-byte vectors of one output tile's xyz weights: [level][lane]

// The double-buffered weight stream shared by the two kernels below (macros, not lambdas over an array: hipcc
// kept a captured array in scratch memory). Needs in scope: wstream, wbuf, tid, stages_per_item, stage, stg0-2.
#define PN2_STREAM_ISSUE(st)                                                                                           \
    do {                                                                                                               \
        const u32x4 *src_ = reinterpret_cast<const u32x4 *>(wstream) + (size_t)((st) % stages_per_item) * kStageVec + tid; \
        stg0 = src_[0]; stg1 = src_[kStreamThreads]; stg2 = src_[2 * kStreamThreads];                                  \
    } while (0)
#define PN2_STREAM_COMMIT(st)                                                                                          \
    do {                                                                                                               \
        u32x4 *dst_ = wbuf[(st) & 1] + tid;                                                                            \
        dst_[0] = stg0; dst_[kStreamThreads] = stg1; dst_[2 * kStreamThreads] = stg2;                                  \
    } while (0)
// when the MFMAs of `stage` are issued: publish stage + 1, fetch stage + 2
#define PN2_NEXT_STAGE()                                                                                               \
    do {                                                                                                               \
        PN2_STREAM_COMMIT(stage + 1);                                                                                  \
        __syncthreads();                                                                                               \
        ++stage;                                                                                                       \
        PN2_STREAM_ISSUE(stage + 1);                                                                                   \
    } while (0)
static_assert(kStageVec == 3 * kStreamThreads, "the staging registers are spelled out for three vectors per thread");

// P (rows, 32 * T1) = points (rows, cfeat) . W1f + b1: the feature part of layer 1, once per point. A wave owns
// 32 consecutive rows; non-swapped operands, so lane (row s, half h) holds channels 8q + 4h .. + 3 of a tile in
// registers 4q .. 4q + 3: one 16-byte store per quartet, the layout the grouped kernel loads back.
// out_stride / col0: row pitch of `pre` in floats and the first column this launch writes (a 256-wide layer is two
// launches of the T1 = 4 instance, fp_mlp.hip).
template <int T1>
__global__ __launch_bounds__(kStreamThreads) void point_layer_kernel(int cfeat, long long rows, int tif,
                                                                    const float *__restrict__ points,
                                                                    const float *__restrict__ wstream,
                                                                    const float *__restrict__ bpacked, float *__restrict__ pre,
                                                                    int out_stride, int col0)
{
    __shared__ __attribute__((aligned(16))) u32x4 wbuf[2][kStageVec];
    __shared__ float bias_s[T1 * 32];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s = lane & 31;
    for (int i = tid; i < T1 * 32; i += kStreamThreads) bias_s[i] = bpacked[i];
    const int stages_per_item = pad_to_stage(tif * T1) / kS;
    int stage = 0;
    u32x4 stg0, stg1, stg2;
    if (stages_per_item > 0) {
        PN2_STREAM_ISSUE(0);
        PN2_STREAM_COMMIT(0);
    }
    __syncthreads();
    if (stages_per_item > 0) PN2_STREAM_ISSUE(1);

    const long long groups = (rows + 31) / 32;
    const long long wave = (long long)blockIdx.x * (kStreamThreads / 64) + (tid >> 6);
    const long long nwaves = (long long)gridDim.x * (kStreamThreads / 64);
    const long long trips = (groups + nwaves - 1) / nwaves;                // lockstep: every wave runs all trips
    const bool vec4 = (cfeat & 3) == 0;
    for (long long trip = 0; trip < trips; ++trip) {
        const long long g = wave + trip * nwaves;
        const long long row = min((g < groups ? g : groups - 1) * 32 + s, rows - 1);
        const bool ok = g < groups && g * 32 + s < rows;
        const float *pf = points + (size_t)row * cfeat;
        auto load_tile = [&](int u) __attribute__((always_inline)) -> f32x16 {
            f32x16 x;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k0 = 32 * u + 8 * q + 4 * h;                     // channels k0 .. k0+3 -> registers 4q .. 4q+3
                if (vec4 && k0 + 3 < cfeat) {
                    const float4 f = *reinterpret_cast<const float4 *>(pf + k0);
                    x[4 * q] = f.x; x[4 * q + 1] = f.y; x[4 * q + 2] = f.z; x[4 * q + 3] = f.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[4 * q + r] = k0 + r < cfeat ? pf[k0 + r] : 0.0f;
                }
            }
            return x;
        };
        f32x16 acc[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t) acc[t] = mlp_bias(bias_s, t, h);
        if (tif > 0) {
            // the NEXT tile is split into its levels in the middle of this tile's MFMAs (the barrier of the stream keeps
            // the eight waves in step: a split at the top of the loop would idle every matrix pipe at the same time)
            ActSplit xs = split_act(load_tile(0));
            f32x16 xn = load_tile(tif > 1 ? 1 : 0);
            int slot = 0;
            for (int u = 0; u < tif; ++u) {
                ActSplit xsn;
#pragma unroll
                for (int t = 0; t < T1; ++t) {
                    if (t == T1 / 2) {
                        xsn = split_act(xn);
                        if (u + 2 < tif) xn = load_tile(u + 2);            // two tiles ahead of the MFMAs
                    }
                    acc[t] = stream_pair<false>(wbuf[stage & 1], slot, lane, xs, acc[t]);
                    if (++slot == kS) { slot = 0; PN2_NEXT_STAGE(); }
                }
                xs = xsn;
            }
            if (slot != 0) PN2_NEXT_STAGE();                                    // padded to whole stages
        }
        if (ok) {
            float *dst = pre + (size_t)row * out_stride + col0 + 4 * h;
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(dst + 32 * t + 8 * q) =
                        make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        }
    }
}

// The same layer for FEW rows (feature-propagation levels with 16 ... 2048 known points, fp_mlp.hip): the streamed
// form above walks the contraction in stages of four pairs behind a barrier, one microsecond a stage whatever the
// row count -- 32 stages for the 1024 input channels of part_seg's first FP level, for a single work item. Here a wave
// group owns ONE output tile of one 32-row item; its four waves take every fourth feature tile each (a split of the
// contraction: the chain is tif / 4 pairs of 12 MFMAs), stream just those pairs from L2 into registers, two ahead,
// and add their partial tiles through LDS at the end. items x tiles workgroups run side by side.
// wstream: the per-point streams of fp_mlp.hip, [tile / 4][feature tile][tile % 4]; writes all `tiles` tiles. bias: the
// packed bias of these tiles ([tile][half][register]) or nullptr.
// VEC4: cfeat % 4 == 0. The tile loads are branch-free (clamped address + select): per-lane branches around loads make
// hipcc wait for every load in flight at each use, the prefetched ones included.
template <bool VEC4>
__global__ __launch_bounds__(256) void point_layer_few_rows_kernel(int cfeat, long long rows, int tif, int tiles,
                                                                  const float *__restrict__ points,
                                                                  const float *__restrict__ wstream,
                                                                  const float *__restrict__ bias, float *__restrict__ pre)
{
    constexpr int kTileVec = kPairWords / 4;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s = lane & 31;
    __shared__ float part[3][16][64];                                      // partial tiles of waves 1-3
    const int w = tid >> 6;
    const long long g = blockIdx.x / tiles;
    const int t = (int)(blockIdx.x % tiles);
    const long long row = min(g * 32 + s, rows - 1);
    const float *pf = points + (size_t)row * cfeat;
    auto load_tile = [&](int u) __attribute__((always_inline)) -> f32x16 {
        f32x16 x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = 32 * u + 8 * q + 4 * h;
            if (VEC4) {
                const bool in = k0 < cfeat;                            // cfeat % 4 == 0: a quartet is inside or outside
                const float4 f = *reinterpret_cast<const float4 *>(pf + (in ? k0 : 0));
                x[4 * q] = in ? f.x : 0.0f; x[4 * q + 1] = in ? f.y : 0.0f; x[4 * q + 2] = in ? f.z : 0.0f; x[4 * q + 3] = in ? f.w : 0.0f;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool in = k0 + r < cfeat;
                    const float f = pf[in ? k0 + r : 0];
                    x[4 * q + r] = in ? f : 0.0f;
                }
            }
        }
        return x;
    };
    // pair of (feature tile u, this output tile)
    const u32x4 *wp4 = reinterpret_cast<const u32x4 *>(wstream) + ((size_t)(t / 4) * tif * 4 + (t % 4)) * kTileVec + lane;
    const int mine = (tif - w + 3) / 4;                                    // this wave's feature tiles: w, w + 4, ...
    const int last = max(mine - 1, 0);
    const bool none = mine <= 0;
    u32x4 a0, a1, a2, a3, a4, a5, b0, b1, b2, b3, b4, b5;                  // pairs u (set a / b by parity), two in flight
#define PN2_FEW_ISSUE(u, S)                                                                                           \
    do {                                                                                                              \
        const u32x4 *q_ = wp4 + (size_t)min(w + 4 * (u), tif - 1) * (4 * kTileVec);                                   \
        S##0 = q_[0]; S##1 = q_[64]; S##2 = q_[128]; S##3 = q_[192]; S##4 = q_[256]; S##5 = q_[320];                   \
    } while (0)
#define PN2_FEW_STEP(S, X)                                                                                            \
    do {                                                                                                              \
        const u32x4 w0_[3] = {S##0, S##1, S##2}, w1_[3] = {S##3, S##4, S##5};                                         \
        const ActSplit xs_ = split_act(X);                                                                            \
        acc = mma_x6<false>(w0_, xs_.p[0], acc);                                                                      \
        acc = mma_x6<false>(w1_, xs_.p[1], acc);                                                                      \
    } while (0)
    f32x16 acc;                                                            // wave 0 starts from the bias (packed tile layout) if any
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = (bias && w == 0) ? bias[(t * 2 + h) * 16 + v] : 0.0f;
    auto my_tile = [&](int j) __attribute__((always_inline)) -> f32x16 { return load_tile(min(w + 4 * min(j, last), tif - 1)); };
    PN2_FEW_ISSUE(0, a);
    PN2_FEW_ISSUE(min(1, last), b);
    f32x16 x0 = my_tile(0), x1 = my_tile(1);
    // straight-line double step (behind a branch hipcc makes every use wait for ALL loads in flight, the prefetch
    // included): a surplus step (odd count, or a wave without tiles) runs on a zero tile
    for (int u = 0; u < max(mine, 1); u += 2) {
        if (none) {
#pragma unroll
            for (int v = 0; v < 16; ++v) x0[v] = 0.0f;
        }
        PN2_FEW_STEP(a, x0);
        PN2_FEW_ISSUE(min(u + 2, last), a);
        x0 = my_tile(u + 2);
        if (u + 1 >= mine) {
#pragma unroll
            for (int v = 0; v < 16; ++v) x1[v] = 0.0f;
        }
        PN2_FEW_STEP(b, x1);
        PN2_FEW_ISSUE(min(u + 3, last), b);
        x1 = my_tile(u + 3);
    }
#undef PN2_FEW_ISSUE
#undef PN2_FEW_STEP
    if (w > 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) part[w - 1][v][lane] = acc[v];
    }
    __syncthreads();
    if (w > 0) return;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = __fadd_rn(__fadd_rn(acc[v], part[0][v][lane]), __fadd_rn(part[1][v][lane], part[2][v][lane]));
    if (g * 32 + s < rows) {
        float *dst = pre + (size_t)row * (32 * tiles) + 32 * t + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(dst + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
}

// POOL: 0 max, 1 avg, 2 weighted_avg, 3 max_and_avg (utils/pointnet_util.py:128-140; see sa_mlp3_kernel in sa_mlp.hip: modes 1-3
// apply bias + ReLU per sample and keep a running (weighted) sum per lane and output tile beside / instead of the raw maximum)
template <int T1, int T2, int T3, int POOL = 0>
__global__ __launch_bounds__(kStreamThreads) void sa_mlp3_stream_kernel(int n, int m, int nsample, int c3, long long rows,
                                                                       const float *__restrict__ xyz,
                                                                       const float *__restrict__ new_xyz,
                                                                       const float *__restrict__ pre,
                                                                       const int *__restrict__ idx,
                                                                       const float *__restrict__ wstream,
                                                                       const float *__restrict__ wxyz,
                                                                       const float *__restrict__ bpacked,
                                                                       float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) u32x4 wbuf[2][kStageVec];
    __shared__ __attribute__((aligned(16))) u32x4 wx_s[T1 * kXyzVec];     // layer 1's xyz rows: [t][level][lane]
    __shared__ float bias_s[(T2 + T3) * 32];
    const float *b2 = bias_s, *b3 = b2 + T2 * 32;                          // b1 went into P
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s = lane & 31;
    for (int i = tid; i < (T2 + T3) * 32; i += kStreamThreads) bias_s[i] = bpacked[T1 * 32 + i];
    for (int i = tid; i < T1 * kXyzVec; i += kStreamThreads) wx_s[i] = reinterpret_cast<const u32x4 *>(wxyz)[i];

    constexpr int stages_per_item = (T2 * T1 + T3 * T2) / kS;              // both products are multiples of kS
    int stage = 0;                                   // running stage number (all items), uniform over the workgroup
    u32x4 stg0, stg1, stg2;                          // this thread's 48 bytes of stage `stage + 1`
    PN2_STREAM_ISSUE(0);
    PN2_STREAM_COMMIT(0);
    __syncthreads();
    PN2_STREAM_ISSUE(1);

    const int parts = nsample / 32;
    const long long wave = (long long)blockIdx.x * (kStreamThreads / 64) + (tid >> 6);
    const long long nwaves = (long long)gridDim.x * (kStreamThreads / 64);
    const long long trips = (rows + nwaves - 1) / nwaves;                  // lockstep: every wave runs all trips

    for (long long trip = 0; trip < trips; ++trip) {
        const long long row_raw = wave + trip * nwaves;
        const bool row_ok = row_raw < rows;
        const long long row = row_ok ? row_raw : rows - 1;
        const long long cloud = row / m;
        const float *c = new_xyz + row * 3;
        const float cx = c[0], cy = c[1], cz = c[2];
        float best[T3];
        float sum[T3], esum = 0.0f;                  // POOL != 0: running sums of relu(raw + bias) (x weight), the weights' sum
        for (int part = 0; part < parts; ++part) {
            const int p = idx[row * nsample + part * 32 + s];
            float wv[16];
            // layer 1: the point's precomputed feature part + the xyz part, one K16 step per output tile
            // (relative coordinates are channels 0-2 = registers 0-2 of lanes 0-31)
            f32x16 h1[T1];
            {
                const float *pp = pre + ((size_t)cloud * n + p) * (32 * T1) + 4 * h;
                const float *px = xyz + ((size_t)cloud * n + p) * 3;
                f32x16 x;
#pragma unroll
                for (int v = 0; v < 16; ++v) x[v] = 0.0f;
                if (h == 0) { x[0] = __fsub_rn(px[0], cx); x[1] = __fsub_rn(px[1], cy); x[2] = __fsub_rn(px[2], cz); }
#pragma unroll
                for (int t = 0; t < T1; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 f = *reinterpret_cast<const float4 *>(pp + 32 * t + 8 * q);
                        h1[t][4 * q] = f.x; h1[t][4 * q + 1] = f.y; h1[t][4 * q + 2] = f.z; h1[t][4 * q + 3] = f.w;
                    }
                if (POOL == 2) {                 // :133-134: exp(-5 |grouped_xyz|) of this lane's sample (lanes 0-31), dealt to the registers' samples
                    const float e = expf(-5.0f * sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x[0], x[0]), __fmul_rn(x[1], x[1])), __fmul_rn(x[2], x[2]))));
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        wv[v] = __shfl(e, 8 * (v >> 2) + 4 * h + (v & 3));
                        esum += wv[v];
                    }
                }
                const ActSplit xs = split_act(x);
#pragma unroll
                for (int t = 0; t < T1; ++t) {
                    const u32x4 *wx = wx_s + t * kXyzVec + lane;
                    const u32x4 w0[3] = {wx[0], wx[64], wx[128]};
                    h1[t] = mma_x6<false>(w0, xs.p[0], h1[t]);
                }
            }
            // layer 2, input tiles outermost: only ONE input tile is alive in its three-level form beside the T2
            // accumulators. The level split of the NEXT tile sits in the middle of this tile's pairs: the stream's
            // barrier keeps the eight waves in step, so a split between two tiles would idle every matrix pipe at once.
            f32x16 a2[T2];
#pragma unroll
            for (int t = 0; t < T2; ++t) a2[t] = mlp_bias(b2, t, h);
            {
                ActSplit su = split_act(mlp_relu(h1[0]));
#pragma unroll
                for (int u = 0; u < T1; ++u) {
                    ActSplit sn;
#pragma unroll
                    for (int t = 0; t < T2; ++t) {
                        if (t == T2 / 2 && u + 1 < T1) sn = split_act(mlp_relu(h1[u + 1]));
                        a2[t] = stream_pair<false>(wbuf[stage & 1], (u * T2 + t) % kS, lane, su, a2[t]);
                        if ((u * T2 + t) % kS == kS - 1) PN2_NEXT_STAGE();
                    }
                    if (u + 1 < T1) su = sn;
                }
            }
            // layer 3, operands swapped: a lane holds 16 samples of channel 32t + (l & 31). Its first output tile
            // walks the input tiles while the following one is still being split (same reason as above).
            ActSplit s2[T2];
            s2[0] = split_act(mlp_relu(a2[0]));
#pragma unroll
            for (int t = 0; t < T3; ++t) {
                f32x16 acc;
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
                for (int u = 0; u < T2; ++u) {
                    if (t == 0 && u + 1 < T2) s2[u + 1] = split_act(mlp_relu(a2[u + 1]));
                    acc = stream_pair<true>(wbuf[stage & 1], (t * T2 + u) % kS, lane, s2[u], acc);
                    if ((t * T2 + u) % kS == kS - 1) PN2_NEXT_STAGE();
                }
                if (POOL != 1 && POOL != 2) {
                    float mx = acc[0];
#pragma unroll
                    for (int v = 1; v < 16; ++v) mx = fmaxf(mx, acc[v]);
                    best[t] = part == 0 ? mx : fmaxf(best[t], mx);
                }
                if (POOL != 0) {
                    const float bias = b3_at(b3, 32 * t + s);
                    float sm = 0.0f;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        float r = fmaxf(__fadd_rn(acc[v], bias), 0.0f);
                        if (POOL == 2) r *= wv[v];
                        sm += r;
                    }
                    sum[t] = part == 0 ? sm : sum[t] + sm;
                }
            }
        }
        if (POOL != 0) {
            const int oc = POOL == 3 ? 2 * c3 : c3;
            float den = (float)nsample;
            if (POOL == 2) den = esum + __shfl_xor(esum, 32);
#pragma unroll
            for (int t = 0; t < T3; ++t) {
                const int ch = 32 * t + s;
                const float sm = sum[t] + __shfl_xor(sum[t], 32);
                float mx = 0.0f;
                if (POOL == 3) mx = fmaxf(best[t], __shfl_xor(best[t], 32));
                if (h == 0 && ch < c3 && row_ok) {
                    out[row * oc + ch] = sm / den;
                    if (POOL == 3) out[row * oc + c3 + ch] = fmaxf(__fadd_rn(mx, b3_at(b3, ch)), 0.0f);
                }
            }
            continue;
        }
#pragma unroll
        for (int t = 0; t < T3; ++t) {
            const int ch = 32 * t + s;
            const float mx = fmaxf(best[t], __shfl_xor(best[t], 32));
            if (h == 0 && ch < c3 && row_ok) out[row * c3 + ch] = fmaxf(__fadd_rn(mx, b3_at(b3, ch)), 0.0f);
        }
    }
}
#undef PN2_STREAM_ISSUE
#undef PN2_STREAM_COMMIT
#undef PN2_NEXT_STAGE

bool mlp_stream_pick(int cin, int c1, int c2, int c3, MlpStreamConfig &cfg)
{
    static const int kShapes[][3] = {{2, 2, 4}, {4, 4, 8}};
    if (cin < 3 || cin > 32 * 12) return false;
    for (const auto &sh : kShapes)
        if (c1 <= 32 * sh[0] && c2 <= 32 * sh[1] && c3 <= 32 * sh[2]) {
            cfg = {(cin - 3 + 31) / 32, sh[0], sh[1], sh[2]};          // ti: tiles of FEATURE channels (the per-point layer)
            return true;
        }
    return false;
}

// packed weights: [grouped kernel's stream: layer 2, layer 3][per-point kernel's stream: feature pairs of layer 1]
// [xyz rows of layer 1: K16 step 0 of one pair per output tile, resident in LDS]
static int stream_main_pairs(const MlpStreamConfig &c) { return c.t2 * c.t1 + c.t3 * c.t2; }
static int stream_point_pairs(const MlpStreamConfig &c) { return pad_to_stage(c.ti * c.t1); }
size_t mlp_stream_w_floats(const MlpStreamConfig &c)
{
    return (size_t)(stream_main_pairs(c) + stream_point_pairs(c)) * kPairWords + (size_t)c.t1 * kXyzVec * 4;
}
size_t mlp_stream_b_floats(const MlpStreamConfig &c) { return (size_t)(c.t1 + c.t2 + c.t3) * 32; }
size_t mlp_stream_ws_bytes(const MlpStreamConfig &c, long long points) { return sizeof(float) * (size_t)points * 32 * c.t1; }

void mlp_stream_pack(const MlpStreamConfig &c, int cin, int c1, int c2, int c3, int xyz_first, const float *const *ws,
                     const float *const *bs, float *wpacked, float *bpacked)
{
    // caller's rows of w1: [xyz, features] when xyz_first, else [features, xyz]
    const int cfeat = cin - 3;
    int xrow[3], *frow = (int *)malloc(sizeof(int) * (size_t)(cfeat > 0 ? cfeat : 1));
    for (int k = 0; k < 3; ++k) xrow[k] = xyz_first ? k : cfeat + k;
    for (int k = 0; k < cfeat; ++k) frow[k] = xyz_first ? 3 + k : k;
    float *wp = wpacked;
    for (int u = 0; u < c.t1; ++u)                        // layer 2 walks its input tiles outermost
        for (int t = 0; t < c.t2; ++t) wp = mlp_pack_pair_x6(wp, ws[1], c1, c2, t, u, nullptr);
    for (int t = 0; t < c.t3; ++t)
        for (int u = 0; u < c.t2; ++u) wp = mlp_pack_pair_x6(wp, ws[2], c2, c3, t, u, nullptr);
    for (int u = 0; u < c.ti; ++u)                        // the per-point layer: feature tiles outermost
        for (int t = 0; t < c.t1; ++t) wp = mlp_pack_pair_x6(wp, ws[0], cfeat, c1, t, u, frow);
    for (int i = c.ti * c.t1; i < pad_to_stage(c.ti * c.t1); ++i)
        for (int j = 0; j < kPairWords; ++j) *wp++ = 0.0f;
    float *tmp = (float *)malloc(sizeof(float) * kPairWords);
    for (int t = 0; t < c.t1; ++t) {                      // xyz rows: channels 0-2 live in K16 step 0 of input tile 0
        mlp_pack_pair_x6(tmp, ws[0], 3, c1, t, 0, xrow);
        memcpy(wp, tmp, sizeof(float) * kXyzVec * 4);
        wp += kXyzVec * 4;
    }
    free(tmp);
    free(frow);
    const int nout[3] = {c1, c2, c3}, tout[3] = {c.t1, c.t2, c.t3};
    float *bp = bpacked;
    for (int L = 0; L < 3; ++L)
        for (int t = 0; t < tout[L]; ++t)
            for (int hh = 0; hh < 2; ++hh)
                for (int v = 0; v < 16; ++v) {
                    const int ch = 32 * t + mlp_chan(v, hh);
                    *bp++ = ch < nout[L] ? bs[L][ch] : 0.0f;
                }
}

// pre (rows, out_stride)[:, col0 : col0 + 32 t1] = points (rows, cfeat) . W + bias; t1 = 2 or 4 output tiles
int point_layer_launch(int t1, int cfeat, long long rows, int tif, const float *points, const float *wstream,
                       const float *bias, float *pre, int out_stride, int col0, hipStream_t st)
{
    const long long groups = (rows + 31) / 32;
    long long blocks = (groups + kStreamThreads / 64 - 1) / (kStreamThreads / 64);
    if (blocks > 256) blocks = 256;
    if (blocks == 0) return PN2_OK;
    if (t1 == 2)
        return launch((point_layer_kernel<2>), dim3((unsigned)blocks), dim3(kStreamThreads), 0, st, cfeat, rows, tif, points, wstream,
                      bias, pre, out_stride, col0);
    if (t1 == 4)
        return launch((point_layer_kernel<4>), dim3((unsigned)blocks), dim3(kStreamThreads), 0, st, cfeat, rows, tif, points, wstream,
                      bias, pre, out_stride, col0);
    return PN2_E_ARG;
}

// The few-rows form re-reads a tile's weights for every 32-row item (from L2): it wins while that traffic stays small
// -- items x tiles x feature tiles pairs of 6 KiB, bounded here at 96 MiB per launch (cls_msg level 2, 120 MiB: 30.6 us against 28.9 streamed).
bool point_layer_prefers_few_rows(long long rows, int tiles, int tif) { return (rows + 31) / 32 * tiles * tif <= 16384; }

// the few-rows form: all `tiles` output tiles (a multiple of 4) in one launch
int point_layer_few_rows_launch(int tiles, int cfeat, long long rows, int tif, const float *points, const float *wstream,
                                const float *bias, float *pre, hipStream_t st)
{
    const long long blocks = (rows + 31) / 32 * tiles;
    if (blocks == 0) return PN2_OK;
    if ((cfeat & 3) == 0)
        return launch(point_layer_few_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, cfeat, rows, tif, tiles, points,
                      wstream, bias, pre);
    return launch(point_layer_few_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, cfeat, rows, tif, tiles, points,
                  wstream, bias, pre);
}

template <int T1, int T2, int T3>
static int launch_stream(const MlpStreamConfig &c, int b, int n, int m, int nsample, int cfeat, int c3, const float *xyz,
                         const float *new_xyz, const float *points, const int *idx, const float *wp, const float *bp,
                         float *out, float *pre, hipStream_t st, int pooling)
{
    const long long cap = 256;                           // one workgroup (one weight stream) per CU
    const long long npoints = (long long)b * n;
    long long blocks;
    const float *wpoint = wp + (size_t)stream_main_pairs(c) * kPairWords;
    const float *wxyz = wpoint + (size_t)stream_point_pairs(c) * kPairWords;
    // few points and a 128-wide first layer (its stream [feature tile][4 tiles] is the few-rows kernel's layout too):
    // the form without stages, see point_layer_few_rows_kernel
    if (T1 == 4 && c.ti > 0 && point_layer_prefers_few_rows(npoints, 4, c.ti)) {
        if (int rc = point_layer_few_rows_launch(4, cfeat, npoints, c.ti, points, wpoint, bp, pre, st)) return rc;
    } else if (int rc = point_layer_launch(T1, cfeat, npoints, c.ti, points, wpoint, bp, pre, 32 * T1, 0, st)) {
        return rc;
    }
    const long long rows = (long long)b * m;
    blocks = (rows + kStreamThreads / 64 - 1) / (kStreamThreads / 64);
    if (blocks > cap) blocks = cap;
#define PN2_STREAM_POOL(P)                                                                                                   \
    if (pooling == P)                                                                                                        \
        return launch((sa_mlp3_stream_kernel<T1, T2, T3, P>), dim3((unsigned)blocks), dim3(kStreamThreads), 0, st, n, m, nsample, c3, \
                      rows, xyz, new_xyz, (const float *)pre, idx, wp, wxyz, bp, out);
    PN2_STREAM_POOL(0) PN2_STREAM_POOL(1) PN2_STREAM_POOL(2) PN2_STREAM_POOL(3)
#undef PN2_STREAM_POOL
    return PN2_E_ARG;
}

int mlp_stream_launch(const MlpStreamConfig &c, int b, int n, int m, int nsample, int cfeat, int c3, const float *xyz,
                      const float *new_xyz, const float *points, const int *idx, const float *wp, const float *bp,
                      float *out, void *ws, hipStream_t st, int pooling)
{
    if (nsample <= 0 || nsample % 32 != 0) return PN2_E_ARG;
    if (!ws) return PN2_E_NULL;
    if (c.t1 == 2 && c.t2 == 2 && c.t3 == 4)
        return launch_stream<2, 2, 4>(c, b, n, m, nsample, cfeat, c3, xyz, new_xyz, points, idx, wp, bp, out, (float *)ws, st, pooling);
    if (c.t1 == 4 && c.t2 == 4 && c.t3 == 8)
        return launch_stream<4, 4, 8>(c, b, n, m, nsample, cfeat, c3, xyz, new_xyz, points, idx, wp, bp, out, (float *)ws, st, pooling);
    return PN2_E_TOO_LARGE;
}

}  // namespace pn2
