// fp_mlp.hip -- a whole feature-propagation layer behind three_nn in ONE kernel, on the matrix cores.
//
// Reference: pointnet_fp_module, utils/pointnet_util.py:199-229 --
//     dist, idx = three_nn(xyz1, xyz2)                                   (:211, stays pn2_three_nn)
//     dist = max(dist, 1e-10); norm = sum(1/dist); weight = (1/dist)/norm (:212-215)
//     interpolated = three_interpolate(points2, idx, weight)             (:216)
//     new_points1 = concat([interpolated, points1], axis=2)              (:219, interpolated FIRST)
//     for each width: conv2d 1x1 + batch_norm + ReLU                     (:223-226)
// i.e. five to ten launches with every intermediate -- weights, the (b,n,c2) interpolated tensor, the
// concatenation, every layer's activations -- in HBM. Here a wave owns 32 unknown points: each lane
// computes its point's three inverse-distance weights from `dist`, gathers the three known rows a
// 32-channel tile at a time straight into the layer-1 MFMA operand registers (weighted sum in the
// reference's order (p1*w1 + p2*w2) + p3*w3, skip-link channels appended), and runs the layer stack with
// the machinery of the streamed-weights SA kernel (sa_mlp_stream.hip: activations chained in accumulator
// registers between layers in their three-level bf16 operand form, weight tile pairs streamed through a
// double-buffered LDS stage shared by the four waves,
// the LAST layer with swapped MFMA operands so that a lane holds 16 points of one output channel). The
// epilogue is a plain store of bias + ReLU instead of a max-pool: 128 contiguous bytes per half wave.
// Batch norm is folded into the weights by the caller (inference), like pn2_sa_mlp3_maxpool.
//
// The first layer is linear in its input and the interpolation is linear in the known features:
//     W1^T [interp(points2), points1] = interp(points2 . W1a) + W1b^T points1.
// point_layer_kernel (sa_mlp_stream.hip) therefore evaluates Q = points2 . W1a ONCE PER KNOWN POINT (m rows, a
// quarter or less of the n unknown ones at every level of the reference models) into caller scratch; the kernels
// here interpolate rows of Q (the first width's channels instead of c2) into the layer-1 accumulators and only the
// skip-link part of layer 1 is left as MFMA work per unknown point (none at all when there is no skip link: sem_seg's
// last level). A reassociation of the sum inside fp32 rounding; the weights w1 + w2 + w3 = 1 multiply Q, not the bias.
//
// Stacks: two or three layers, widths up to 256 with (tiles of layer 1) + (tiles of layer 2) <= 16 --
// every FP stack of the reference models: [256,256], [256,128], [128,128,128]
// (pointnet2_part_seg.py:31-33, pointnet2_sem_seg.py:34-37) -- and any number of input channels.
#include "sa_mlp_common.h"

#include <limits.h>
#include <stdlib.h>

namespace pn2 {

__device__ __forceinline__ float fp_interp3(float p1, float p2, float p3, float w1, float w2, float w3)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(p1, w1), __fmul_rn(p2, w2)), __fmul_rn(p3, w3));   // tf_interpolate.cpp:122
}

// T3 == 0: two layers (layer 2 is the last one). TL = tiles of the last layer.
template <int T1, int T2, int T3>
__global__ __launch_bounds__(kMlpThreads) void fp_mlp_stream_kernel(int n, int m, int c1, int cout, long long rows,
                                                                    int ti, const float *__restrict__ pre,
                                                                    const float *__restrict__ points1,
                                                                    const int *__restrict__ idx,
                                                                    const float *__restrict__ dist,
                                                                    const float *__restrict__ wstream,
                                                                    const float *__restrict__ bpacked,
                                                                    float *__restrict__ out)
{
    constexpr int TB = T1 + T2 + T3;
    __shared__ __attribute__((aligned(16))) u32x4 wbuf[2][kStageVec];
    __shared__ float bias_s[TB * 32];
    const float *b1 = bias_s, *b2 = b1 + T1 * 32, *b3 = b2 + T2 * 32;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s = lane & 31;
    for (int i = tid; i < TB * 32; i += kMlpThreads) bias_s[i] = bpacked[i];

    const int l1_pairs = pad_to_stage(ti * T1);
    const int stages_per_item = (l1_pairs + T2 * T1 + T3 * T2) / kS;
    int stage = 0;
    u32x4 stg0, stg1, stg2, stg3, stg4, stg5;
    static_assert(kStageVec == 6 * kMlpThreads, "the staging registers are spelled out for six vectors per thread");
    static_assert((T2 * T1) % kS == 0 && (T3 * T2) % kS == 0, "layers must start on stage boundaries");
#define PN2_STREAM_ISSUE(st)                                                                                           \
    do {                                                                                                               \
        const u32x4 *src_ = reinterpret_cast<const u32x4 *>(wstream) + (size_t)((st) % stages_per_item) * kStageVec + tid; \
        stg0 = src_[0]; stg1 = src_[256]; stg2 = src_[512]; stg3 = src_[768]; stg4 = src_[1024]; stg5 = src_[1280];   \
    } while (0)
#define PN2_STREAM_COMMIT(st)                                                                                          \
    do {                                                                                                               \
        u32x4 *dst_ = wbuf[(st) & 1] + tid;                                                                            \
        dst_[0] = stg0; dst_[256] = stg1; dst_[512] = stg2; dst_[768] = stg3; dst_[1024] = stg4; dst_[1280] = stg5;   \
    } while (0)
#define PN2_NEXT_STAGE()                                                                                               \
    do {                                                                                                               \
        PN2_STREAM_COMMIT(stage + 1);                                                                                  \
        __syncthreads();                                                                                               \
        ++stage;                                                                                                       \
        PN2_STREAM_ISSUE(stage + 1);                                                                                   \
    } while (0)
    PN2_STREAM_ISSUE(0);
    PN2_STREAM_COMMIT(0);
    __syncthreads();
    PN2_STREAM_ISSUE(1);

    const long long groups = (rows + 31) / 32;                              // work item = 32 consecutive unknown points
    const long long wave = (long long)blockIdx.x * (kMlpThreads / 64) + (tid >> 6);
    const long long nwaves = (long long)gridDim.x * (kMlpThreads / 64);
    const long long trips = (groups + nwaves - 1) / nwaves;                 // lockstep: every wave runs all trips
    const bool vec1 = (c1 & 3) == 0;

    for (long long trip = 0; trip < trips; ++trip) {
        const long long g = wave + trip * nwaves;
        const long long row0 = (g < groups ? g : groups - 1) * 32;
        const bool item_ok = g < groups;
        const long long row = min(row0 + s, rows - 1);                       // this lane's point (clamped: loads stay valid)
        const long long cloud = row / n;
        // inverse-distance weights, pointnet_util.py:212-215
        const int *ip = idx + row * 3;
        const float *dp = dist + row * 3;
        const int i1 = ip[0], i2 = ip[1], i3 = ip[2];
        const float r1 = __fdiv_rn(1.0f, fmaxf(dp[0], 1e-10f)), r2 = __fdiv_rn(1.0f, fmaxf(dp[1], 1e-10f)),
                    r3 = __fdiv_rn(1.0f, fmaxf(dp[2], 1e-10f));
        const float norm = __fadd_rn(__fadd_rn(r1, r2), r3);
        const float w1 = __fdiv_rn(r1, norm), w2 = __fdiv_rn(r2, norm), w3 = __fdiv_rn(r3, norm);
        // rows of Q = points2 . W1a of the three neighbours (32 * T1 floats each), this lane's channels 4h .. 4h + 3 of a quartet
        const float *qbase = pre + (size_t)cloud * m * (32 * T1) + 4 * h;
        const float *qa = qbase + (size_t)i1 * (32 * T1), *qb = qbase + (size_t)i2 * (32 * T1), *qc = qbase + (size_t)i3 * (32 * T1);
        const float *p1 = points1 ? points1 + (size_t)row * c1 : nullptr;

        // one 32-channel tile of the skip features: register v <- channel 32u + mlp_chan(v, h) of points1
        auto gather = [&](int u) __attribute__((always_inline)) -> f32x16 {
            f32x16 x;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k0 = 32 * u + 8 * q + 4 * h;
                if (vec1 && k0 + 3 < c1) {
                    const float4 f = *reinterpret_cast<const float4 *>(p1 + k0);
                    x[4 * q] = f.x; x[4 * q + 1] = f.y; x[4 * q + 2] = f.z; x[4 * q + 3] = f.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[4 * q + r] = k0 + r < c1 ? p1[k0 + r] : 0.0f;
                }
            }
            return x;
        };

        // layer 1: bias + the interpolated rows of Q (tf_interpolate.cpp:122's order per channel) + the skip part on
        // the matrix pipe, input tiles outermost (one gathered tile alive, all T1 accumulators alive)
        f32x16 h1[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            h1[t] = mlp_bias(b1, t, h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = *reinterpret_cast<const float4 *>(qa + 32 * t + 8 * q), bq = *reinterpret_cast<const float4 *>(qb + 32 * t + 8 * q),
                             c = *reinterpret_cast<const float4 *>(qc + 32 * t + 8 * q);
                h1[t][4 * q] = __fadd_rn(h1[t][4 * q], fp_interp3(a.x, bq.x, c.x, w1, w2, w3));
                h1[t][4 * q + 1] = __fadd_rn(h1[t][4 * q + 1], fp_interp3(a.y, bq.y, c.y, w1, w2, w3));
                h1[t][4 * q + 2] = __fadd_rn(h1[t][4 * q + 2], fp_interp3(a.z, bq.z, c.z, w1, w2, w3));
                h1[t][4 * q + 3] = __fadd_rn(h1[t][4 * q + 3], fp_interp3(a.w, bq.w, c.w, w1, w2, w3));
            }
        }
        if (ti > 0) {
            f32x16 x = gather(0);
            int slot = 0;
            for (int u = 0; u < ti; ++u) {
                f32x16 xn = x;
                if (u + 1 < ti) xn = gather(u + 1);
                const ActSplit xs = split_act(x);
#pragma unroll
                for (int t = 0; t < T1; ++t) {
                    h1[t] = stream_pair<false>(wbuf[stage & 1], slot, lane, xs, h1[t]);
                    if (++slot == kS) { slot = 0; PN2_NEXT_STAGE(); }
                }
                x = xn;
            }
            if (slot != 0) PN2_NEXT_STAGE();
        }
        ActSplit s1[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t) s1[t] = split_act(mlp_relu(h1[t]));

        // the last layer: operands swapped -> register v of lane (c, hh) holds point mlp_chan(v, hh) of the
        // item, channel 32t + c; bias + ReLU and a plain store (128 contiguous bytes per half wave)
        auto store_tile = [&](int t, const f32x16 &acc, const float *blast) __attribute__((always_inline)) {
            const int ch = 32 * t + s;
            const float bias = b3_at(blast, ch);
            if (item_ok && ch < cout) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const long long r = row0 + mlp_chan(v, h);
                    if (r < rows) out[r * cout + ch] = fmaxf(__fadd_rn(acc[v], bias), 0.0f);
                }
            }
        };
        if (T3 == 0) {
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
                for (int u = 0; u < T1; ++u) {
                    acc = stream_pair<true>(wbuf[stage & 1], (t * T1 + u) % kS, lane, s1[u], acc);
                    if ((t * T1 + u) % kS == kS - 1) PN2_NEXT_STAGE();
                }
                store_tile(t, acc, b2);
            }
        } else {
            ActSplit s2[T2];
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                f32x16 acc = mlp_bias(b2, t, h);
#pragma unroll
                for (int u = 0; u < T1; ++u) {
                    acc = stream_pair<false>(wbuf[stage & 1], (t * T1 + u) % kS, lane, s1[u], acc);
                    if ((t * T1 + u) % kS == kS - 1) PN2_NEXT_STAGE();
                }
                s2[t] = split_act(mlp_relu(acc));
            }
#pragma unroll
            for (int t = 0; t < (T3 ? T3 : 1); ++t) {
                f32x16 acc;
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
                for (int u = 0; u < T2; ++u) {
                    acc = stream_pair<true>(wbuf[stage & 1], (t * T2 + u) % kS, lane, s2[u], acc);
                    if ((t * T2 + u) % kS == kS - 1) PN2_NEXT_STAGE();
                }
                store_tile(t, acc, b3);
            }
        }
    }
#undef PN2_STREAM_ISSUE
#undef PN2_STREAM_COMMIT
#undef PN2_NEXT_STAGE
}

struct FpConfig { int ti, tif, t1, t2, t3; };        // ti: tiles of skip channels (c1), tif: tiles of known features (c2)

// tile shapes with an instantiated kernel; widths are padded up to them with zero weights
static bool fp_pick(int c2, int c1, int nlayers, const int *widths, FpConfig &cfg)
{
    if (c2 < 1 || c1 < 0 || (nlayers != 2 && nlayers != 3)) return false;
    for (int i = 0; i < nlayers; ++i)
        if (widths[i] < 1 || widths[i] > 256) return false;
    static const int kShapes[][3] = {{4, 4, 0}, {8, 4, 0}, {8, 8, 0}, {4, 4, 4}, {8, 4, 4}, {8, 8, 4}, {8, 8, 8}};
    for (const auto &sh : kShapes) {
        if ((sh[2] != 0) != (nlayers == 3)) continue;
        if (widths[0] <= 32 * sh[0] && widths[1] <= 32 * sh[1] && (nlayers == 2 || widths[2] <= 32 * sh[2])) {
            cfg = {(c1 + 31) / 32, (c2 + 31) / 32, sh[0], sh[1], sh[2]};
            return true;
        }
    }
    return false;
}

// packed weights: [this kernel's stream: skip rows of layer 1 (padded to a stage), layer 2, layer 3]
// [per-point kernel: known-feature rows of layer 1, one stream per 128 output channels]
static long long fp_main_pairs(const FpConfig &c) { return (long long)pad_to_stage(c.ti * c.t1) + c.t2 * c.t1 + c.t3 * c.t2; }
static long long fp_point_pairs(const FpConfig &c) { return (long long)c.tif * c.t1; }       // t1 is 4 or 8: whole stages
static long long fp_pairs(const FpConfig &c) { return fp_main_pairs(c) + fp_point_pairs(c); }
// bias: [layer 1][layer 2][layer 3][128 zeros: the per-point kernel adds no bias, the interpolation weights multiply Q only]
static long long fp_bias_floats(int t1, int t2, int t3) { return (long long)(t1 + t2 + t3) * 32 + 128; }

// Q = points2 . W1a: few known points (most FP levels) -> one workgroup per (item, output tile), no staging;
// many -> the streamed per-point kernel, 128 output channels per launch

static int fp_point_layer(int t1, long long known_rows, int c2, int tif, const float *points2, const float *wpoint,
                          const float *zero_bias, float *pre, hipStream_t st)
{
    if (point_layer_prefers_few_rows(known_rows, t1, tif)) return point_layer_few_rows_launch(t1, c2, known_rows, tif, points2, wpoint, nullptr, pre, st);
    for (int half = 0; half < t1 / 4; ++half)
        if (int rc = point_layer_launch(4, c2, known_rows, tif, points2, wpoint + (size_t)half * tif * 4 * kPairWords, zero_bias,
                                        pre, 32 * t1, 128 * half, st)) return rc;
    return PN2_OK;
}

template <int T1, int T2, int T3>
static int launch_fp(const FpConfig &c, long long rows, int b, int n, int m, int c2, int c1, int cout, const float *points2,
                     const float *points1, const int *idx, const float *dist, const float *wp, const float *bp, float *out,
                     float *pre, hipStream_t st)
{
    const float *wpoint = wp + (size_t)fp_main_pairs(c) * kPairWords, *zeros = bp + (size_t)(T1 + T2 + T3) * 32;
    if (int rc = fp_point_layer(T1, (long long)b * m, c2, c.tif, points2, wpoint, zeros, pre, st)) return rc;
    const long long groups = (rows + 31) / 32;
    long long blocks = (groups + 3) / 4;
    if (blocks > 256) blocks = 256;                               // persistent, one workgroup per CU: every workgroup streams the weights
    return launch((fp_mlp_stream_kernel<T1, T2, T3>), dim3((unsigned)blocks), dim3(kMlpThreads), 0, st, n, m, c1, cout,
                  rows, c.ti, (const float *)pre, points1, idx, dist, wp, bp, out);
}

}  // namespace pn2

// kind: 0 = one wave per 32 points, weights streamed through LDS (this file: many points);
//       1 = four waves per 32 points (coop_mlp.hip: few points, wide layers). Same results, different packing.
// tiles4 = {tiles of skip channels, tiles of layer 1, 2, 3}.
extern "C" int pn2_fp_mlp_config(int c2, int c1, int nlayers, const int *widths, int kind, int *tiles4, long long *w_floats,
                                 long long *b_floats)
{
    using namespace pn2;
    if (c2 <= 0 || c1 < 0 || !widths || (kind != 0 && kind != 1)) return PN2_E_ARG;
    if (kind == 1) {
        MlpCoopConfig cc;
        if (!mlp_coop_pick(c1, nlayers, widths, cc) || !mlp_coop_has_kernel(cc, 1)) return PN2_E_TOO_LARGE;
        if (tiles4) { tiles4[0] = cc.ti; tiles4[1] = 4 * cc.q1; tiles4[2] = 4 * cc.q2; tiles4[3] = 4 * cc.q3; }
        if (w_floats) *w_floats = (long long)mlp_coop_w_floats(cc) + (long long)((c2 + 31) / 32) * 4 * cc.q1 * kPairWords;
        if (b_floats) *b_floats = (long long)mlp_coop_b_floats(cc) + 128;
        return PN2_OK;
    }
    FpConfig c;
    if (!fp_pick(c2, c1, nlayers, widths, c)) return PN2_E_TOO_LARGE;
    if (tiles4) { tiles4[0] = c.ti; tiles4[1] = c.t1; tiles4[2] = c.t2; tiles4[3] = c.t3; }
    if (w_floats) *w_floats = fp_pairs(c) * kPairWords;
    if (b_floats) *b_floats = fp_bias_floats(c.t1, c.t2, c.t3);
    return PN2_OK;
}

// Scratch of a pn2_fp_mlp call: Q = points2 . W1a, one row of the padded first width per known point.
extern "C" long long pn2_fp_mlp_ws_bytes(int b, int m, int c2, int c1, int nlayers, const int *widths, int kind)
{
    using namespace pn2;
    if (b <= 0 || m <= 0 || c2 <= 0 || c1 < 0 || !widths || (kind != 0 && kind != 1)) return 0;
    int t1;
    if (kind == 1) {
        MlpCoopConfig cc;
        if (!mlp_coop_pick(c1, nlayers, widths, cc) || !mlp_coop_has_kernel(cc, 1)) return 0;
        t1 = 4 * cc.q1;
    } else {
        FpConfig c;
        if (!fp_pick(c2, c1, nlayers, widths, c)) return 0;
        t1 = c.t1;
    }
    return (long long)sizeof(float) * b * m * 32 * t1;
}

// Host code: permute (cin_i, cout_i) row-major weights (rows of layer 1 in the reference's concat order
// [interpolated, points1]) into the tile-pair streams the chosen kernel and the per-point kernel consume.
extern "C" int pn2_fp_mlp_pack(int c2, int c1, int nlayers, const int *widths, int kind, const float *const *w,
                               const float *const *bias, float *wpacked, float *bpacked)
{
    using namespace pn2;
    if (c2 <= 0 || c1 < 0 || !widths || !w || !bias || !wpacked || !bpacked) return PN2_E_NULL;
    if (kind != 0 && kind != 1) return PN2_E_ARG;
    const int tif = (c2 + 31) / 32;
    int *skip_row = (int *)malloc(sizeof(int) * (size_t)(c1 > 0 ? c1 : 1));
    for (int k = 0; k < c1; ++k) skip_row[k] = c2 + k;              // rows of w[0] behind the interpolated channels
    int t1, t2, t3;
    float *wp = wpacked, *bp = bpacked;
    if (kind == 1) {
        MlpCoopConfig cc;
        if (!mlp_coop_pick(c1, nlayers, widths, cc) || !mlp_coop_has_kernel(cc, 1)) { free(skip_row); return PN2_E_TOO_LARGE; }
        mlp_coop_pack(cc, c1, nlayers, widths, skip_row, w, bias, wpacked, bpacked, false);
        wp += mlp_coop_w_floats(cc);
        bp += mlp_coop_b_floats(cc);
        t1 = 4 * cc.q1; t2 = 4 * cc.q2; t3 = 4 * cc.q3;
    } else {
        FpConfig c;
        if (!fp_pick(c2, c1, nlayers, widths, c)) { free(skip_row); return PN2_E_TOO_LARGE; }
        for (int u = 0; u < c.ti; ++u)
            for (int t = 0; t < c.t1; ++t) wp = mlp_pack_pair_x6(wp, w[0], c1, widths[0], t, u, skip_row);
        for (int i = c.ti * c.t1; i < pad_to_stage(c.ti * c.t1); ++i)
            for (int j = 0; j < kPairWords; ++j) *wp++ = 0.0f;
        for (int t = 0; t < c.t2; ++t)
            for (int u = 0; u < c.t1; ++u) wp = mlp_pack_pair_x6(wp, w[1], widths[0], widths[1], t, u, nullptr);
        for (int t = 0; t < c.t3; ++t)
            for (int u = 0; u < c.t2; ++u) wp = mlp_pack_pair_x6(wp, w[2], widths[1], widths[2], t, u, nullptr);
        const int tout[3] = {c.t1, c.t2, c.t3};
        for (int L = 0; L < 3; ++L)
            for (int t = 0; t < tout[L]; ++t)
                for (int hh = 0; hh < 2; ++hh)
                    for (int v = 0; v < 16; ++v) {
                        const int ch = 32 * t + mlp_chan(v, hh);
                        *bp++ = (L < nlayers && ch < widths[L]) ? bias[L][ch] : 0.0f;
                    }
        t1 = c.t1; t2 = c.t2; t3 = c.t3;
    }
    (void)t2; (void)t3;
    free(skip_row);
    // the per-point kernel's streams: known-feature rows of layer 1, 128 output channels (4 tiles) per stream, feature
    // tiles outermost
    for (int half = 0; half < t1 / 4; ++half)
        for (int u = 0; u < tif; ++u)
            for (int t = 4 * half; t < 4 * half + 4; ++t) wp = mlp_pack_pair_x6(wp, w[0], c2, widths[0], t, u, nullptr);
    for (int i = 0; i < 128; ++i) *bp++ = 0.0f;
    return PN2_OK;
}

extern "C" int pn2_fp_mlp(int b, int n, int m, int c2, int c1, const float *points2, const float *points1, const int *idx,
                          const float *dist, int nlayers, const int *widths, int kind, const float *wpacked,
                          const float *bpacked, float *out, void *ws, void *stream)
{
    using namespace pn2;
    if (b < 0 || n < 0 || m <= 0 || c2 <= 0 || c1 < 0) return PN2_E_SHAPE;
    if (!widths) return PN2_E_NULL;
    if (kind != 0 && kind != 1) return PN2_E_ARG;
    FpConfig c;
    MlpCoopConfig cc;
    if (kind == 0 ? !fp_pick(c2, c1, nlayers, widths, c)
                  : !(mlp_coop_pick(c1, nlayers, widths, cc) && mlp_coop_has_kernel(cc, 1))) return PN2_E_TOO_LARGE;
    const long long rows = (long long)b * n;
    if (rows == 0) return PN2_OK;
    if (!points2 || (c1 > 0 && !points1) || !idx || !dist || !wpacked || !bpacked || !out || !ws) return PN2_E_NULL;
    if ((long long)m * c2 > INT_MAX) return PN2_E_TOO_LARGE;
    const int cout = widths[nlayers - 1];
    hipStream_t st = as_stream(stream);
    if (kind == 1) {
        const int t1 = 4 * cc.q1, tif = (c2 + 31) / 32;
        const float *wpoint = wpacked + mlp_coop_w_floats(cc), *zeros = bpacked + mlp_coop_b_floats(cc);
        if (int rc = fp_point_layer(t1, (long long)b * m, c2, tif, points2, wpoint, zeros, (float *)ws, st)) return rc;
        CoopParams p = {n, m, 0, c2, c1, cout, cc.ti, rows, nullptr, nullptr, (const float *)ws, c1 > 0 ? points1 : nullptr, idx, dist,
                        wpacked, bpacked, out, 0};
        return mlp_coop_launch(cc, 1, p, st, nullptr);
    }
#define PN2_FP_CASE(A, B, C)                                                                                           \
    if (c.t1 == A && c.t2 == B && c.t3 == C)                                                                            \
        return launch_fp<A, B, C>(c, rows, b, n, m, c2, c1, cout, points2, points1, idx, dist, wpacked, bpacked, out, \
                                  (float *)ws, st)
    PN2_FP_CASE(4, 4, 0);
    PN2_FP_CASE(8, 4, 0);
    PN2_FP_CASE(8, 8, 0);
    PN2_FP_CASE(4, 4, 4);
    PN2_FP_CASE(8, 4, 4);
    PN2_FP_CASE(8, 8, 4);
    PN2_FP_CASE(8, 8, 8);
#undef PN2_FP_CASE
    return PN2_E_TOO_LARGE;
}
