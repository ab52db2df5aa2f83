// coop_mlp.hip -- the fused MLP kernels for WIDE layer stacks over FEW rows: the four waves of a
// workgroup share ONE work item (32 samples) and split every layer's OUTPUT tiles between them.
//
// What it covers (reference call sites):
//   * set-abstraction stacks beyond the streamed kernel's (128,128,256): sem_seg's last level
//     pointnet_sa_module(..., mlp=[256,256,512], ...)  (models/pointnet2_sem_seg.py:31) and the group_all
//     level mlp=[256,512,1024] of the classification / part-segmentation nets (pointnet2_cls_ssg.py:34,
//     pointnet2_cls_msg.py:29, pointnet2_part_seg.py:28; sample_and_group_all, utils/pointnet_util.py:59-84:
//     one group holding every point, no centroid subtraction) -- conv2d 1x1 + batch_norm + ReLU x 3 and
//     reduce_max over the group (pointnet_util.py:117-127), any nsample (the last 32-sample part is masked);
//   * feature-propagation layers with few unknown points (pointnet_fp_module, pointnet_util.py:199-229;
//     sem_seg FP1-FP3, part_seg FP1-FP2: 512-8192 points, up to 1280 input channels), see fp_mlp.hip.
// With one wave per item (sa_mlp_stream.hip, fp_mlp.hip) such a level is a handful of serial MFMA chains --
// sem_seg FP1 is 16 items of 256 tile pairs each: 138 us with 16 waves busy on the whole GPU. Here a wave
// computes tiles t = w, w+4, ... of every layer, so an item's chain is four times shorter and four times as
// many SIMDs work; the hidden activations are exchanged through LDS in their three-level bf16 operand form
// (split ONCE by the wave that produced the tile; a layer's C/D layout IS the next layer's operand slot
// layout, sa_mlp.hip), one s_barrier per layer and item.
// Every wave consumes DIFFERENT weight tile pairs, so nothing is shared through LDS: each wave streams its
// own 6 KiB pairs from L2 straight into registers, one pair ahead of the MFMAs (the packed array is the
// exact per-wave consumption order). Inputs are gathered by all four waves (redundant L2 reads, negligible
// at these sizes). The last layer runs with swapped operands: a lane holds 16 samples of one channel ->
// lane-local max-pool, or 128-byte row segments for the plain store of a feature-propagation layer.
#include "sa_mlp_common.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>

namespace pn2 {

constexpr int kGatherGrouped = 0, kGatherInterp = 1;

__device__ __forceinline__ float coop_interp3(float p1, float p2, float p3, float w1, float w2, float w3)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(p1, w1), __fmul_rn(p2, w2)), __fmul_rn(p3, w3));   // tf_interpolate.cpp:122
}

// Q1, Q2, Q3: output tiles PER WAVE of the three layers (layer widths 128 * Q); Q3 == 0: two layers.
// SPLIT_OUT (two grouped layers): the second layer is a HIDDEN layer whose output tiles leave for global memory
// in their three-level operand form, [unit][tile][e][level][lane] -- the A operand of pool_gemm_kernel below,
// which runs the wide last layer of such a stack as a tiled GEMM.
template <int GATHER, int Q1, int Q2, int Q3, bool SPLIT_OUT = false>
__global__ __launch_bounds__(kMlpThreads) void coop_mlp_kernel(CoopParams p)
{
    constexpr int T1 = 4 * Q1, T2 = 4 * Q2, T3 = 4 * Q3;
    constexpr bool THREE = Q3 > 0;
    static_assert(!SPLIT_OUT || (Q3 == 0 && GATHER == kGatherGrouped), "split output: two grouped layers");
    constexpr int QL = THREE ? Q3 : Q2;                                   // the last layer's tiles per wave
    constexpr bool POOL = GATHER == kGatherGrouped;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kTileVec = kPairWords / 4;                               // a split activation tile: [e][level][64] vectors
    u32x4 *act1 = reinterpret_cast<u32x4 *>(smem);                         // [T1] tiles
    u32x4 *act2 = act1 + T1 * kTileVec;                                    // [T2] tiles (three layers only)
    float *bias_s = reinterpret_cast<float *>(act2 + (THREE ? T2 * kTileVec : 0));
    const float *b1 = bias_s, *b2 = b1 + T1 * 32, *b3 = b2 + T2 * 32;
    const float *blast = THREE ? b3 : b2;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (T1 + T2 + T3) * 32; i += kMlpThreads) bias_s[i] = p.bp[i];
    __syncthreads();

    // ---- this wave's weight stream: pair number 4 * k + w of the packed array, k = 0 .. per_item - 1 per item
    const int per_item = p.ti * Q1 + T1 * Q2 + T2 * Q3;
    const u32x4 *wp4 = reinterpret_cast<const u32x4 *>(p.wp) + (size_t)w * kTileVec + lane;
    u32x4 nx0, nx1, nx2, nx3, nx4, nx5;                                   // the NEXT pair, in flight: [e][level]
    int k = 0;                                                            // pair counter within the item
    auto issue = [&]() __attribute__((always_inline)) {
        const u32x4 *q = wp4 + (size_t)(k == per_item ? 0 : k) * (4 * kTileVec);   // 4 pairs per step; wraps to the next item
        nx0 = q[0]; nx1 = q[64]; nx2 = q[128]; nx3 = q[192]; nx4 = q[256]; nx5 = q[320];
    };
#define PN2_COOP_PAIR(SWAP, ACT, ACC)                                                                              \
    do {                                                                                                           \
        const u32x4 w0_[3] = {nx0, nx1, nx2}, w1_[3] = {nx3, nx4, nx5};                                            \
        ++k;                                                                                                       \
        issue();                                                                                                   \
        ACC = mma_x6<SWAP>(w0_, (ACT).p[0], ACC);                                                                  \
        ACC = mma_x6<SWAP>(w1_, (ACT).p[1], ACC);                                                                  \
    } while (0)
    issue();

    auto act_load = [&](const u32x4 *act, int u) __attribute__((always_inline)) -> ActSplit {
        const u32x4 *a = act + u * kTileVec + lane;
        ActSplit x;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int l = 0; l < 3; ++l) x.p[e][l] = a[(e * 3 + l) * 64];
        return x;
    };
    auto act_store = [&](u32x4 *act, int t, const f32x16 &v) __attribute__((always_inline)) {
        const ActSplit x = split_act(v);
        u32x4 *a = act + t * kTileVec + lane;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int l = 0; l < 3; ++l) a[(e * 3 + l) * 64] = x.p[e][l];
    };

    const int parts = POOL ? (p.nsample + 31) / 32 : 1;
    // p.split: the 32-sample parts of a group are separate work units (few, large groups -- the group_all level:
    // 32 clouds x 4 parts), merged with an integer atomic max on the pre-zeroed output (values are >= 0 after ReLU)
    const int parts_in = p.split ? 1 : parts;                             // SPLIT_OUT always runs with p.split
    const long long units = POOL ? (p.split ? p.rows * parts : p.rows) : (p.rows + 31) / 32;
    const int cin = POOL ? p.cf + 3 : p.c1;                               // FP: layer 1's MFMA input is the skip link only

    for (long long unit_raw = blockIdx.x; unit_raw < units; unit_raw += gridDim.x) {
        const long long unit = p.split ? unit_raw / parts : unit_raw;       // centroid (SA) / group of 32 points (FP)
        float best[QL];
#pragma unroll
        for (int g = 0; g < QL; ++g) best[g] = -INFINITY;
        for (int part_i = 0; part_i < parts_in; ++part_i) {
            const int part = p.split ? (int)(unit_raw % parts) : part_i;
            k = 0;
            // ---- per-lane gather context ----------------------------------------------------------------
            const float *gf = nullptr, *gx = nullptr, *gc = nullptr;       // grouped: feature row, xyz row, centroid
            const float *pa = nullptr, *pb = nullptr, *pc = nullptr, *p1 = nullptr;
            float w1 = 0.f, w2 = 0.f, w3 = 0.f;
            if (POOL) {
                const long long cloud = unit / p.m;
                const int smp = part * 32 + s;
                int pt;
                if (p.idx) pt = p.idx[unit * p.nsample + min(smp, p.nsample - 1)];
                else pt = min(smp, p.n - 1);                               // group_all: the group IS the cloud
                gf = p.feat + ((size_t)cloud * p.n + pt) * p.cf;
                gx = p.xyz + ((size_t)cloud * p.n + pt) * 3;
                gc = p.new_xyz ? p.new_xyz + unit * 3 : nullptr;
            } else {
                const long long row = min(unit * 32 + s, p.rows - 1);
                const long long cloud = row / p.n;
                const int *ip = p.idx + row * 3;
                const float *dp = p.dist + row * 3;
                // inverse-distance weights, pointnet_util.py:212-215
                const float r1 = __fdiv_rn(1.0f, fmaxf(dp[0], 1e-10f)), r2 = __fdiv_rn(1.0f, fmaxf(dp[1], 1e-10f)),
                            r3 = __fdiv_rn(1.0f, fmaxf(dp[2], 1e-10f));
                const float norm = __fadd_rn(__fadd_rn(r1, r2), r3);
                w1 = __fdiv_rn(r1, norm); w2 = __fdiv_rn(r2, norm); w3 = __fdiv_rn(r3, norm);
                // p.feat: Q = points2 . W1a (fp_mlp.hip), one row of 32 * T1 floats per known point
                const float *base2 = p.feat + (size_t)cloud * p.m * (32 * T1) + 4 * h;
                pa = base2 + (size_t)ip[0] * (32 * T1); pb = base2 + (size_t)ip[1] * (32 * T1); pc = base2 + (size_t)ip[2] * (32 * T1);
                p1 = p.skip ? p.skip + (size_t)row * p.c1 : nullptr;
            }
            const bool vec_a = (p.cf & 3) == 0, vec_b = (p.c1 & 3) == 0;
            // one 32-channel tile of layer-1 inputs: register v <- channel 32u + mlp_chan(v, h).
            // grouped: [features (cf), xyz - centroid (3)]; feature propagation: the skip link points1 (c1)
            auto gather = [&](int u) __attribute__((always_inline)) -> f32x16 {
                f32x16 x;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k0 = 32 * u + 8 * q + 4 * h;
                    if (POOL) {
                        if (vec_a && k0 + 3 < p.cf) {
                            const float4 f = *reinterpret_cast<const float4 *>(gf + k0);
                            x[4 * q] = f.x; x[4 * q + 1] = f.y; x[4 * q + 2] = f.z; x[4 * q + 3] = f.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int kk = k0 + r;
                                float val = 0.0f;
                                if (kk < p.cf) val = gf[kk];
                                else if (kk < cin) val = gc ? __fsub_rn(gx[kk - p.cf], gc[kk - p.cf]) : gx[kk - p.cf];
                                x[4 * q + r] = val;
                            }
                        }
                    } else {
                        if (vec_b && k0 + 3 < cin) {
                            const float4 f = *reinterpret_cast<const float4 *>(p1 + k0);
                            x[4 * q] = f.x; x[4 * q + 1] = f.y; x[4 * q + 2] = f.z; x[4 * q + 3] = f.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) x[4 * q + r] = k0 + r < cin ? p1[k0 + r] : 0.0f;
                        }
                    }
                }
                return x;
            };

            // ---- layer 1: this wave's tiles t = 4g + w; input tiles outermost ---------------------------------
            f32x16 a1[Q1];
#pragma unroll
            for (int g = 0; g < Q1; ++g) {
                a1[g] = mlp_bias(b1, 4 * g + w, h);
                if (!POOL) {                                               // + the interpolated rows of Q, this wave's channels
                    const int t = 4 * g + w;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = *reinterpret_cast<const float4 *>(pa + 32 * t + 8 * q), bq = *reinterpret_cast<const float4 *>(pb + 32 * t + 8 * q),
                                     c = *reinterpret_cast<const float4 *>(pc + 32 * t + 8 * q);
                        a1[g][4 * q] = __fadd_rn(a1[g][4 * q], coop_interp3(a.x, bq.x, c.x, w1, w2, w3));
                        a1[g][4 * q + 1] = __fadd_rn(a1[g][4 * q + 1], coop_interp3(a.y, bq.y, c.y, w1, w2, w3));
                        a1[g][4 * q + 2] = __fadd_rn(a1[g][4 * q + 2], coop_interp3(a.z, bq.z, c.z, w1, w2, w3));
                        a1[g][4 * q + 3] = __fadd_rn(a1[g][4 * q + 3], coop_interp3(a.w, bq.w, c.w, w1, w2, w3));
                    }
                }
            }
            f32x16 x;
            if (p.ti > 0) x = gather(0);
            for (int u = 0; u < p.ti; ++u) {
                f32x16 xn = x;
                if (u + 1 < p.ti) xn = gather(u + 1);
                const ActSplit xs = split_act(x);
#pragma unroll
                for (int g = 0; g < Q1; ++g) PN2_COOP_PAIR(false, xs, a1[g]);
                x = xn;
            }
#pragma unroll
            for (int g = 0; g < Q1; ++g) act_store(act1, 4 * g + w, mlp_relu(a1[g]));
            __syncthreads();

            // ---- the last layer on `src` (TS input tiles): swapped operands; pool or store -----------------
            auto last_layer = [&](const u32x4 *src, const int TS) __attribute__((always_inline)) {
                f32x16 acc[QL];
#pragma unroll
                for (int g = 0; g < QL; ++g)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[g][v] = 0.0f;
                ActSplit xa = act_load(src, 0);
                for (int u = 0; u < TS; ++u) {
                    ActSplit xb = xa;
                    if (u + 1 < TS) xb = act_load(src, u + 1);
#pragma unroll
                    for (int g = 0; g < QL; ++g) PN2_COOP_PAIR(true, xa, acc[g]);
                    xa = xb;
                }
                // register v of lane (c, hh) holds sample mlp_chan(v, hh) of the item, channel 32 (4g + w) + c
#pragma unroll
                for (int g = 0; g < QL; ++g) {
                    if (POOL) {
                        float mx = -INFINITY;
#pragma unroll
                        for (int v = 0; v < 16; ++v)
                            if (part * 32 + mlp_chan(v, h) < p.nsample) mx = fmaxf(mx, acc[g][v]);   // masked tail of the group
                        best[g] = fmaxf(best[g], mx);
                    } else {
                        const int ch = 32 * (4 * g + w) + s;
                        const float bias = b3_at(blast, ch);
                        if (ch < p.cout) {
#pragma unroll
                            for (int v = 0; v < 16; ++v) {
                                const long long r = unit * 32 + mlp_chan(v, h);
                                if (r < p.rows) p.out[r * p.cout + ch] = fmaxf(__fadd_rn(acc[g][v], bias), 0.0f);
                            }
                        }
                    }
                }
            };

            if (!THREE && !SPLIT_OUT) {
                last_layer(act1, T1);
                __syncthreads();                                              // act1 is rewritten by the next item
            } else {
                // ---- layer 2 ---------------------------------------------------------------------------------
                f32x16 a2[Q2];
#pragma unroll
                for (int g = 0; g < Q2; ++g) a2[g] = mlp_bias(b2, 4 * g + w, h);
                ActSplit xa = act_load(act1, 0);
                for (int u = 0; u < T1; ++u) {
                    ActSplit xb = xa;
                    if (u + 1 < T1) xb = act_load(act1, u + 1);
#pragma unroll
                    for (int g = 0; g < Q2; ++g) PN2_COOP_PAIR(false, xa, a2[g]);
                    xa = xb;
                }
                if (SPLIT_OUT) {
                    u32x4 *dst = reinterpret_cast<u32x4 *>(p.out) + (size_t)unit_raw * T2 * kTileVec;
#pragma unroll
                    for (int g = 0; g < Q2; ++g) act_store(dst, 4 * g + w, mlp_relu(a2[g]));
                    __syncthreads();                                          // everyone is done reading act1
                } else {
#pragma unroll
                    for (int g = 0; g < Q2; ++g) act_store(act2, 4 * g + w, mlp_relu(a2[g]));
                    __syncthreads();                                          // also: everyone is done reading act1
                    last_layer(act2, T2);
                    // act2 is rewritten only after the next item's first barrier: no barrier needed here
                }
            }
        }
        if (POOL && !SPLIT_OUT) {
#pragma unroll
            for (int g = 0; g < QL; ++g) {
                const int ch = 32 * (4 * g + w) + s;
                const float mx = fmaxf(best[g], __shfl_xor(best[g], 32));
                if (h == 0 && ch < p.cout) {
                    const float val = fmaxf(__fadd_rn(mx, b3_at(blast, ch)), 0.0f);
                    if (p.split) atomicMax(reinterpret_cast<int *>(p.out + unit * p.cout + ch), __float_as_int(val));
                    else p.out[unit * p.cout + ch] = val;
                }
            }
        }
    }
#undef PN2_COOP_PAIR
}

// ---- the wide last layer of a (c1, c2, c3 > 512) stack as a tiled GEMM + max-pool ------------------------------
// With one item per workgroup the cooperative kernel streams EVERY weight of the stack through every CU's vector
// memory path for 32 rows of work (64 B/clk per CU at full MFMA rate: it is L1-bound), and the group_all level --
// 73 % of it the 512 x 1024 last layer -- has only 128 items. Here that layer is a GEMM: a workgroup owns one group
// (centroid / cloud) x 128 output channels; per 32-channel step of the contraction it stages the group's input tiles
// (already in three-level operand form, written by coop_mlp_kernel<..., SPLIT_OUT>) and its slice of the weights into
// LDS, double-buffered through registers like the streamed kernels, and every weight fragment serves four row tiles.
// Wave w: row tile w & 3 of the current chunk of four, output tiles 2 (w >> 2) and 2 (w >> 2) + 1 of the block; swapped
// operands (lane = channel), so the pool is lane-local + one LDS atomic max per channel on relu(x + bias) >= 0.
constexpr int kGemmThreads = 512, kGemmColTiles = 4, kGemmRowTiles = 4;
constexpr int kGemmStageVec = (kGemmColTiles + kGemmRowTiles) * (kPairWords / 4);   // 48 KiB per contraction step

__global__ __launch_bounds__(kGemmThreads) void pool_gemm_kernel(int parts, int tk, int cout, long long groups,
                                                                const float *__restrict__ asplit,
                                                                const float *__restrict__ wgemm,
                                                                const float *__restrict__ bias, float *__restrict__ out)
{
    constexpr int kTileVec = kPairWords / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *buf0 = reinterpret_cast<u32x4 *>(smem), *buf1 = buf0 + kGemmStageVec;
    int *colmax = reinterpret_cast<int *>(buf1 + kGemmStageVec);          // [32 * kGemmColTiles]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), rt = w & 3, cp = w >> 2;
    const int colblocks = (cout + 32 * kGemmColTiles - 1) / (32 * kGemmColTiles);
    static_assert(kGemmStageVec == 6 * kGemmThreads, "six staging vectors per thread: three of A, three of W");
    // this thread's six vectors of a stage: vector i = tid + j * 512; i < 4 tiles: row tile i / kTileVec of the chunk
    // (a tail chunk repeats the last part: harmless under max), else the weights of the contraction step
    const int ar0 = tid / kTileVec, ao0 = tid % kTileVec;
    const int ar1 = (tid + kGemmThreads) / kTileVec, ao1 = (tid + kGemmThreads) % kTileVec;
    const int ar2 = (tid + 2 * kGemmThreads) / kTileVec, ao2 = (tid + 2 * kGemmThreads) % kTileVec;
    for (long long job = blockIdx.x; job < groups * colblocks; job += gridDim.x) {
        const long long grp = job / colblocks;
        const int cb = (int)(job % colblocks);
        for (int i = tid; i < 32 * kGemmColTiles; i += kGemmThreads) colmax[i] = 0;
        const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(wgemm) + (size_t)cb * tk * (kGemmColTiles * kTileVec) + tid;
        for (int chunk = 0; chunk < parts; chunk += kGemmRowTiles) {
            const u32x4 *a0 = reinterpret_cast<const u32x4 *>(asplit) + (size_t)(grp * parts + min(chunk + ar0, parts - 1)) * tk * kTileVec + ao0;
            const u32x4 *a1 = reinterpret_cast<const u32x4 *>(asplit) + (size_t)(grp * parts + min(chunk + ar1, parts - 1)) * tk * kTileVec + ao1;
            const u32x4 *a2 = reinterpret_cast<const u32x4 *>(asplit) + (size_t)(grp * parts + min(chunk + ar2, parts - 1)) * tk * kTileVec + ao2;
            // two stages in flight in registers (sets E / O for even / odd contraction steps): a step is ~0.8 us of MFMA
            // work per SIMD, less than an L2 round trip under load, so one stage ahead left the loads exposed
            u32x4 e0, e1, e2, e3, e4, e5, o0, o1, o2, o3, o4, o5;
#define PN2_GEMM_ISSUE(u, S)                                                                                          \
    do {                                                                                                              \
        S##0 = a0[(size_t)(u) * kTileVec]; S##1 = a1[(size_t)(u) * kTileVec]; S##2 = a2[(size_t)(u) * kTileVec];       \
        const u32x4 *ws_ = wsrc + (size_t)(u) * (kGemmColTiles * kTileVec);                                           \
        S##3 = ws_[0]; S##4 = ws_[kGemmThreads]; S##5 = ws_[2 * kGemmThreads];                                        \
    } while (0)
#define PN2_GEMM_COMMIT(B, S)                                                                                         \
    do {                                                                                                              \
        u32x4 *d_ = (B) + tid;                                                                                        \
        d_[0] = S##0; d_[kGemmThreads] = S##1; d_[2 * kGemmThreads] = S##2;                                           \
        d_[3 * kGemmThreads] = S##3; d_[4 * kGemmThreads] = S##4; d_[5 * kGemmThreads] = S##5;                        \
    } while (0)
#define PN2_GEMM_STEP(cur)                                                                                            \
    do {                                                                                                              \
        const u32x4 *a = (cur) + rt * kTileVec + lane;                                                                \
        const u32x4 *b0 = (cur) + (kGemmRowTiles + 2 * cp) * kTileVec + lane;                                         \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                               \
            const u32x4 x[3] = {a[(e * 3 + 0) * 64], a[(e * 3 + 1) * 64], a[(e * 3 + 2) * 64]};                       \
            _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                           \
                const u32x4 *b = b0 + c * kTileVec;                                                                   \
                const u32x4 wv[3] = {b[(e * 3 + 0) * 64], b[(e * 3 + 1) * 64], b[(e * 3 + 2) * 64]};                  \
                acc[c] = mma_x6<true>(wv, x, acc[c]);                                                                 \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
            f32x16 acc[2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[c][v] = 0.0f;
            // (every ISSUE is unconditional, clamped to the last step: behind a branch the compiler must assume the loads
            // may not have been issued and makes the COMMIT of the OTHER set wait for them too -- vmcnt counts in order)
            const int last = tk - 1;
            PN2_GEMM_ISSUE(0, e);
            PN2_GEMM_ISSUE(min(1, last), o);
            __syncthreads();                                   // the previous chunk's readers are done with both buffers
            PN2_GEMM_COMMIT(buf0, e);
            __syncthreads();
            PN2_GEMM_ISSUE(min(2, last), e);
            // A stage is WRITTEN to LDS before the MFMAs of the step that precedes it (its registers were loaded a whole
            // step earlier): the barrier keeps the eight waves in step, so writes issued after the MFMAs would be 600
            // LDS cycles in which every matrix pipe of the CU idles.
            for (int u = 0; u < tk; u += 2) {
                PN2_GEMM_COMMIT(buf1, o);                         // stage u + 1 (buf1's readers finished before the last barrier)
                PN2_GEMM_STEP(buf0);                              // step u
                __syncthreads();
                PN2_GEMM_ISSUE(min(u + 3, last), o);
                PN2_GEMM_COMMIT(buf0, e);                         // stage u + 2 (tk is even: 4 tiles per 128 channels)
                PN2_GEMM_STEP(buf1);                              // step u + 1
                __syncthreads();
                PN2_GEMM_ISSUE(min(u + 4, last), e);
            }
#undef PN2_GEMM_ISSUE
#undef PN2_GEMM_COMMIT
#undef PN2_GEMM_STEP
            // register v of lane (c, hh) holds row mlp_chan(v, hh) of the row tile, channel 32 (block tile) + c
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float mx = acc[c][0];
#pragma unroll
                for (int v = 1; v < 16; ++v) mx = fmaxf(mx, acc[c][v]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const int col = 32 * (2 * cp + c) + s, ch = 32 * kGemmColTiles * cb + col;
                if (h == 0 && ch < cout) atomicMax(&colmax[col], __float_as_int(fmaxf(__fadd_rn(mx, bias[ch]), 0.0f)));
            }
        }
        __syncthreads();
        for (int i = tid; i < 32 * kGemmColTiles; i += kGemmThreads) {
            const int ch = 32 * kGemmColTiles * cb + i;
            if (ch < cout) out[grp * cout + ch] = __int_as_float(colmax[i]);
        }
        __syncthreads();
    }
}

// ---- host side ------------------------------------------------------------------------------------------
bool mlp_coop_pick(int cin, int nlayers, const int *widths, MlpCoopConfig &cfg)
{
    if (cin < 0 || (nlayers != 2 && nlayers != 3)) return false;          // cin == 0: an FP level without skip link
    int q[3] = {0, 0, 0};
    for (int i = 0; i < nlayers; ++i) {
        if (widths[i] < 1 || widths[i] > 1024) return false;
        q[i] = (widths[i] + 127) / 128;
    }
    // hidden activations live in LDS: 6 KiB per 32-channel tile (three bf16 levels)
    const int hidden_tiles = 4 * q[0] + (nlayers == 3 ? 4 * q[1] : 0);
    if (hidden_tiles > 24) return false;
    cfg = {(cin + 31) / 32, q[0], q[1], q[2]};
    return true;
}

// the (.., .., > 512) grouped stacks run their last layer in pool_gemm_kernel: the packed array then holds the
// cooperative kernel's stream of layers 1-2 followed by the last layer in GEMM order (same number of pairs)
bool mlp_coop_gemm_last(const MlpCoopConfig &c, int fp) { return !fp && c.q3 == 8; }

static long long coop_pairs(const MlpCoopConfig &c)
{
    return 4ll * ((long long)c.ti * c.q1 + 4ll * c.q1 * c.q2 + 4ll * c.q2 * c.q3);
}
size_t mlp_coop_w_floats(const MlpCoopConfig &c) { return (size_t)coop_pairs(c) * kPairWords; }
size_t mlp_coop_ws_bytes(const MlpCoopConfig &c, int fp, long long rows, int nsample)
{
    if (!mlp_coop_gemm_last(c, fp)) return 0;
    return sizeof(float) * kPairWords * (size_t)rows * ((nsample + 31) / 32) * (4 * c.q2);
}
size_t mlp_coop_b_floats(const MlpCoopConfig &c) { return (size_t)(4 * (c.q1 + c.q2 + c.q3)) * 32; }

// krow: permutation of the first layer's weight rows (kernel channel order -> caller's row), or nullptr
void mlp_coop_pack(const MlpCoopConfig &c, int cin, int nlayers, const int *widths, const int *krow, const float *const *ws,
                   const float *const *bs, float *wpacked, float *bpacked, bool krow_is_grouped)
{
    float *wp = wpacked;
    const int tin[3] = {c.ti, 4 * c.q1, 4 * c.q2}, qq[3] = {c.q1, c.q2, c.q3};
    const int kin[3] = {cin, widths[0], nlayers > 1 ? widths[1] : 0};
    const bool gemm_last = krow_is_grouped && mlp_coop_gemm_last(c, 0);
    for (int L = 0; L < nlayers; ++L) {
        if (L == 2 && gemm_last) {                             // [column block][contraction tile][tile of the block]
            for (int cb = 0; cb < 4 * qq[2] / kGemmColTiles; ++cb)
                for (int u = 0; u < tin[2]; ++u)
                    for (int ct = 0; ct < kGemmColTiles; ++ct)
                        wp = mlp_pack_pair_x6(wp, ws[2], kin[2], widths[2], kGemmColTiles * cb + ct, u, nullptr);
            continue;
        }
        for (int u = 0; u < tin[L]; ++u)                       // input tiles outermost, then the wave's tile groups,
            for (int g = 0; g < qq[L]; ++g)                    // then the four waves: pair 4k + w belongs to wave w
                for (int wv = 0; wv < 4; ++wv)
                    wp = mlp_pack_pair_x6(wp, ws[L], kin[L], widths[L], 4 * g + wv, u, L == 0 ? krow : nullptr);
    }
    float *bp = bpacked;
    for (int L = 0; L < 3; ++L) {
        if (L == 2 && gemm_last) {                             // plain channel order
            for (int ch = 0; ch < 32 * 4 * qq[2]; ++ch) *bp++ = ch < widths[2] ? bs[2][ch] : 0.0f;
            continue;
        }
        for (int t = 0; t < 4 * qq[L]; ++t)
            for (int hh = 0; hh < 2; ++hh)
                for (int v = 0; v < 16; ++v) {
                    const int ch = 32 * t + mlp_chan(v, hh);
                    *bp++ = (L < nlayers && ch < widths[L]) ? bs[L][ch] : 0.0f;
                }
    }
}

template <int GATHER, int Q1, int Q2, int Q3>
static int launch_coop(const CoopParams &p, long long units, hipStream_t st)
{
    const size_t lds = sizeof(float) * kPairWords * (size_t)(4 * Q1 + (Q3 > 0 ? 4 * Q2 : 0)) + sizeof(float) * 32 * (size_t)(4 * (Q1 + Q2 + Q3));
    auto kern = coop_mlp_kernel<GATHER, Q1, Q2, Q3>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    long long blocks = units < 512 ? units : 512;
    return launch(kern, dim3((unsigned)blocks), dim3(kMlpThreads), lds, st, p);
}

// layers 1-2 by the cooperative kernel into `ws` (three-level operand tiles), the last layer + pool as a GEMM
static int launch_gemm_last(const MlpCoopConfig &c, const CoopParams &p_in, void *ws, hipStream_t st)
{
    if (!ws) return PN2_E_NULL;
    CoopParams p = p_in;
    const int parts = (p.nsample + 31) / 32, t2 = 4 * c.q2;
    p.split = 1;
    float *final_out = p.out;
    p.out = (float *)ws;
    {
        auto kern = coop_mlp_kernel<kGatherGrouped, 2, 4, 0, true>;
        const size_t lds = sizeof(float) * kPairWords * (size_t)(4 * 2) + sizeof(float) * 32 * (size_t)(4 * (2 + 4));
        if (int rc = allow_dynamic_lds(kern, lds)) return rc;
        const long long units = p.rows * parts;
        if (int rc = launch(kern, dim3((unsigned)(units < 512 ? units : 512)), dim3(kMlpThreads), lds, st, p)) return rc;
    }
    const size_t two_layers = (size_t)kPairWords * 4 * (size_t)(c.ti * c.q1 + 4 * c.q1 * c.q2);      // words of layers 1-2
    const float *wgemm = p.wp + two_layers, *b3 = p.bp + 32 * (size_t)(4 * (c.q1 + c.q2));
    const size_t lds = sizeof(u32x4) * 2 * kGemmStageVec + sizeof(int) * 32 * kGemmColTiles;
    if (int rc = allow_dynamic_lds(pool_gemm_kernel, lds)) return rc;
    const long long jobs = p.rows * ((p.cout + 32 * kGemmColTiles - 1) / (32 * kGemmColTiles));
    return launch(pool_gemm_kernel, dim3((unsigned)(jobs < 1024 ? jobs : 1024)), dim3(kGemmThreads), lds, st, parts, t2, p.cout,
                  p.rows, (const float *)ws, wgemm, b3, final_out);
}

int mlp_coop_launch(const MlpCoopConfig &c, int fp, const CoopParams &p_in, hipStream_t st, void *ws)
{
    if (mlp_coop_gemm_last(c, fp)) return launch_gemm_last(c, p_in, ws, st);
    CoopParams p = p_in;
    const int parts = (p.nsample + 31) / 32;
    p.split = (!fp && parts > 1 && p.rows < 256) ? 1 : 0;          // few large groups: one work unit per 32-sample part
    if (p.split) {
        if (int rc = clear_async(p.out, sizeof(float) * (size_t)p.rows * p.cout, st)) return rc;
    }
    const long long units = fp ? (p.rows + 31) / 32 : (p.split ? p.rows * parts : p.rows);
#define PN2_COOP_CASE(G, A, B, C) \
    if (fp == (G == kGatherInterp) && c.q1 == A && c.q2 == B && c.q3 == C) return launch_coop<G, A, B, C>(p, units, st)
    PN2_COOP_CASE(kGatherGrouped, 1, 1, 2);
    PN2_COOP_CASE(kGatherGrouped, 2, 2, 4);
    PN2_COOP_CASE(kGatherInterp, 1, 1, 0);
    PN2_COOP_CASE(kGatherInterp, 2, 1, 0);
    PN2_COOP_CASE(kGatherInterp, 2, 2, 0);
    PN2_COOP_CASE(kGatherInterp, 1, 1, 1);
    PN2_COOP_CASE(kGatherInterp, 2, 2, 2);
#undef PN2_COOP_CASE
    return PN2_E_TOO_LARGE;
}

bool mlp_coop_has_kernel(const MlpCoopConfig &c, int fp)
{
    static const int sa[][3] = {{1, 1, 2}, {2, 2, 4}, {2, 4, 8}};
    static const int fpk[][3] = {{1, 1, 0}, {2, 1, 0}, {2, 2, 0}, {1, 1, 1}, {2, 2, 2}};
    if (fp) {
        for (const auto &k : fpk) if (c.q1 == k[0] && c.q2 == k[1] && c.q3 == k[2]) return true;
    } else {
        for (const auto &k : sa) if (c.q1 == k[0] && c.q2 == k[1] && c.q3 == k[2]) return true;
    }
    return false;
}

}  // namespace pn2
