// sa_mlp.hip -- the grouped local MLP of a set-abstraction layer fused with its max-pool, on the
// matrix cores. gfx950.
//
// Replaces, for inference, reference utils/pointnet_util.py:44-50 + :117-127:
//     grouped_xyz = group_point(xyz, idx) - new_xyz ; new_points = concat(grouped_xyz, group_point(points, idx))
//     3 x [ tf_util.conv2d 1x1 + batch_norm (eps 1e-3) + ReLU ]  (tf_util.py:88-150, :512-531)
//     new_points = reduce_max(new_points, axis=[2])
// i.e. per centroid an (nsample x Cin) matrix through three dense layers and a column max. The
// reference (and a plain PyTorch port) materialises the (b, m, nsample, C) tensor of every layer in
// HBM: 128-640 MB per SA level in the segmentation configs. Here NOTHING between the idx tensor and
// the (b, m, C3) result leaves the CU: rows are gathered straight into MFMA operand registers, the
// activations of all three layers stay in registers, the max runs over lanes.
// Batch norm is folded into the weights by the caller (inference statistics): W' = W*s, b' = (b-mu)*s+beta.
//
// Arithmetic: fp32 results on the bf16 matrix pipe. Every fp32 operand is the exact sum of three bf16 values
// (nearest-even residuals: 8 + 8 + 8 significant bits), a product a.b is evaluated as the six bf16 x bf16 terms
// a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 (each exact in the fp32 accumulator; the three dropped terms are below 2^-24
// relative), i.e. six v_mfma_f32_32x32x16_bf16 per 16 contraction indices where the fp32-input
// v_mfma_f32_32x32x2_f32 needs eight at twice the cycles each: 2.7x fewer matrix-pipe cycles for the same
// accuracy (measured against an fp64 evaluation: the same error as an fp32 fma chain, tests/test_sa_mlp_gpu.py).
// The weights are split on the host when they are packed (three levels stored side by side, 6 bytes per weight);
// the activations are split in registers right after each layer's ReLU: 5.5 VALU operations per value
// (v_cvt_pk_bf16_f32, shift/mask back to fp32, subtract, twice) -- once per value and layer, amortised over the
// layer's output width.
//
// Formulation. One wave owns 32 samples (one centroid at nsample = 32) and computes the TRANSPOSED
// layer  H^T (channels x samples) = W^T (Cout x Cin) . X^T (Cin x samples):
//   MFMA "A" operand = a 32x16 block of W^T: lane l supplies W[k][n = 32t + (l & 31)] for eight k (slots j = 0..7
//                      of lane half l >> 5), chosen below;
//   MFMA "B" operand = a 16x32 block of X^T: lane l supplies X[sample = l & 31][k], the same eight k;
//   C/D: lane l, register v holds H^T[channel 8(v>>2) + 4(l>>5) + (v&3)][sample l & 31].
// The contraction index k may be visited in ANY order as long as A and B agree. Giving slot j of K16 step e
// (e = 0, 1) of input tile u the index k = 32u + 8((8e + j) >> 2) + 4(l >> 5) + (j & 3) makes the B operand of step e
// EXACTLY accumulator registers 8e .. 8e + 7 of the previous layer's tile u (after the split into levels): the
// activations never move between layers -- no LDS round trip, no shuffles. The weights are stored pre-permuted to
// match (pn2_sa_mlp3_pack), in LDS, one ds_read_b128 per lane, level and K16 step.
// The LAST layer swaps the two MFMA operands (their lane maps are the same: index l & 31, slots by l >> 5),
// which yields the untransposed H (samples x channels): a lane then holds 16 SAMPLES of one
// channel, so the max-pool is 15 lane-local v_max plus one exchange with lane l ^ 32, and bias + ReLU
// are applied once to the pooled value (x -> relu(x + b) is monotone, so max and it commute exactly).
// Pooling across lanes instead (the first version) cost 5 cross-lane steps for each of 64 registers.
// Eight waves (two per SIMD) share one copy of the weights in LDS.
#include "sa_mlp_common.h"

#include <stdlib.h>
#include <string.h>

namespace pn2 {

// One dense layer on the wave's 32 samples: out[t] = relu(bias + sum_u W^T[t][u] . in[u]), 12 bf16 MFMAs per
// (t, u) pair (two K16 steps of six). The weights of step i + 1 are read from LDS while the MFMAs of step i
// run (the compiler fence keeps the prefetch where it is written; left alone, hipcc hoists EVERY weight read
// of the layer to its top and runs out of registers). `ksteps` = K16 steps of an input tile that can be
// non-zero (2, except for a narrow first layer). LAST: operands swapped (see the header): out[t] holds raw
// sums H[sample][channel l & 31], no bias, no ReLU, max-accumulated over the centroid's 32-sample groups
// unless `first`.
template <int TOUT, int TIN, bool LAST>
__device__ __forceinline__ void mlp_layer(const float *wp, const float *bp, const ActSplit (&in)[TIN], f32x16 (&out)[TOUT],
                                          int lane, int h, int ksteps, bool first)
{
    const u32x4 *w4 = reinterpret_cast<const u32x4 *>(wp) + lane;
    u32x4 cur[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) cur[l] = w4[l * 64];
    f32x16 acc;
    constexpr int kSteps = TOUT * TIN * 2;
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
        const int pair = i >> 1, e = i & 1, t = pair / TIN, u = pair % TIN;
        u32x4 nxt[3];
        if (i + 1 < kSteps) {
#pragma unroll
            for (int l = 0; l < 3; ++l) nxt[l] = w4[((i + 1) * 3 + l) * 64];
        }
        if (u == 0 && e == 0) {
            if (LAST) {
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
            } else {
                acc = mlp_bias(bp, t, h);
            }
        }
        asm volatile("" ::: "memory");
        if (e < ksteps) acc = mma_x6<LAST>(cur, in[u].p[e], acc);
        if (u == TIN - 1 && e == 1) {
            if (!LAST) {
                out[t] = mlp_relu(acc);
            } else if (first) {
                out[t] = acc;
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) out[t][v] = vmax(out[t][v], acc[v]);
            }
        }
        if (i + 1 < kSteps) {
#pragma unroll
            for (int l = 0; l < 3; ++l) cur[l] = nxt[l];
        }
    }
}

// T1, T2, T3: output tiles (32 channels each) of the three layers; the input is one tile (Cin <= 32).
// SPAN: samples per centroid inside one 32-sample group (32, or 16 when nsample = 16).
// POOL: the reference's pooling modes (utils/pointnet_util.py:128-140): 0 max (the text above), 1 avg, 2 weighted_avg
// (weights exp(-5 |grouped_xyz|) normalised over the group), 3 max_and_avg (out = [avg (c3), max (c3)] per centroid). An average
// does not commute with bias + ReLU, so modes 1-3 apply them to every sample's raw sum and accumulate per lane: a lane holds 16
// samples of one channel, the sums of registers 0-7 and 8-15 are kept apart (SPAN = 16: two centroids), one exchange with
// lane l ^ 32 at the end. The weights of mode 2 come from the layer-1 operand (lanes 0-31 hold their sample's centred
// coordinates) through one ds_bpermute per register and 32-sample part.
template <int T1, int T2, int T3, int SPAN, int NT, int POOL = 0>
__global__ __launch_bounds__(NT) void sa_mlp3_kernel(int n, int m, int nsample, int cfeat, int c3, long long rows,
                                                              const float *__restrict__ xyz,
                                                              const float *__restrict__ new_xyz,
                                                              const float *__restrict__ points,
                                                              const int *__restrict__ idx,
                                                              const float *__restrict__ wpacked,
                                                              const float *__restrict__ bpacked, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *w1 = reinterpret_cast<float *>(smem);
    float *w2 = w1 + mlp_w_floats(T1, 1);
    float *w3 = w2 + mlp_w_floats(T2, T1);
    float *b1 = w3 + mlp_w_floats(T3, T2);
    float *b2 = b1 + mlp_b_floats(T1);
    float *b3 = b2 + mlp_b_floats(T2);
    {
        const size_t wf = mlp_w_floats(T1, 1) + mlp_w_floats(T2, T1) + mlp_w_floats(T3, T2);
        const size_t bf = mlp_b_floats(T1) + mlp_b_floats(T2) + mlp_b_floats(T3);
        const float4 *src = reinterpret_cast<const float4 *>(wpacked);
        float4 *dst = reinterpret_cast<float4 *>(w1);
        for (size_t i = threadIdx.x; i < wf / 4; i += NT) dst[i] = src[i];
        for (size_t i = threadIdx.x; i < bf; i += NT) b1[i] = bpacked[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, s = lane & 31;
    const int cin = 3 + cfeat;
    const int quartets1 = (cin + 7) / 8;                 // register quartet q covers channels 8q .. 8q+7
    const int ksteps1 = cin > 16 ? 2 : 1;                // K16 step e covers channels 16e .. 16e+15
    const int parts = SPAN == 32 ? nsample / 32 : 1;     // 32-sample groups per centroid
    const long long wave = (long long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (NT / 64);
    const long long groups = SPAN == 32 ? rows : (rows + 1) / 2;

    // The gather is software-pipelined over the work items (a 32-sample group of a centroid): the index
    // of item i + 1 is fetched when item i starts, its coordinates/features after item i's second layer
    // (the index has arrived by then), so both dependent global round trips hide under MFMA work.
    auto item_row = [&](long long g) -> long long {
        const long long r = SPAN == 32 ? g : g * 2 + (s >> 4);
        return r < rows ? r : rows - 1;                          // an odd tail at nsample = 16 recomputes the last row
    };
    auto load_index = [&](long long g, int part) -> int {
        const int sample = SPAN == 32 ? part * 32 + s : (s & 15);
        return idx[item_row(g) * nsample + sample];
    };
    // layer-1 operand: register v <- input channel mlp_chan(v, h) of this lane's sample. The loads only
    // (raw coordinates; `cen` = the centroid coordinate to subtract, 0 for feature channels): the
    // subtraction is done when the operand is consumed, one item later, so no wait sits in the MFMA stream
    auto load_x0 = [&](long long g, int p, f32x16 &cen) -> f32x16 {
        const long long row = item_row(g), cloud = row / m;
        const float *px = xyz + ((size_t)cloud * n + p) * 3;
        const float *pf = points ? points + ((size_t)cloud * n + p) * cfeat : nullptr;
        const float *c = new_xyz + row * 3;
        f32x16 x0;
        if (!points) {                                   // xyz only (every first SA level): three loads, no per-register tests
#pragma unroll
            for (int v = 0; v < 16; ++v) { x0[v] = 0.0f; cen[v] = 0.0f; }
            if (h == 0) {
                x0[0] = px[0]; x0[1] = px[1]; x0[2] = px[2];
                cen[0] = c[0]; cen[1] = c[1]; cen[2] = c[2];
            }
            return x0;
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int k = mlp_chan(v, h);
            float val = 0.0f, sub = 0.0f;
            if ((v >> 2) < quartets1) {
                if (k < 3) { val = px[k]; sub = c[k]; }
                else if (k < cin) val = pf[k - 3];
            }
            x0[v] = val;
            cen[v] = sub;
        }
        return x0;
    };

    long long g = wave;
    int part = 0;
    f32x16 x0, cen;
    if (g < groups) x0 = load_x0(g, load_index(g, 0), cen);
    f32x16 best[T3];
    float sa[T3], sb[T3], ma[T3], mb[T3], ea = 0.0f, eb = 0.0f;      // POOL != 0: sums / maxima of registers 0-7 and 8-15, weight sums
    while (g < groups) {
        // the item after this one
        const bool last_part = part + 1 == parts;
        const long long gn = last_part ? g + nwaves : g;
        const int partn = last_part ? 0 : part + 1;
        const bool more = gn < groups;
        int pn = 0;
        if (more) pn = load_index(gn, partn);
        ActSplit s0[1], s1[T1], s2[T2];
        float e_lane = 0.0f;
        {
            f32x16 in0;
#pragma unroll
            for (int v = 0; v < 4; ++v) in0[v] = __fsub_rn(x0[v], cen[v]);      // channels 0-2 live in registers 0-2 of lanes 0-31
#pragma unroll
            for (int v = 4; v < 16; ++v) in0[v] = x0[v];
            s0[0] = split_act(in0);
            if (POOL == 2)                                       // :133-134, tf.norm + exp(-5 d); lanes 0-31 hold the coordinates
                e_lane = expf(-5.0f * sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(in0[0], in0[0]), __fmul_rn(in0[1], in0[1])), __fmul_rn(in0[2], in0[2]))));
        }
        {
            f32x16 h1[T1];
            mlp_layer<T1, 1, false>(w1, b1, s0, h1, lane, h, ksteps1, true);
#pragma unroll
            for (int t = 0; t < T1; ++t) s1[t] = split_act(h1[t]);
        }
        {
            f32x16 h2[T2];
            mlp_layer<T2, T1, false>(w2, b2, s1, h2, lane, h, 2, true);
#pragma unroll
            for (int t = 0; t < T2; ++t) s2[t] = split_act(h2[t]);
        }
        if (more) x0 = load_x0(gn, pn, cen);
        mlp_layer<T3, T2, true>(w3, b3, s2, best, lane, h, 2, POOL != 0 || part == 0);
        if (POOL != 0) {
            // this part's 16 samples per lane: bias + ReLU per sample, then the (weighted) sums; max_and_avg keeps the raw maxima too
            float wv[16];
            if (POOL == 2) {
#pragma unroll
                for (int v = 0; v < 16; ++v) wv[v] = __shfl(e_lane, 8 * (v >> 2) + 4 * h + (v & 3));
            }
            if (part == 0) {
                ea = eb = 0.0f;
#pragma unroll
                for (int t = 0; t < T3; ++t) { sa[t] = sb[t] = 0.0f; ma[t] = mb[t] = -3.0e38f; }
            }
            if (POOL == 2) {
#pragma unroll
                for (int v = 0; v < 8; ++v) { ea += wv[v]; eb += wv[8 + v]; }
            }
#pragma unroll
            for (int t = 0; t < T3; ++t) {
                const float bias = b3_at(b3, 32 * t + s);
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    float r = fmaxf(__fadd_rn(best[t][v], bias), 0.0f);
                    if (POOL == 2) r *= wv[v];
                    if (v < 8) sa[t] += r; else sb[t] += r;
                    if (POOL == 3) { if (v < 8) ma[t] = fmaxf(ma[t], best[t][v]); else mb[t] = fmaxf(mb[t], best[t][v]); }
                }
            }
            if (!last_part) { part = partn; continue; }
            const long long row = item_row(g);
            const int oc = POOL == 3 ? 2 * c3 : c3;                 // floats per output row
            float d0 = (float)nsample, d1 = (float)nsample;
            if (POOL == 2) {
                if (SPAN == 32) { d0 = ea + eb; d0 += __shfl_xor(d0, 32); }
                else { d0 = ea + __shfl_xor(ea, 32); d1 = eb + __shfl_xor(eb, 32); }
            }
#pragma unroll
            for (int t = 0; t < T3; ++t) {
                const int ch = 32 * t + s;
                const float bias = b3_at(b3, ch);
                if (SPAN == 32) {
                    float sum = sa[t] + sb[t], mx = fmaxf(ma[t], mb[t]);
                    sum += __shfl_xor(sum, 32);
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    if (h == 0 && ch < c3) {
                        out[row * oc + ch] = sum / d0;
                        if (POOL == 3) out[row * oc + c3 + ch] = fmaxf(__fadd_rn(mx, bias), 0.0f);
                    }
                } else {
                    const float s0_ = sa[t] + __shfl_xor(sa[t], 32), s1_ = sb[t] + __shfl_xor(sb[t], 32);
                    const float m0 = fmaxf(ma[t], __shfl_xor(ma[t], 32)), m1 = fmaxf(mb[t], __shfl_xor(mb[t], 32));
                    if (h == 0 && ch < c3) {
                        out[(2 * g) * oc + ch] = s0_ / d0;
                        if (POOL == 3) out[(2 * g) * oc + c3 + ch] = fmaxf(__fadd_rn(m0, bias), 0.0f);
                        if (2 * g + 1 < rows) {
                            out[(2 * g + 1) * oc + ch] = s1_ / d1;
                            if (POOL == 3) out[(2 * g + 1) * oc + c3 + ch] = fmaxf(__fadd_rn(m1, bias), 0.0f);
                        }
                    }
                }
            }
            g = gn;
            part = 0;
            continue;
        }
        if (!last_part) { part = partn; continue; }
        const long long row = item_row(g);
        // Pool: lane l holds channel 32t + (l & 31) for the samples 8(v >> 2) + 4(l >> 5) + (v & 3), v = 0..15
        // (SPAN = 16: registers 0-7 are the first centroid's 16 samples, 8-15 the second's); bias and ReLU
        // on the pooled value; one coalesced 128-byte store per tile and centroid.
#pragma unroll
        for (int t = 0; t < T3; ++t) {
            const int ch = 32 * t + s;
            const float bias = b3_at(b3, ch);
            if (SPAN == 32) {
                float mx = best[t][0];
#pragma unroll
                for (int v = 1; v < 16; ++v) mx = fmaxf(mx, best[t][v]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (h == 0 && ch < c3) out[row * c3 + ch] = fmaxf(__fadd_rn(mx, bias), 0.0f);
            } else {
                float m0 = best[t][0], m1 = best[t][8];
#pragma unroll
                for (int v = 1; v < 8; ++v) { m0 = fmaxf(m0, best[t][v]); m1 = fmaxf(m1, best[t][8 + v]); }
                m0 = fmaxf(m0, __shfl_xor(m0, 32));
                m1 = fmaxf(m1, __shfl_xor(m1, 32));
                if (h == 0 && ch < c3) {
                    out[(2 * g) * c3 + ch] = fmaxf(__fadd_rn(m0, bias), 0.0f);
                    if (2 * g + 1 < rows) out[(2 * g + 1) * c3 + ch] = fmaxf(__fadd_rn(m1, bias), 0.0f);
                }
            }
        }
        g = gn;
        part = 0;
    }
}

// ---- two items per wave -------------------------------------------------------------------------------------------
// What the kernel above leaves on the table (profiles/r04/ubench_mfma_valu.txt): on gfx950 a wave's vector instructions
// hide completely under its OWN queued MFMAs (four per v_mfma_f32_32x32x16_bf16 cost nothing) as long as they do not
// depend on them, while the vector work of the OTHER wave of the SIMD hides only by half. Within one 32-sample item the
// level split of a layer's output depends on the layer's last MFMAs and feeds the next layer's first ones, so with one
// item per wave the kernel runs at MFMA time + vector time (81 + 54 us at the metric shape). Here a wave carries TWO
// items, A and B, half a phase apart: every layer of one item ("host": its weight reads and MFMAs, step by step) is
// followed, step by step, by pieces of the other item's vector work ("guest": ReLU + level split of the previous layer's
// accumulators, the input subtraction, the pooling tail),
//     L1(A) | input split B      L1(B) | split h1 A      L2(A) | split h1 B
//     L2(B) | split h2 A         L3(A) | split h2 B      L3(B) | pool A, input split of A's next item
// so the guest's instructions are independent of the MFMAs in flight. One wave per SIMD (256 threads, up to 512 registers
// with the accumulation registers as overflow) carries the state of both items. A sched_barrier after every step keeps
// hipcc from regrouping the two streams. Arithmetic and results are bit-identical to the kernel above. Levels of one
// 32-sample group per centroid (nsample 32 or 16) and at most 16 input channels; the rest keeps one item per wave.
template <bool RELU>
__device__ __forceinline__ void split_piece(const f32x16 &x, int e, int d, ActSplit &s)
{
    float a = x[8 * e + 2 * d], b = x[8 * e + 2 * d + 1];
    // a piece belongs to the step of the host layer it is written in: the empty volatile statement is ordered with the steps'
    // sched_barriers, and the piece's arithmetic depends on it (left alone, instruction selection emits a guest's whole split
    // in front of the host layer, and the barriers then keep it there)
    asm volatile("" : "+v"(a), "+v"(b));
    if (RELU) { a = vmax(a, 0.0f); b = vmax(b, 0.0f); }
    const unsigned int p1 = pack_bf16(a, b);
    const float ra = __fsub_rn(a, __uint_as_float(p1 << 16)), rb = __fsub_rn(b, __uint_as_float(p1 & 0xffff0000u));
    const unsigned int p2 = pack_bf16(ra, rb);
    const float sa = __fsub_rn(ra, __uint_as_float(p2 << 16)), sb = __fsub_rn(rb, __uint_as_float(p2 & 0xffff0000u));
    unsigned int q1 = p1, q2 = p2, q3 = pack_bf16(sa, sb);
    asm volatile("" : "+v"(q1), "+v"(q2), "+v"(q3));      // ... and its results exist when the step ends (nothing sinks to the consumer)
    s.p[e][0][d] = q1;
    s.p[e][1][d] = q2;
    s.p[e][2][d] = q3;
}

// pieces [lo, hi) of "ReLU + level split of the T tiles x" (eight pieces per tile); lo, hi are literals after unrolling
template <int T>
__device__ __forceinline__ void split_pieces(const f32x16 (&x)[T], ActSplit (&s)[T], int lo, int hi)
{
#pragma unroll
    for (int c = 0; c < T * 8; ++c)
        if (c >= lo && c < hi) split_piece<true>(x[c >> 3], (c >> 2) & 1, c & 3, s[c >> 3]);
}

// mlp_layer with a guest: hook(i) runs after the MFMAs of step i (i = 2 (t TIN + u) + e), and nothing is scheduled across
// the end of a step. The ReLU of a hidden layer is left to the consumer's split (out[t] = bias + sums).
template <int TOUT, int TIN, bool LAST, int KSTEPS, typename Hook>
__device__ __forceinline__ void mlp_layer_host(const float *wp, const float *bp, const ActSplit (&in)[TIN], f32x16 (&out)[TOUT],
                                               int lane, int h, Hook hook)
{
    const u32x4 *w4 = reinterpret_cast<const u32x4 *>(wp) + lane;
    // the weights of a step are read from LDS PN2_PAIR_PREFETCH steps ahead (three buffers in rotation)
#ifndef PN2_PAIR_PREFETCH
#define PN2_PAIR_PREFETCH 1
#endif
    constexpr int PF = PN2_PAIR_PREFETCH;
    u32x4 wb[PF + 1][3];
    constexpr int kSteps = TOUT * TIN * 2;
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (j < kSteps) {
#pragma unroll
            for (int l = 0; l < 3; ++l) wb[j][l] = w4[(j * 3 + l) * 64];
        }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < kSteps; ++i) {
        const int pair = i >> 1, e = i & 1, t = pair / TIN, u = pair % TIN;
        if (i + PF < kSteps) {
#pragma unroll
            for (int l = 0; l < 3; ++l) wb[(i + PF) % (PF + 1)][l] = w4[((i + PF) * 3 + l) * 64];
        }
        if (u == 0 && e == 0) {
            if (LAST) {
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
            } else {
                acc = mlp_bias(bp, t, h);
            }
        }
        if (e < KSTEPS) acc = mma_x6<LAST>(wb[i % (PF + 1)], in[u].p[e], acc);
        if (u == TIN - 1 && e == 1) out[t] = acc;
        hook(i);
#ifndef PN2_PAIR_NOBARRIER                 /* lab switch (scripts/build_mlp_labs.sh): what the step barriers are worth */
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

template <int T1, int T2, int T3, int SPAN, int NT>
__global__ __launch_bounds__(NT) void sa_mlp3_pair_kernel(int n, int m, int nsample, int cfeat, int c3, long long rows,
                                                           const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                           const float *__restrict__ points, const int *__restrict__ idx,
                                                           const float *__restrict__ wpacked, const float *__restrict__ bpacked,
                                                           float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *w1 = reinterpret_cast<float *>(smem);
    float *w2 = w1 + mlp_w_floats(T1, 1);
    float *w3 = w2 + mlp_w_floats(T2, T1);
    float *b1 = w3 + mlp_w_floats(T3, T2);
    float *b2 = b1 + mlp_b_floats(T1);
    float *b3 = b2 + mlp_b_floats(T2);
    {
        const size_t wf = mlp_w_floats(T1, 1) + mlp_w_floats(T2, T1) + mlp_w_floats(T3, T2);
        const size_t bf = mlp_b_floats(T1) + mlp_b_floats(T2) + mlp_b_floats(T3);
        const float4 *src = reinterpret_cast<const float4 *>(wpacked);
        float4 *dst = reinterpret_cast<float4 *>(w1);
        for (size_t i = threadIdx.x; i < wf / 4; i += NT) dst[i] = src[i];
        for (size_t i = threadIdx.x; i < bf; i += NT) b1[i] = bpacked[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, s = lane & 31;
    const int cin = 3 + cfeat;                           // <= 16: one K16 step in layer 1, input registers 0 .. 7
    const long long wave = (long long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (NT / 64);
    const long long groups = SPAN == 32 ? rows : (rows + 1) / 2;

    auto item_row = [&](long long g) __attribute__((always_inline)) -> long long {
        const long long r = SPAN == 32 ? g : g * 2 + (s >> 4);
        return r < rows ? r : rows - 1;
    };
    auto load_index = [&](long long g) __attribute__((always_inline)) -> int {
        const int sample = SPAN == 32 ? s : (s & 15);
        return idx[item_row(g) * nsample + sample];
    };
    // layer-1 operand registers 0 .. 7 of this lane's sample (channels mlp_chan(v, h)) and what to subtract from them
    auto load_x0 = [&](long long g, int p, float (&x0)[8], float (&cen)[8]) __attribute__((always_inline)) {
        const long long row = item_row(g), cloud = row / m;
        const float *px = xyz + ((size_t)cloud * n + p) * 3;
        const float *c = new_xyz + row * 3;
        if (!points) {                                   // xyz only: channels 0-2 live in registers 0-2 of lanes 0-31
            const float keep = h == 0 ? 1.0f : 0.0f;
#pragma unroll
            for (int v = 0; v < 8; ++v) { x0[v] = 0.0f; cen[v] = 0.0f; }
#pragma unroll
            for (int v = 0; v < 3; ++v) { x0[v] = px[v] * keep; cen[v] = c[v] * keep; }
            return;
        }
        const float *pf = points + ((size_t)cloud * n + p) * cfeat;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int k = mlp_chan(v, h);
            float val = 0.0f, sub = 0.0f;
            if (k < 3) { val = px[k]; sub = c[k]; }
            else if (k < cin) val = pf[k - 3];
            x0[v] = val;
            cen[v] = sub;
        }
    };
    // the input split: (x0 - centroid) -> the three levels of K16 step 0
    auto input_split = [&](const float (&x0)[8], const float (&cen)[8], ActSplit &s0) __attribute__((always_inline)) {
        f32x16 in0;
#pragma unroll
        for (int v = 0; v < 4; ++v) in0[v] = __fsub_rn(x0[v], cen[v]);
#pragma unroll
        for (int v = 4; v < 8; ++v) in0[v] = x0[v];
#pragma unroll
        for (int v = 8; v < 16; ++v) in0[v] = 0.0f;
#pragma unroll
        for (int d = 0; d < 4; ++d) split_piece<false>(in0, 0, d, s0);
#pragma unroll
        for (int l = 0; l < 3; ++l) s0.p[1][l] = u32x4{0u, 0u, 0u, 0u};
    };
    // pool + bias + ReLU + store of output tile t of item g (see sa_mlp3_kernel)
    auto pool_tile = [&](const f32x16 (&best)[T3], int t, long long g) __attribute__((always_inline)) {
        const long long row = item_row(g);
        const int ch = 32 * t + s;
        const float bias = b3_at(b3, ch);
        const bool live = g < groups && h == 0 && ch < c3;
        if (SPAN == 32) {
            float mx = best[t][0];
            asm volatile("" : "+v"(mx));                 // (pinned to its step, see split_piece)
#pragma unroll
            for (int v = 1; v < 16; ++v) mx = fmaxf(mx, best[t][v]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (live) out[row * c3 + ch] = fmaxf(__fadd_rn(mx, bias), 0.0f);
        } else {
            float m0 = best[t][0], m1 = best[t][8];
            asm volatile("" : "+v"(m0), "+v"(m1));
#pragma unroll
            for (int v = 1; v < 8; ++v) { m0 = fmaxf(m0, best[t][v]); m1 = fmaxf(m1, best[t][8 + v]); }
            m0 = fmaxf(m0, __shfl_xor(m0, 32));
            m1 = fmaxf(m1, __shfl_xor(m1, 32));
            if (live) {
                out[(2 * g) * c3 + ch] = fmaxf(__fadd_rn(m0, bias), 0.0f);
                if (2 * g + 1 < rows) out[(2 * g + 1) * c3 + ch] = fmaxf(__fadd_rn(m1, bias), 0.0f);
            }
        }
    };
    auto clampg = [&](long long g) __attribute__((always_inline)) -> long long { return g < groups ? g : groups - 1; };

    // stream A: items wave, wave + 2 nwaves, ...; stream B: wave + nwaves, wave + 3 nwaves, ... (an item beyond the end
    // recomputes the last one and stores nothing)
    long long gA = wave, gB = wave + nwaves;
    if (gA >= groups) return;
    float xA[8], cA[8], xB[8], cB[8];
    load_x0(clampg(gA), load_index(clampg(gA)), xA, cA);
    load_x0(clampg(gB), load_index(clampg(gB)), xB, cB);
    ActSplit s0A[1], s0B[1];
    input_split(xA, cA, s0A[0]);
    while (gA < groups) {
        const long long gAn = clampg(gA + 2 * nwaves), gBn = clampg(gB + 2 * nwaves);
        const int pnA = load_index(gAn);                 // x, c of A are consumed: its next item's index first, the row later
        f32x16 h1A[T1], h1B[T1], h2A[T2], h2B[T2], accA[T3], accB[T3];
        ActSplit s1A[T1], s1B[T1], s2A[T2], s2B[T2];
        // L1(A) | input split of B
        mlp_layer_host<T1, 1, false, 1>(w1, b1, s0A, h1A, lane, h, [&](int i) __attribute__((always_inline)) {
            if (i == 0) input_split(xB, cB, s0B[0]);
        });
        const int pnB = load_index(gBn);
        // L1(B) | ReLU + split of A's first layer
        mlp_layer_host<T1, 1, false, 1>(w1, b1, s0B, h1B, lane, h, [&](int i) __attribute__((always_inline)) {
            constexpr int NP = T1 * 8, NS = T1 * 2;
            split_pieces<T1>(h1A, s1A, i * NP / NS, (i + 1) * NP / NS);
        });
        // L2(A) | ReLU + split of B's first layer
        mlp_layer_host<T2, T1, false, 2>(w2, b2, s1A, h2A, lane, h, [&](int i) __attribute__((always_inline)) {
            constexpr int NP = T1 * 8, NS = T2 * T1 * 2;
            split_pieces<T1>(h1B, s1B, i * NP / NS, (i + 1) * NP / NS);
        });
        load_x0(gAn, pnA, xA, cA);                       // in flight under the next layers
        // L2(B) | ReLU + split of A's second layer
        mlp_layer_host<T2, T1, false, 2>(w2, b2, s1B, h2B, lane, h, [&](int i) __attribute__((always_inline)) {
            constexpr int NP = T2 * 8, NS = T2 * T1 * 2;
            split_pieces<T2>(h2A, s2A, i * NP / NS, (i + 1) * NP / NS);
        });
        load_x0(gBn, pnB, xB, cB);
        // L3(A) | ReLU + split of B's second layer
        mlp_layer_host<T3, T2, true, 2>(w3, b3, s2A, accA, lane, h, [&](int i) __attribute__((always_inline)) {
            constexpr int NP = T2 * 8, NS = T3 * T2 * 2;
            split_pieces<T2>(h2B, s2B, i * NP / NS, (i + 1) * NP / NS);
        });
        // L3(B) | pool of A (its MFMAs were queued before this layer's), then the input split of A's next item
        mlp_layer_host<T3, T2, true, 2>(w3, b3, s2B, accB, lane, h, [&](int i) __attribute__((always_inline)) {
            constexpr int NS = T3 * T2 * 2, first = NS / 4;          // leave the first steps to A's last MFMAs
#pragma unroll
            for (int t = 0; t < T3; ++t)
                if (i == first + t * ((NS - first - 1) / T3)) pool_tile(accA, t, gA);
            if (i == NS - 1) input_split(xA, cA, s0A[0]);
        });
        // pool of B: nothing left to hide it under
#pragma unroll
        for (int t = 0; t < T3; ++t) pool_tile(accB, t, gB);
        gA += 2 * nwaves;
        gB += 2 * nwaves;
    }
}

// ---- host side: bf16 levels of the weights ------------------------------------------------------------------
static unsigned short bf16_nearest_even(float f)
{
    unsigned int u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);      // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

static float bf16_value(unsigned short b)
{
    const unsigned int u = (unsigned int)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void mlp_split_weight(float w, unsigned short out[3])
{
    out[0] = bf16_nearest_even(w);
    const float r1 = w - bf16_value(out[0]);              // exact: the residual of a nearest rounding fits in fp32
    out[1] = bf16_nearest_even(r1);
    const float r2 = r1 - bf16_value(out[1]);
    out[2] = bf16_nearest_even(r2);
}

// value for K16 step e, level, lane l, slot j = level of W[krow(32u + mlp_chan(8e + j, l >> 5))][32t + (l & 31)]
float *mlp_pack_pair_x6(float *wp, const float *w, int kin, int nout, int t, int u, const int *krow)
{
    unsigned short *o = reinterpret_cast<unsigned short *>(wp);
    for (int e = 0; e < 2; ++e)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * u + mlp_chan(8 * e + j, lane >> 5), nn = 32 * t + (lane & 31);
                const float val = (k < kin && nn < nout) ? w[(size_t)(krow ? krow[k] : k) * nout + nn] : 0.0f;
                unsigned short lv[3];
                mlp_split_weight(val, lv);
                for (int level = 0; level < 3; ++level) o[(((e * 3 + level) * 64 + lane) * 8) + j] = lv[level];
            }
    return wp + kPairWords;
}

struct MlpConfig { int t1, t2, t3; };

// smallest instantiated tile configuration that covers (c1, c2, c3); padded channels carry zero
// weights and zero bias, cost MFMA time and never reach memory
static bool mlp_pick(int c1, int c2, int c3, MlpConfig &cfg)
{
    static const MlpConfig kConfigs[] = {{1, 1, 2}, {2, 2, 4}, {2, 3, 4}};
    for (const MlpConfig &c : kConfigs)
        if (c1 <= 32 * c.t1 && c2 <= 32 * c.t2 && c3 <= 32 * c.t3) { cfg = c; return true; }
    return false;
}

static size_t mlp_total_w(const MlpConfig &c) { return mlp_w_floats(c.t1, 1) + mlp_w_floats(c.t2, c.t1) + mlp_w_floats(c.t3, c.t2); }
static size_t mlp_total_b(const MlpConfig &c) { return mlp_b_floats(c.t1) + mlp_b_floats(c.t2) + mlp_b_floats(c.t3); }

template <int T1, int T2, int T3>
static int launch_mlp(int b, int n, int m, int nsample, int cfeat, int c3, const float *xyz, const float *new_xyz,
                      const float *points, const int *idx, const float *wp, const float *bp, float *out, hipStream_t st, int variant,
                      int pooling = 0)
{
    const MlpConfig cfg = {T1, T2, T3};
    const size_t lds = sizeof(float) * (mlp_total_w(cfg) + mlp_total_b(cfg));
    if (lds > (size_t)kMlpMaxLds) return PN2_E_TOO_LARGE;
    const long long rows = (long long)b * m;
    const bool half = nsample == 16;
    const long long groups = half ? (rows + 1) / 2 : rows;
    // two items per wave (sa_mlp3_pair_kernel): one 32-sample group per centroid, at most 16 input channels
    // (variant: 0 by the size rule below, 1 never, 2 / 3 wherever the kernel covers the shape: four / eight waves per workgroup)
    const bool pair_ok = (nsample == 32 || nsample == 16) && 3 + cfeat <= 16 && T2 <= 2;      // (T2 = 3: two items' state spills)
    // size rule (measured, profiles/r04/mlp_two_items.txt): eight waves x two items is 6-8 % faster than eight waves x one from
    // ~8k items (metric shape 139 -> 131 us, cls_ssg SA1 36 -> 33), neutral below; four waves x two items is no faster than
    // one item per wave (142 us): kept for tests / A-B only
    if (variant == 0 && pair_ok && groups >= 8192) variant = 3;
    if (pooling != 0) {
        // avg / weighted_avg / max_and_avg (pointnet_util.py:130-140): one item per wave, eight waves (sa_mlp3_kernel<.., POOL>)
        constexpr int NT = 512;
        long long blocks = (groups + NT / 64 - 1) / (NT / 64);
        if (blocks > 256) blocks = 256;
#define PN2_POOL_CASE(P)                                                                                                      \
        if (pooling == P) {                                                                                                   \
            auto kern = half ? sa_mlp3_kernel<T1, T2, T3, 16, NT, P> : sa_mlp3_kernel<T1, T2, T3, 32, NT, P>;                 \
            if (int rc = allow_dynamic_lds(kern, lds)) return rc;                                                             \
            return launch(kern, dim3((unsigned)blocks), dim3(NT), lds, st, n, m, nsample, cfeat, c3, rows, xyz, new_xyz, points, \
                          idx, wp, bp, out);                                                                                  \
        }
        PN2_POOL_CASE(1) PN2_POOL_CASE(2) PN2_POOL_CASE(3)
#undef PN2_POOL_CASE
        return PN2_E_ARG;
    }
    if (pair_ok && variant >= 2) {
        if constexpr (T2 <= 2) {
          if (variant == 2) {                                           // four waves x two items
            long long blocks = (groups + 7) / 8;
            if (blocks > 256) blocks = 256;
            auto kern = half ? sa_mlp3_pair_kernel<T1, T2, T3, 16, 256> : sa_mlp3_pair_kernel<T1, T2, T3, 32, 256>;
            if (int rc = allow_dynamic_lds(kern, lds)) return rc;
            return launch(kern, dim3((unsigned)blocks), dim3(256), lds, st, n, m, nsample, cfeat, c3, rows, xyz, new_xyz, points, idx,
                          wp, bp, out);
          }
          // eight waves x two items (fits 256 registers)
            long long blocks = (groups + 15) / 16;
            if (blocks > 256) blocks = 256;
            auto kern = half ? sa_mlp3_pair_kernel<T1, T2, T3, 16, 512> : sa_mlp3_pair_kernel<T1, T2, T3, 32, 512>;
            if (int rc = allow_dynamic_lds(kern, lds)) return rc;
            return launch(kern, dim3((unsigned)blocks), dim3(512), lds, st, n, m, nsample, cfeat, c3, rows, xyz, new_xyz, points, idx,
                          wp, bp, out);
        }
    }
    // eight waves share one copy of the weights (two per SIMD: one wave's operand splitting overlaps the other's
    // MFMAs) when a wave fits in 256 registers; persistent: every workgroup stages the weights once
    constexpr int NT = 512;
    long long blocks = (groups + NT / 64 - 1) / (NT / 64);
    long long cap = 256;
    if (blocks > cap) blocks = cap;
    auto kern = half ? sa_mlp3_kernel<T1, T2, T3, 16, NT> : sa_mlp3_kernel<T1, T2, T3, 32, NT>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    if (int rc = launch(kern, dim3((unsigned)blocks), dim3(NT), lds, st, n, m, nsample, cfeat, c3, rows, xyz, new_xyz,
                       points, idx, wp, bp, out)) return rc;
    return PN2_OK;
}

// Which kernel runs a stack: the resident one when the input is narrow and the weights fit in LDS, the
// streamed one (sa_mlp_stream.hip) up to widths (128,128,256), else the cooperative one (coop_mlp.hip: wide
// stacks such as (256,256,512) and the group_all level's (256,512,1024); any nsample, masked).
// kind: 0 resident, 1 streamed, 2 cooperative.
static bool mlp_choose(int cin, int c1, int c2, int c3, int nsample, int &kind, MlpConfig &rc, MlpStreamConfig &sc,
                       MlpCoopConfig &cc)
{
    const bool whole = nsample == 16 || (nsample > 0 && nsample % 32 == 0);
    if (whole && cin <= 32 && mlp_pick(c1, c2, c3, rc) &&
        sizeof(float) * (mlp_total_w(rc) + mlp_total_b(rc)) <= (size_t)kMlpMaxLds) {
        kind = 0;
        return true;
    }
    if (whole && nsample != 16 && mlp_stream_pick(cin, c1, c2, c3, sc)) {
        kind = 1;
        return true;
    }
    const int widths[3] = {c1, c2, c3};
    if (nsample > 0 && mlp_coop_pick(cin, 3, widths, cc) && mlp_coop_has_kernel(cc, 0)) {
        kind = 2;
        return true;
    }
    return false;
}

}  // namespace pn2

extern "C" int pn2_sa_mlp3_config(int cin, int c1, int c2, int c3, int nsample, int *info4, long long *w_floats,
                                  long long *b_floats)
{
    using namespace pn2;
    if (cin < 3 || c1 <= 0 || c2 <= 0 || c3 <= 0) return PN2_E_ARG;
    if (nsample <= 0) return PN2_E_ARG;
    int kind;
    MlpConfig rc;
    MlpStreamConfig sc;
    MlpCoopConfig cc;
    if (!mlp_choose(cin, c1, c2, c3, nsample, kind, rc, sc, cc)) return PN2_E_TOO_LARGE;
    if (info4) {
        info4[0] = kind;
        info4[1] = kind == 2 ? 4 * cc.q1 : kind ? sc.t1 : rc.t1;
        info4[2] = kind == 2 ? 4 * cc.q2 : kind ? sc.t2 : rc.t2;
        info4[3] = kind == 2 ? 4 * cc.q3 : kind ? sc.t3 : rc.t3;
    }
    if (w_floats) *w_floats = (long long)(kind == 2 ? mlp_coop_w_floats(cc) : kind ? mlp_stream_w_floats(sc) : mlp_total_w(rc));
    if (b_floats) *b_floats = (long long)(kind == 2 ? mlp_coop_b_floats(cc) : kind ? mlp_stream_b_floats(sc) : mlp_total_b(rc));
    return PN2_OK;
}

// Host-side packing (plain C loops, no device work): w_i is (cin_i, cout_i) row-major -- the layout of
// the reference's conv kernel [1,1,cin,cout] (tf_util.py:113-117) -- with batch norm already folded in.
// xyz_first: rows of w1 are [xyz (3), features] (pointnet_util.py:50, single-scale modules) or
// [features, xyz (3)] (:184, the MSG module).
extern "C" int pn2_sa_mlp3_pack(int cin, int c1, int c2, int c3, int nsample, int xyz_first, const float *w1,
                                const float *bias1, const float *w2, const float *bias2, const float *w3,
                                const float *bias3, float *wpacked, float *bpacked)
{
    using namespace pn2;
    int kind;
    MlpConfig cfg;
    MlpStreamConfig sc;
    MlpCoopConfig cc;
    if (cin < 3 || c1 <= 0 || c2 <= 0 || c3 <= 0 || !mlp_choose(cin, c1, c2, c3, nsample, kind, cfg, sc, cc)) return PN2_E_TOO_LARGE;
    if (!w1 || !w2 || !w3 || !bias1 || !bias2 || !bias3 || !wpacked || !bpacked) return PN2_E_NULL;
    const float *ws[3] = {w1, w2, w3}, *bs[3] = {bias1, bias2, bias3};
    if (kind == 1) {
        mlp_stream_pack(sc, cin, c1, c2, c3, xyz_first, ws, bs, wpacked, bpacked);
        return PN2_OK;
    }
    if (kind == 2) {
        // kernel channel order of layer 1: [features, xyz]; caller's weight rows: [xyz, features] when xyz_first
        const int cf = cin - 3, widths[3] = {c1, c2, c3};
        int *krow = (int *)malloc(sizeof(int) * (size_t)cin);
        for (int k = 0; k < cin; ++k) krow[k] = xyz_first ? (k < cf ? 3 + k : k - cf) : k;
        mlp_coop_pack(cc, cin, 3, widths, krow, ws, bs, wpacked, bpacked, true);
        free(krow);
        return PN2_OK;
    }
    const int cfeat = cin - 3;
    const int kin[3] = {cin, c1, c2}, nout[3] = {c1, c2, c3};
    const int tin[3] = {1, cfg.t1, cfg.t2}, tout[3] = {cfg.t1, cfg.t2, cfg.t3};
    float *wp = wpacked, *bp = bpacked;
    for (int L = 0; L < 3; ++L) {
        // kernel channel order of layer 1: [xyz, features]
        int krow[32];
        for (int k = 0; k < 32; ++k) krow[k] = (L == 0 && !xyz_first && k < cin) ? (k < 3 ? cfeat + k : k - 3) : k;
        for (int t = 0; t < tout[L]; ++t)
            for (int u = 0; u < tin[L]; ++u) wp = mlp_pack_pair_x6(wp, ws[L], kin[L], nout[L], t, u, L == 0 ? krow : nullptr);
        for (int t = 0; t < tout[L]; ++t)
            for (int hh = 0; hh < 2; ++hh)
                for (int v = 0; v < 16; ++v) {
                    const int ch = 32 * t + mlp_chan(v, hh);
                    *bp++ = ch < nout[L] ? bs[L][ch] : 0.0f;
                }
    }
    return PN2_OK;
}

// Scratch pn2_sa_mlp3_maxpool needs for this call: the streamed kernel keeps the per-point part of layer 1 there
// (b * n rows of the padded first width); stacks with a last layer wider than 512 (the group_all level) keep the
// second layer's output there, in operand form, for the GEMM that runs the last layer; 0 otherwise.
extern "C" long long pn2_sa_mlp3_ws_bytes(int b, int n, int m, int cin, int c1, int c2, int c3, int nsample)
{
    using namespace pn2;
    int kind;
    MlpConfig rc;
    MlpStreamConfig sc;
    MlpCoopConfig cc;
    if (b <= 0 || n <= 0 || m <= 0 || cin < 3 || c1 <= 0 || c2 <= 0 || c3 <= 0 || nsample <= 0) return 0;
    if (!mlp_choose(cin, c1, c2, c3, nsample, kind, rc, sc, cc)) return 0;
    if (kind == 1) return (long long)mlp_stream_ws_bytes(sc, (long long)b * n);
    if (kind == 2) return (long long)mlp_coop_ws_bytes(cc, 0, (long long)b * m, nsample);
    return 0;
}

extern "C" int pn2_sa_mlp3_maxpool(int b, int n, int m, int nsample, int cfeat, const float *xyz, const float *new_xyz,
                                   const float *points, const int *idx, int c1, int c2, int c3, const float *wpacked,
                                   const float *bpacked, float *out, void *ws, void *stream)
{
    return pn2_sa_mlp3_maxpool_ex(b, n, m, nsample, cfeat, xyz, new_xyz, points, idx, c1, c2, c3, wpacked, bpacked, out, ws, 0, stream);
}

// pn2_sa_mlp3_maxpool with the reference's other pooling modes (utils/pointnet_util.py:128-140). pooling: 0 max, 1 avg,
// 2 weighted_avg, 3 max_and_avg (out is (b, m, 2 c3): [avg, max], the reference's concat order :142). Modes 1-3 exist for the
// stacks the RESIDENT and the STREAMED kernel cover (pn2_sa_mlp3_config kind 0 / 1: widths up to (128, 128, 256), nsample 16 or
// a multiple of 32; ws as for pn2_sa_mlp3_maxpool) -- the cooperative kernel's shapes return PN2_E_TOO_LARGE and the caller
// evaluates the level layer by layer.
extern "C" int pn2_sa_mlp3_pool_supported(int cin, int c1, int c2, int c3, int nsample, int pooling)
{
    using namespace pn2;
    if (pooling < 0 || pooling > 3 || cin < 3 || c1 <= 0 || c2 <= 0 || c3 <= 0 || nsample <= 0) return 0;
    int kind;
    MlpConfig rc;
    MlpStreamConfig sc;
    MlpCoopConfig cc;
    if (!mlp_choose(cin, c1, c2, c3, nsample, kind, rc, sc, cc)) return 0;
    return pooling == 0 || kind != 2;
}

extern "C" int pn2_sa_mlp3_pool(int b, int n, int m, int nsample, int cfeat, const float *xyz, const float *new_xyz,
                                const float *points, const int *idx, int c1, int c2, int c3, const float *wpacked,
                                const float *bpacked, int pooling, float *out, void *ws, void *stream)
{
    using namespace pn2;
    if (pooling < 0 || pooling > 3) return PN2_E_ARG;
    if (pooling == 0)
        return pn2_sa_mlp3_maxpool_ex(b, n, m, nsample, cfeat, xyz, new_xyz, points, idx, c1, c2, c3, wpacked, bpacked, out, ws, 0, stream);
    if (b < 0 || n <= 0 || m < 0 || cfeat < 0) return PN2_E_SHAPE;
    if (nsample <= 0) return PN2_E_ARG;
    if (b == 0 || m == 0) return PN2_OK;
    if (!xyz || !new_xyz || !idx || !wpacked || !bpacked || !out || (cfeat > 0 && !points)) return PN2_E_NULL;
    int kind;
    MlpConfig cfg;
    MlpStreamConfig sc;
    MlpCoopConfig cc;
    if (!mlp_choose(3 + cfeat, c1, c2, c3, nsample, kind, cfg, sc, cc) || kind == 2) return PN2_E_TOO_LARGE;
    hipStream_t st = as_stream(stream);
    const float *pts = cfeat > 0 ? points : nullptr;
    if (kind == 1)
        return mlp_stream_launch(sc, b, n, m, nsample, cfeat, c3, xyz, new_xyz, pts ? pts : xyz, idx, wpacked, bpacked, out, ws, st, pooling);
#define PN2_MLP_CASE(A, B, C) \
    if (cfg.t1 == A && cfg.t2 == B && cfg.t3 == C) \
        return launch_mlp<A, B, C>(b, n, m, nsample, cfeat, c3, xyz, new_xyz, pts, idx, wpacked, bpacked, out, st, 1, pooling)
    PN2_MLP_CASE(1, 1, 2);
    PN2_MLP_CASE(2, 2, 4);
    PN2_MLP_CASE(2, 3, 4);
#undef PN2_MLP_CASE
    return PN2_E_TOO_LARGE;
}

// variant: the organisation of the resident kernel -- 0: by the size rule, 1: one item per wave, 2 / 3: two items per wave
// (four / eight waves per workgroup) wherever that kernel covers the shape (results are bit-identical; tests and A/B timing)
extern "C" int pn2_sa_mlp3_maxpool_ex(int b, int n, int m, int nsample, int cfeat, const float *xyz, const float *new_xyz,
                                      const float *points, const int *idx, int c1, int c2, int c3, const float *wpacked,
                                      const float *bpacked, float *out, void *ws, int variant, void *stream)
{
    using namespace pn2;
    if (variant < 0 || variant > 3) return PN2_E_ARG;
    if (b < 0 || n <= 0 || m < 0 || cfeat < 0) return PN2_E_SHAPE;
    if (nsample <= 0) return PN2_E_ARG;
    if (b == 0 || m == 0) return PN2_OK;
    // idx == NULL and new_xyz == NULL together: the group_all level (sample_and_group_all, pointnet_util.py:59-84):
    // m = 1, the group is the whole cloud in index order (nsample = n), no centroid subtraction
    const bool group_all = !idx && !new_xyz;
    if (group_all && (m != 1 || nsample != n)) return PN2_E_ARG;
    if (!xyz || (!group_all && (!new_xyz || !idx)) || !wpacked || !bpacked || !out || (cfeat > 0 && !points)) return PN2_E_NULL;
    int kind;
    MlpConfig cfg;
    MlpStreamConfig sc;
    MlpCoopConfig cc;
    if (!mlp_choose(3 + cfeat, c1, c2, c3, nsample, kind, cfg, sc, cc)) return PN2_E_TOO_LARGE;
    if (group_all && kind != 2) return PN2_E_TOO_LARGE;           // only the cooperative kernel gathers without idx
    hipStream_t st = as_stream(stream);
    const float *pts = cfeat > 0 ? points : nullptr;
    if (kind == 2) {
        CoopParams p = {n, m, nsample, cfeat, 0, c3, cc.ti, (long long)b * m, xyz, new_xyz, pts, nullptr, idx, nullptr,
                        wpacked, bpacked, out, 0};
        return mlp_coop_launch(cc, 0, p, st, ws);
    }
    if (kind == 1)
        return mlp_stream_launch(sc, b, n, m, nsample, cfeat, c3, xyz, new_xyz, pts ? pts : xyz, idx, wpacked, bpacked, out, ws, st);
#define PN2_MLP_CASE(A, B, C) \
    if (cfg.t1 == A && cfg.t2 == B && cfg.t3 == C) \
        return launch_mlp<A, B, C>(b, n, m, nsample, cfeat, c3, xyz, new_xyz, pts, idx, wpacked, bpacked, out, st, variant)
    PN2_MLP_CASE(1, 1, 2);
    PN2_MLP_CASE(2, 2, 4);
    PN2_MLP_CASE(2, 3, 4);
#undef PN2_MLP_CASE
    return PN2_E_TOO_LARGE;
}
