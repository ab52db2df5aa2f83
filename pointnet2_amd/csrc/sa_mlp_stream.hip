// sa_mlp_stream.hip -- the fused grouped MLP + max-pool (see sa_mlp.hip for the formulation) for layer
// stacks whose weights do not fit in LDS: SA2/SA3-sized stacks such as 131 -> 128 -> 128 -> 256
// (264 KB of fp32 weights), reference models/pointnet2_*: pointnet_sa_module(... mlp=[128,128,256] ...).
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
//   * layers 1 and 2 walk the INPUT tiles in the outer loop (only one 32-channel tile of inputs is alive in
//     its three-level form, all output accumulators are); input channels are ordered [features, xyz] so that a
//     lane's four channels of a register quartet are one aligned 16-byte load when cfeat % 4 == 0;
//   * the last layer's 16 registers of a tile are max-reduced right away (they hold 16 samples of one
//     channel, see sa_mlp.hip), so the running maximum over a centroid's sample groups is T3 registers.
#include "sa_mlp_common.h"

#include <stdlib.h>

namespace pn2 {

template <int T1, int T2, int T3>
__global__ __launch_bounds__(kStreamThreads) void sa_mlp3_stream_kernel(int n, int m, int nsample, int cfeat, int c3, long long rows,
                                                                       int ti, const float *__restrict__ xyz,
                                                                       const float *__restrict__ new_xyz,
                                                                       const float *__restrict__ points,
                                                                       const int *__restrict__ idx,
                                                                       const float *__restrict__ wstream,
                                                                       const float *__restrict__ bpacked,
                                                                       float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) u32x4 wbuf[2][kStageVec];
    __shared__ float bias_s[(T1 + T2 + T3) * 32];
    const float *b1 = bias_s, *b2 = b1 + T1 * 32, *b3 = b2 + T2 * 32;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s = lane & 31;
    for (int i = tid; i < (T1 + T2 + T3) * 32; i += kStreamThreads) bias_s[i] = bpacked[i];

    const int l1_pairs = pad_to_stage(ti * T1);
    const int stages_per_item = (l1_pairs + T2 * T1 + T3 * T2) / kS;       // T2*T1 and T3*T2 are multiples of kS
    // ---- the weight stream -------------------------------------------------------------------------
    int stage = 0;                                   // running stage number (all items), uniform over the workgroup
    u32x4 stg0, stg1, stg2;                          // this thread's 48 bytes of stage `stage + 1`
    static_assert(kStageVec == 3 * kStreamThreads, "the staging registers are spelled out for three vectors per thread");
    // (macros, not lambdas over an array: hipcc kept a captured array in scratch memory)
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
    PN2_STREAM_ISSUE(0);
    PN2_STREAM_COMMIT(0);
    __syncthreads();
    PN2_STREAM_ISSUE(1);

    // ---- work items -----------------------------------------------------------------------------------
    const int parts = nsample / 32;
    const long long wave = (long long)blockIdx.x * (kStreamThreads / 64) + (tid >> 6);
    const long long nwaves = (long long)gridDim.x * (kStreamThreads / 64);
    const long long trips = (rows + nwaves - 1) / nwaves;                  // lockstep: every wave runs all trips
    const int cin = cfeat + 3;
    const bool vec4 = (cfeat & 3) == 0;

    // gather one 32-channel tile of layer-1 inputs for this lane's sample: register v <- channel
    // 32u + mlp_chan(v, h) in the order [features 0..cfeat-1, x, y, z]
    auto gather = [&](long long row, int p, int u) __attribute__((always_inline)) -> f32x16 {
        const long long cloud = row / m;
        const float *pf = points + ((size_t)cloud * n + p) * cfeat;
        const float *px = xyz + ((size_t)cloud * n + p) * 3;
        const float *c = new_xyz + row * 3;
        f32x16 x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = 32 * u + 8 * q + 4 * h;                         // channels k0 .. k0+3 -> registers 4q .. 4q+3
            if (vec4 && k0 + 3 < cfeat) {
                const float4 f = *reinterpret_cast<const float4 *>(pf + k0);
                x[4 * q] = f.x; x[4 * q + 1] = f.y; x[4 * q + 2] = f.z; x[4 * q + 3] = f.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = k0 + r;
                    float val = 0.0f;
                    if (k < cfeat) val = pf[k];
                    else if (k < cin) val = __fsub_rn(px[k - cfeat], c[k - cfeat]);
                    x[4 * q + r] = val;
                }
            }
        }
        return x;
    };

    for (long long trip = 0; trip < trips; ++trip) {
        const long long row_raw = wave + trip * nwaves;
        const bool row_ok = row_raw < rows;
        const long long row = row_ok ? row_raw : rows - 1;
        float best[T3];
        for (int part = 0; part < parts; ++part) {
            const int p = idx[row * nsample + part * 32 + s];
            // layer 1, input tiles outermost
            f32x16 h1[T1];
            {
#pragma unroll
                for (int t = 0; t < T1; ++t) h1[t] = mlp_bias(b1, t, h);
                f32x16 x = gather(row, p, 0);
                int slot = 0;
                for (int u = 0; u < ti; ++u) {
                    f32x16 xn = x;
                    if (u + 1 < ti) xn = gather(row, p, u + 1);            // one tile ahead of the MFMAs
                    const ActSplit xs = split_act(x);
#pragma unroll
                    for (int t = 0; t < T1; ++t) {
                        h1[t] = stream_pair<false>(wbuf[stage & 1], slot, lane, xs, h1[t]);
                        if (++slot == kS) { slot = 0; PN2_NEXT_STAGE(); }
                    }
                    x = xn;
                }
                if (slot != 0) PN2_NEXT_STAGE();                                // layer 1 is padded to whole stages
            }
            // layer 2, input tiles outermost as well: a tile of layer-1 output is split into its bf16 levels right
            // before its pairs run, so only ONE split input tile is alive beside the T2 accumulators
            ActSplit s2[T2];
            {
                f32x16 a2[T2];
#pragma unroll
                for (int t = 0; t < T2; ++t) a2[t] = mlp_bias(b2, t, h);
#pragma unroll
                for (int u = 0; u < T1; ++u) {
                    __builtin_amdgcn_sched_barrier(0);       // or hipcc splits every tile up front and spills
                    const ActSplit su = split_act(mlp_relu(h1[u]));
#pragma unroll
                    for (int t = 0; t < T2; ++t) {
                        a2[t] = stream_pair<false>(wbuf[stage & 1], (u * T2 + t) % kS, lane, su, a2[t]);
                        if ((u * T2 + t) % kS == kS - 1) PN2_NEXT_STAGE();
                    }
                }
#pragma unroll
                for (int t = 0; t < T2; ++t) s2[t] = split_act(mlp_relu(a2[t]));
            }
            // layer 3, operands swapped: a lane holds 16 samples of channel 32t + (l & 31)
#pragma unroll
            for (int t = 0; t < T3; ++t) {
                f32x16 acc;
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
                for (int u = 0; u < T2; ++u) {
                    acc = stream_pair<true>(wbuf[stage & 1], (t * T2 + u) % kS, lane, s2[u], acc);
                    if ((t * T2 + u) % kS == kS - 1) PN2_NEXT_STAGE();
                }
                float mx = acc[0];
#pragma unroll
                for (int v = 1; v < 16; ++v) mx = fmaxf(mx, acc[v]);
                best[t] = part == 0 ? mx : fmaxf(best[t], mx);
            }
        }
#pragma unroll
        for (int t = 0; t < T3; ++t) {
            const int ch = 32 * t + s;
            const float mx = fmaxf(best[t], __shfl_xor(best[t], 32));
            if (h == 0 && ch < c3 && row_ok) out[row * c3 + ch] = fmaxf(__fadd_rn(mx, b3_at(b3, ch)), 0.0f);
        }
    }
#undef PN2_STREAM_ISSUE
#undef PN2_STREAM_COMMIT
#undef PN2_NEXT_STAGE
}

bool mlp_stream_pick(int cin, int c1, int c2, int c3, MlpStreamConfig &cfg)
{
    static const int kShapes[][3] = {{2, 2, 4}, {4, 4, 8}};
    if (cin < 3 || cin > 32 * 12) return false;
    for (const auto &sh : kShapes)
        if (c1 <= 32 * sh[0] && c2 <= 32 * sh[1] && c3 <= 32 * sh[2]) {
            cfg = {(cin + 31) / 32, sh[0], sh[1], sh[2]};
            return true;
        }
    return false;
}

static int stream_pairs(const MlpStreamConfig &c) { return pad_to_stage(c.ti * c.t1) + c.t2 * c.t1 + c.t3 * c.t2; }
size_t mlp_stream_w_floats(const MlpStreamConfig &c) { return (size_t)stream_pairs(c) * kPairWords; }
size_t mlp_stream_b_floats(const MlpStreamConfig &c) { return (size_t)(c.t1 + c.t2 + c.t3) * 32; }

void mlp_stream_pack(const MlpStreamConfig &c, int cin, int c1, int c2, int c3, int xyz_first, const float *const *ws,
                     const float *const *bs, float *wpacked, float *bpacked)
{
    // kernel channel order of layer 1: [features, xyz]; caller's weight rows: [xyz, features] when xyz_first
    const int cfeat = cin - 3;
    int *krow = (int *)malloc(sizeof(int) * (size_t)cin);
    for (int k = 0; k < cin; ++k) krow[k] = xyz_first ? (k < cfeat ? 3 + k : k - cfeat) : k;
    float *wp = wpacked;
    for (int u = 0; u < c.ti; ++u)
        for (int t = 0; t < c.t1; ++t) wp = mlp_pack_pair_x6(wp, ws[0], cin, c1, t, u, krow);
    for (int i = c.ti * c.t1; i < pad_to_stage(c.ti * c.t1); ++i)
        for (int j = 0; j < kPairWords; ++j) *wp++ = 0.0f;
    for (int u = 0; u < c.t1; ++u)                        // layer 2 walks its input tiles outermost, like layer 1
        for (int t = 0; t < c.t2; ++t) wp = mlp_pack_pair_x6(wp, ws[1], c1, c2, t, u, nullptr);
    for (int t = 0; t < c.t3; ++t)
        for (int u = 0; u < c.t2; ++u) wp = mlp_pack_pair_x6(wp, ws[2], c2, c3, t, u, nullptr);
    free(krow);
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

template <int T1, int T2, int T3>
static int launch_stream(const MlpStreamConfig &c, int b, int n, int m, int nsample, int cfeat, int c3, const float *xyz,
                         const float *new_xyz, const float *points, const int *idx, const float *wp, const float *bp,
                         float *out, hipStream_t st)
{
    const long long rows = (long long)b * m;
    long long blocks = (rows + kStreamThreads / 64 - 1) / (kStreamThreads / 64);
    long long cap = 256;                                 // one workgroup (one weight stream) per CU
    if (blocks > cap) blocks = cap;
    if (int rc = launch((sa_mlp3_stream_kernel<T1, T2, T3>), dim3((unsigned)blocks), dim3(kStreamThreads), 0, st, n, m, nsample,
                       cfeat, c3, rows, c.ti, xyz, new_xyz, points, idx, wp, bp, out)) return rc;
    return PN2_OK;
}

int mlp_stream_launch(const MlpStreamConfig &c, int b, int n, int m, int nsample, int cfeat, int c3, const float *xyz,
                      const float *new_xyz, const float *points, const int *idx, const float *wp, const float *bp,
                      float *out, hipStream_t st)
{
    if (nsample <= 0 || nsample % 32 != 0) return PN2_E_ARG;
    if (c.t1 == 2 && c.t2 == 2 && c.t3 == 4)
        return launch_stream<2, 2, 4>(c, b, n, m, nsample, cfeat, c3, xyz, new_xyz, points, idx, wp, bp, out, st);
    if (c.t1 == 4 && c.t2 == 4 && c.t3 == 8)
        return launch_stream<4, 4, 8>(c, b, n, m, nsample, cfeat, c3, xyz, new_xyz, points, idx, wp, bp, out, st);
    return PN2_E_TOO_LARGE;
}

}  // namespace pn2
