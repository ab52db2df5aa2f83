// fps_pruned_body.h -- farthest point sampling with EXACT spatial pruning of the distance update (round 5).
//
// Same results as fps_reg_body (fps_body.h) / the reference kernel (tf_sampling_g.cu:105-170), bit for bit. What
// changes is how much of a round's work is done. In fps_reg_body every round updates the running min-distance of
// EVERY point against the new sample: 9 fp32 operations per point, 36,864 per round at n = 4096 -- a third of the
// round on one CU's vector units (the rest is the arg-max exchange). But the new sample can only lower the running
// distance of points closer to it than the current farthest-point distance v* (every running distance is <= v*, the
// value that just won the arg-max), and after a few dozen samples that ball holds a few per cent of the cloud.
//
// Organisation. 256 threads (four waves, one per SIMD), P = 16 or 32 rank slots per thread as in fps_reg_body, but the
// points are dealt to the slots SPATIALLY: the cloud is cut into 32 leaves of equal size (three counting passes in LDS,
// below), a leaf becomes a GROUP = GS = 2 or 4 slots of all 64 lanes of one wave (eight groups per wave), and every group
// keeps the tight bounding box of its 64 * GS points. Per round, lane l of every wave tests group l's box against the new
// sample (14 vector instructions for all groups at once); a group whose box lies farther from the sample than sqrt(v*)
// cannot change and is SKIPPED, a wave none of whose groups is touched skips its whole reduction and re-publishes its
// cached wave key. The tie rule is untouched: a slot's key is still (value bits : NS - 1 - rank) with
// rank = (k mod 512) * ceil(n / 512) + k / 512, it just lives in a register loaded once instead of being derived from
// the thread number, and the winner's (x, y, z, k) still comes from the LDS mirror kept in rank order.
//
// Exactness of the skip. Let bd be the fp32-evaluated squared distance from the sample s to a group's box. For a point p
// of the group the exact |p - s|^2 >= the exact box distance; the fp32 evaluations of both carry relative errors below
// 1e-6 and absolute errors (underflow) below 1e-37. The group is skipped only when bd >= v* * 1.00001f + 1e-30f, which
// therefore implies d_fp32(p, s) >= v* >= mind[p] for every p in the group: min(d, mind[p]) = mind[p], exactly what the
// reference computes. Non-finite coordinates (unspecified in the reference, as NaN is) can only make boxes infinite,
// i.e. tests fail towards "update". Measured on the bench clouds (simulation of this exact test in numpy, then the
// GPU): 2.6 of 16 slots per thread are updated per round at n = 4096 -> 1024 on sphere-surface clouds, 3.2 on
// uniform-cube clouds, 3.5-4.6 of 32 at n = 8192; a cloud with 87 % of its points on one spot (provider.py:227-233)
// prunes little (10 of 16) and costs what fps_reg_body costs.
//
// Grouping. Any partition gives the same samples; a compact one gives more skips. A balanced kd-tree built level by level
// (histogram medians with per-segment axis choice and tickets for the median bin: the first version of this file) grouped well
// but took 27-34 us per cloud -- as much as the pruning saved at n = 4096. The three counting passes of the body (4 x 4 x 2
// parts along the axes sorted by extent, one returning LDS atomic and one LDS read per item and pass, six barriers)
// prune as well in simulation and on the GPU (profiles/r05/fps_pruned.txt).
#pragma once
#include "fps_body.h"

#include <math.h>

#include <type_traits>

namespace pn2 {

#ifndef PN2_PR_DISPATCH
#define PN2_PR_DISPATCH 1
#endif
// Lab switches of round 6 (scripts/build_labs.sh, profiles/r06/fps_round6.txt); the product builds with both at 0.
//   PN2_PR_EARLY_ALL = J: rounds 1 .. J update EVERY group in straight-line code, without the box test and the dispatch
//                         (the first rounds prune nothing and pay the pruned round);
//   PN2_PR_SPEC_MIRROR = 4 / 2: the mirror rows of all four wave candidates (or of the two semi-final winners) are read
//                         BEFORE the tournament is decided and the winner's row is selected in registers (next 2d).
#ifndef PN2_PR_EARLY_ALL
#define PN2_PR_EARLY_ALL 0
#endif
#ifndef PN2_PR_SPEC_MIRROR
#define PN2_PR_SPEC_MIRROR 0
#endif
constexpr int kPrT = 256;              // threads of the pruned tier
constexpr int kPrW = kPrT / PN2_WAVE;  // 4 waves: one per SIMD
constexpr int kPrBins = 64;            // histogram bins per axis (= one wave)
constexpr int kPrK0 = 4, kPrK1 = 4, kPrK2 = 2;                 // parts per pass: 32 leaves
constexpr int kPrGroups = kPrK0 * kPrK1 * kPrK2;
constexpr int kPrHistRows = 1 + kPrK0 + kPrK0 * kPrK1;         // segments of the three passes
// small clouds (the batched tier at 1024 / 2048 rank slots) are cut into 8 / 16 leaves: 2 x 2 x 2 and 4 x 2 x 2 parts
constexpr int pr_k0(int G) { return G >= 16 ? 4 : 2; }
constexpr int pr_k1(int G) { return G == 32 ? 4 : 2; }
constexpr int pr_k2(int) { return 2; }

// LDS layout (bytes): [0,64) wave keys (2 parities x 4) | [64,256) reduction scratch | mirror / staging 16 * NS |
// hist 21 x 64 ints | gbox 32 x 8 floats
__host__ __device__ constexpr size_t fps_pruned_lds_bytes(int P, int T = kPrT)
{
    return 256 + (size_t)16 * T * P + (size_t)kPrHistRows * kPrBins * 4 + (size_t)kPrGroups * 32;
}

// Where the pruned tier pays (measured, profiles/r05/fps_pruned.txt): its grouping costs 8.5 us (16 slots per thread) / 14 us
// (32) more than fps_reg_body's prologue and its first ~64 rounds prune nothing; after that a round is 353-362 ns against 392
// at 4096 rank slots and 386-395 against 545 at 8192. ranks = 512 * ceil(n / 512).
inline bool fps_pruned_pays(int ranks, int m)
{
    if (ranks > 2048 && ranks <= 4096) return m >= 768;
    return ranks > 4096 && ranks <= 8192 && m >= 128;
}

__device__ __forceinline__ int pr_prefix_sum_incl(int v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); v += t;    // row_shr:1
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true); v += t;    // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); v += t;    // row_shr:4
    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true); v += t;    // row_shr:8
    t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); v += t;   // row_bcast:15 -> rows 1,3
    t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); v += t;   // row_bcast:31 -> rows 2,3
    return v;
}

// wave-wide min / max, result in lane 63 (a lane without a DPP source keeps its own value)
template <bool MAX>
__device__ __forceinline__ float pr_minmax_lane63(float v)
{
#define PN2_PR_STEP(ctrl, rmask)                                                                               \
    {                                                                                                          \
        const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, \
                                                                   rmask, 0xf, false));                        \
        v = MAX ? fmaxf(v, o) : fminf(v, o);                                                                   \
    }
    PN2_PR_STEP(0x111, 0xf) PN2_PR_STEP(0x112, 0xf) PN2_PR_STEP(0x114, 0xf) PN2_PR_STEP(0x118, 0xf)
    PN2_PR_STEP(0x142, 0xa) PN2_PR_STEP(0x143, 0xc)
#undef PN2_PR_STEP
    return v;
}

// The slots of one thread after the grouping: coordinates as register pairs, running distances, key low words, and (lane l
// = group l mod 32) the tight bounding box of a group.
template <int P>
struct PrSlots {
    pn2_f2 xx[P / 2], yy[P / 2], zz[P / 2];
    float md[P];
    unsigned low[P];
    float blx, bly, blz, bhx, bhy, bhz;
};

constexpr int kPrPrologueBarriers = 10;   // __syncthreads() executed by fps_pruned_prologue (a wave that sits the prologue out must match them)

// Grouping prologue shared by the chains of fps_pruned_body (one sample per exchange) and fps_batch_body.h (several): deals the
// cloud to the slots spatially, writes the rank-ordered LDS mirror and the group boxes. Ends behind a barrier.
// T = 256 threads (four waves of eight groups: the pruned tier) or 512 (eight waves of four groups: the batched tier's updaters).
template <int P, int GS, int T = kPrT>
__device__ __forceinline__ void fps_pruned_prologue(int n, int Q, const float *__restrict__ src, char *smem, PrSlots<P> &S)
{
    constexpr int W = T / PN2_WAVE, NS = T * P;
    constexpr int GW = P / GS;                    // groups per wave
    constexpr int G = W * GW;                     // groups = leaves = test lanes
    static_assert((G == kPrGroups || ((G == 16 || G == 8) && W == 8)) && (W == 4 || W == 8) && (GS & 1) == 0,
                  "32 groups: four waves of eight or eight waves of four; 16 / 8 groups: eight waves of two / one");
    constexpr int K0 = pr_k0(G), K1 = pr_k1(G), K2 = pr_k2(G), R = K1 * K2;
    static_assert(K0 * K1 * K2 == G && NS / G == 64 * GS, "a leaf = GS slots of all 64 lanes of one wave");
    float *scratch = reinterpret_cast<float *>(smem + (W > 4 ? 0 : 64));                // 8 floats per wave (eight waves: the whole 256-byte header)
    float4 *lds_rank = reinterpret_cast<float4 *>(smem + 256);                          // mirror [NS], first the staging copy
    char *tab = smem + 256 + (size_t)16 * NS;
    int *hist = reinterpret_cast<int *>(tab);                                           // [1 + K0 + K0 * K1][64]: one histogram row per segment and level
    float *gbox = reinterpret_cast<float *>(tab + (size_t)kPrHistRows * kPrBins * 4);   // [G][8]
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    pn2_f2 (&xx)[P / 2] = S.xx;
    pn2_f2 (&yy)[P / 2] = S.yy;
    pn2_f2 (&zz)[P / 2] = S.zz;
    float (&md)[P] = S.md;
    unsigned (&low)[P] = S.low;

    // ---- the items in natural order: item i = point i, or (i >= n) a padding item at point 0's position ----------------
    float px[P], py[P], pz[P];
    float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int i = t + j * T;
        const int k = i < n ? i : 0;
        px[j] = src[(size_t)k * 3 + 0]; py[j] = src[(size_t)k * 3 + 1]; pz[j] = src[(size_t)k * 3 + 2];
        lx = fminf(lx, px[j]); ly = fminf(ly, py[j]); lz = fminf(lz, pz[j]);
        hx = fmaxf(hx, px[j]); hy = fmaxf(hy, py[j]); hz = fmaxf(hz, pz[j]);
    }
    lx = pr_minmax_lane63<false>(lx); ly = pr_minmax_lane63<false>(ly); lz = pr_minmax_lane63<false>(lz);
    hx = pr_minmax_lane63<true>(hx); hy = pr_minmax_lane63<true>(hy); hz = pr_minmax_lane63<true>(hz);
    if (lane == 63) {
        scratch[w * 8 + 0] = lx; scratch[w * 8 + 1] = ly; scratch[w * 8 + 2] = lz;
        scratch[w * 8 + 4] = hx; scratch[w * 8 + 5] = hy; scratch[w * 8 + 6] = hz;
    }
    for (int i = t; i < kPrHistRows * kPrBins; i += T) hist[i] = 0;
    __syncthreads();

    // ---- spatial grouping: three counting passes --------------------------------------------------------------------------
    // The cloud's bounding box is cut into 64 bins per axis once. Pass 1 cuts the cloud into K0 = 4 equal parts along its
    // widest axis, pass 2 every part into K1 = 4 along the second widest, pass 3 every piece into K2 = 2 along the third:
    // 32 leaves of NS / 32 items. A pass is: every item takes a ticket in its (segment, bin) counter (one returning LDS
    // atomic: the counters end up as the histogram), a wave turns each segment's histogram into exclusive prefix sums,
    // and an item's RANK inside its segment is prefix[bin] + ticket -- its child is rank / child size. Items of one bin are
    // ordered by ticket, i.e. arbitrarily: which of them lands on which side of a cut is timing dependent and irrelevant
    // (header). The last pass's rank is also the item's position inside its leaf.
    float blo[3], bsc[3];
    int ax[3];
    {
        float lo3[3], ex[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float l = scratch[a], h = scratch[4 + a];
#pragma unroll
            for (int ww = 1; ww < W; ++ww) { l = fminf(l, scratch[ww * 8 + a]); h = fmaxf(h, scratch[ww * 8 + 4 + a]); }
            const float e = h - l;
            const bool ok = e > 0.0f && e < INFINITY;     // degenerate / non-finite axis: every item in bin 0, tickets cut it
            lo3[a] = ok ? l : 0.0f;
            ex[a] = ok ? e : 0.0f;
        }
        // axes by extent, widest first (wave-uniform values; ties keep x, y, z order)
        int a0 = 0, a1 = 1, a2 = 2;
        if (ex[a1] > ex[a0]) { const int q = a0; a0 = a1; a1 = q; }
        if (ex[a2] > ex[a0]) { const int q = a0; a0 = a2; a2 = q; }
        if (ex[a2] > ex[a1]) { const int q = a1; a1 = a2; a2 = q; }
        ax[0] = a0; ax[1] = a1; ax[2] = a2;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int a = ax[l];
            const float e = a == 0 ? ex[0] : a == 1 ? ex[1] : ex[2];
            blo[l] = a == 0 ? lo3[0] : a == 1 ? lo3[1] : lo3[2];
            bsc[l] = e > 0.0f ? (float)kPrBins / e : 0.0f;
        }
    }
    int seg[P], rank[P];
#pragma unroll
    for (int j = 0; j < P; ++j) seg[j] = 0;
    constexpr int kLevelK[3] = {K0, K1, K2};
    constexpr int kLevelRow[3] = {0, 1, 1 + K0};                      // first histogram row of the level
    constexpr int kLevelSegs[3] = {1, K0, K0 * K1};
#pragma unroll
    for (int lev = 0; lev < 3; ++lev) {
        int *hl = hist + kLevelRow[lev] * kPrBins;
#pragma unroll
        for (int j = 0; j < P; ++j) {                                            // seg[j] becomes the item's counter: segment * 64 + bin
            const float c = ax[lev] == 0 ? px[j] : ax[lev] == 1 ? py[j] : pz[j];
            int q = (int)((c - blo[lev]) * bsc[lev]);                             // NaN -> 0
            q = q < 0 ? 0 : q > kPrBins - 1 ? kPrBins - 1 : q;
            seg[j] = seg[j] * kPrBins + q;
        }
#pragma unroll
        for (int j = 0; j < P; ++j) rank[j] = atomicAdd(&hl[seg[j]], 1);        // the ticket; all P in flight
        __syncthreads();
        for (int sg = w; sg < kLevelSegs[lev]; sg += W) {                        // counts -> exclusive prefix sums, in place
            const int h = hl[sg * kPrBins + lane];
            hl[sg * kPrBins + lane] = pr_prefix_sum_incl(h) - h;
        }
        __syncthreads();
        const int child = NS / (lev == 0 ? K0 : lev == 1 ? K0 * K1 : K0 * K1 * K2);   // a power of two, constant once unrolled
#pragma unroll
        for (int j = 0; j < P; ++j) {
            rank[j] += hl[seg[j]];
            seg[j] = (seg[j] >> 6) * kLevelK[lev] + (int)((unsigned)rank[j] / (unsigned)child);
            rank[j] &= child - 1;                                               // rank inside the child
        }
    }

    // ---- scatter into the slots. Leaf id = 8 * a + r (a = part along the widest axis, r = the rest): wave (r + a) % 4 --
    // neighbours along every axis go to different waves (simulation: the busiest wave has 1.26 touched groups per round
    // with this map, 1.56 with id % 4) --, group 2 * a + r / 4 of that wave; position p of the leaf -> lane p % 64, slot p / 64
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int a = seg[j] / R, r = seg[j] % R;
        // 32 leaves: wave (r + a) % W, group 2 a + r / 4 (four waves) or a (eight waves); fewer leaves (eight waves): wave id % 8,
        // group id / 8 -- neighbours along every axis still land on different waves
        const int dt = (G == kPrGroups ? ((r + a) & (W - 1)) : (seg[j] & (W - 1))) * 64 + (rank[j] & 63);
        const int dp = (G == kPrGroups ? (a * (8 / W) + r / W) : seg[j] / W) * GS + (rank[j] >> 6);
        lds_rank[dt * P + dp] = make_float4(px[j], py[j], pz[j], __int_as_float(t + j * T));
    }
    __syncthreads();
    // md: until the mirror is written, the bits of the slot's item number
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float4 v = lds_rank[t * P + p];
        const int i = __float_as_int(v.w);
        const int rank = (i & (kRefThreads - 1)) * Q + (i >> 9);      // tie rank of point i (fps_body.h)
        low[p] = i < n ? (unsigned)(NS - 1 - rank) : 0u;               // padding: value 0, lowest key (never wins over rank 0)
        md[p] = v.w;
        if (p & 1) { xx[p / 2].y = v.x; yy[p / 2].y = v.y; zz[p / 2].y = v.z; }
        else { xx[p / 2].x = v.x; yy[p / 2].x = v.y; zz[p / 2].x = v.z; }
    }
    __syncthreads();                                    // everybody has read the staging copy: the region becomes the mirror
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float X = (p & 1) ? xx[p / 2].y : xx[p / 2].x, Y = (p & 1) ? yy[p / 2].y : yy[p / 2].x,
                    Z = (p & 1) ? zz[p / 2].y : zz[p / 2].x;
        const int i = __float_as_int(md[p]);
        const bool real = i < n;
        if (real) lds_rank[low[p]] = make_float4(X, Y, Z, md[p]);     // (x, y, z, bits of k)
        md[p] = real ? 1e38f : 0.0f;                                   // tf_sampling_g.cu:118
    }
    // tight box of every group of this wave -> gbox[w * GW + gi]
#pragma unroll
    for (int gi = 0; gi < GW; ++gi) {
        float blx = INFINITY, bly = INFINITY, blz = INFINITY, bhx = -INFINITY, bhy = -INFINITY, bhz = -INFINITY;
#pragma unroll
        for (int h = 0; h < GS / 2; ++h) {
            const pn2_f2 a = xx[gi * (GS / 2) + h], b = yy[gi * (GS / 2) + h], c = zz[gi * (GS / 2) + h];
            blx = fminf(blx, fminf(a.x, a.y)); bhx = fmaxf(bhx, fmaxf(a.x, a.y));
            bly = fminf(bly, fminf(b.x, b.y)); bhy = fmaxf(bhy, fmaxf(b.x, b.y));
            blz = fminf(blz, fminf(c.x, c.y)); bhz = fmaxf(bhz, fmaxf(c.x, c.y));
        }
        blx = pr_minmax_lane63<false>(blx); bly = pr_minmax_lane63<false>(bly); blz = pr_minmax_lane63<false>(blz);
        bhx = pr_minmax_lane63<true>(bhx); bhy = pr_minmax_lane63<true>(bhy); bhz = pr_minmax_lane63<true>(bhz);
        if (lane == 63) {
            float *o = gbox + (w * GW + gi) * 8;
            o[0] = blx; o[1] = bly; o[2] = blz; o[4] = bhx; o[5] = bhy; o[6] = bhz;
        }
    }
    __syncthreads();
    // lane l tests group l (lanes beyond G repeat group l mod G; their bits are never looked at)
    {
        const float *o = gbox + (lane & (G - 1)) * 8;
        S.blx = o[0]; S.bly = o[1]; S.blz = o[2]; S.bhx = o[4]; S.bhy = o[5]; S.bhz = o[6];
    }
}

// P rank slots per thread, GS slots per group (both powers of two, GS even: the update runs on register pairs).
template <int P, int GS, bool PUBLISH>
__device__ __forceinline__ void fps_pruned_body(int n, int m, int Q, int cloud, const float *__restrict__ xyz,
                                                int *__restrict__ out, float *__restrict__ out_xyz,
                                                unsigned long long *__restrict__ tagged, char *smem, unsigned tag = 1u)
{
    // A chain is one dependent instruction after another on b CUs; beside another stream's kernels (the layer stacks of the
    // previous batch, pointnet2_amd/geometry.py) its waves would queue behind theirs at every issue. Highest wave priority: the
    // neighbours lose a few issue slots on b of 256 CUs, the chain keeps its pace (sem_seg training step with the geometry one
    // step ahead: 10.5 ms without this line, profiles/r05/geometry_ahead.txt).
#ifndef PN2_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    constexpr int T = kPrT, W = kPrW, NS = T * P;
    constexpr int GW = P / GS;                    // groups per wave
    constexpr int G = W * GW;                     // groups = leaves = test lanes
    static_assert(G == kPrGroups && GW == 8 && (GS & 1) == 0, "32 groups: 16 slots per thread in groups of 2, 32 in groups of 4");

    unsigned long long *partial = reinterpret_cast<unsigned long long *>(smem);        // [2][W]
    float4 *lds_rank = reinterpret_cast<float4 *>(smem + 256);                          // mirror [NS], first the staging copy

    const float *__restrict__ src = xyz + (size_t)cloud * n * 3;
    int *__restrict__ dst = out + (size_t)cloud * m;
    float *__restrict__ dxyz = out_xyz ? out_xyz + (size_t)cloud * m * 3 : nullptr;
    pn2_gu64 *gtag = PUBLISH ? (pn2_gu64 *)(tagged + (size_t)cloud * m) : nullptr;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);

    PrSlots<P> S;
    fps_pruned_prologue<P, GS>(n, Q, src, smem, S);
    pn2_f2 (&xx)[P / 2] = S.xx;
    pn2_f2 (&yy)[P / 2] = S.yy;
    pn2_f2 (&zz)[P / 2] = S.zz;
    float (&md)[P] = S.md;
    unsigned (&low)[P] = S.low;
    const float blx = S.blx, bly = S.bly, blz = S.blz, bhx = S.bhx, bhy = S.bhy, bhz = S.bhz;

    // ---- the chain -------------------------------------------------------------------------------------------------------
    // the point selected last (starts at k = 0 = rank 0), every coordinate in the LOW half of a register pair: the
    // high-half broadcast form of v_pk_add_f32 is not safe beside other kernels' MFMAs (fps_body.h, pk_sub_bcast_lo)
    pn2_f2 sxy, syy = {0.f, 0.f}, szk;
    {
        const float4 s = lds_rank[NS - 1];
        sxy.x = s.x; sxy.y = s.y; syy.x = s.y; szk.x = s.z; szk.y = s.w;
    }
    float vstar = 1e38f;                                // every running distance is <= vstar
    if (t == 0) {
        dst[0] = 0;                                     // tf_sampling_g.cu:114-116
        if (PUBLISH) __hip_atomic_store(gtag, (unsigned long long)tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // lane-level key of every group and of every pair of groups, and (lane 63) the wave's key: cached across rounds, so they
    // start as the keys of the INITIAL running distances (1e38 : low) -- a group whose box is farther than ~1e19 from point 0 is
    // skipped in round 1 already, and its points must then compete with the reference's td = 1e38 (tf_sampling_g.cu:118), not
    // with a key of 0
    double gk[GW], gp[GW / 2];
#pragma unroll
    for (int gi = 0; gi < GW; ++gi) {
        double kd[GS];
#pragma unroll
        for (int q = 0; q < GS; ++q) kd[q] = __hiloint2double(__float_as_int(md[gi * GS + q]), (int)low[gi * GS + q]);
#pragma unroll
        for (int st = 1; st < GS; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < GS; i += 2 * st)
                asm("v_max_f64 %0, %1, %2" : "=v"(kd[i]) : "v"(kd[i]), "v"(kd[i + st]));
        gk[gi] = kd[0];
    }
#pragma unroll
    for (int q = 0; q < GW / 2; ++q) asm("v_max_f64 %0, %1, %2" : "=v"(gp[q]) : "v"(gk[2 * q]), "v"(gk[2 * q + 1]));
    double wave_key;                                    // lane 63: the wave's key (cached while no group of the wave changes)
    {
        double k01, k23, kl;
        asm("v_max_f64 %0, %1, %2" : "=v"(k01) : "v"(gp[0]), "v"(gp[1]));
        asm("v_max_f64 %0, %1, %2" : "=v"(k23) : "v"(gp[2]), "v"(gp[3]));
        asm("v_max_f64 %0, %1, %2" : "=v"(kl) : "v"(k01), "v"(k23));
        wave_key = wave_max_f64_lane63(kl);
    }
    int kprev = 0;                                      // index selected by the previous round (stored one round late, below)

    auto round = [&](const int j, const int par, auto allc) __attribute__((always_inline)) {
        constexpr bool ALL = decltype(allc)::value;
        unsigned mybits = (1u << GW) - 1u;
        if constexpr (!ALL) {
        // which groups can the new sample change? (all lanes, lane l = group l)
        const float thr = __fadd_rn(__fmul_rn(vstar, 1.00001f), 1e-30f);
        // distance from the sample to the box = sample - clamp(sample, lo, hi) per axis (v_med3_f32 is the clamp)
        const float ax = __fsub_rn(sxy.x, __builtin_amdgcn_fmed3f(sxy.x, blx, bhx));
        const float ay = __fsub_rn(syy.x, __builtin_amdgcn_fmed3f(syy.x, bly, bhy));
        const float az = __fsub_rn(szk.x, __builtin_amdgcn_fmed3f(szk.x, blz, bhz));
        const float bd = __fadd_rn(__fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay)), __fmul_rn(az, az));
        const unsigned long long far_mask = __ballot(bd >= thr);          // NaN -> not far -> updated
        mybits = (unsigned)(~far_mask >> (w * GW)) & ((1u << GW) - 1u);   // scalar
        }
        unsigned long long *slot = partial + par * W;
        if (ALL || mybits != 0u) {                                        // wave-uniform
            auto update_group = [&](auto gic) __attribute__((always_inline)) {
                constexpr int gi = decltype(gic)::value;
                constexpr int H = GS / 2;
                pn2_f2 dx[H], dy[H], dz[H];
#pragma unroll
                for (int h = 0; h < H; ++h) dx[h] = pk_sub_bcast_lo(xx[gi * H + h], sxy);
#pragma unroll
                for (int h = 0; h < H; ++h) dy[h] = pk_sub_bcast_lo(yy[gi * H + h], syy);
#pragma unroll
                for (int h = 0; h < H; ++h) dz[h] = pk_sub_bcast_lo(zz[gi * H + h], szk);
#pragma unroll
                for (int h = 0; h < H; ++h) dx[h] = pk_mul(dx[h], dx[h]);
#pragma unroll
                for (int h = 0; h < H; ++h) dy[h] = pk_mul(dy[h], dy[h]);
#pragma unroll
                for (int h = 0; h < H; ++h) dz[h] = pk_mul(dz[h], dz[h]);
#pragma unroll
                for (int h = 0; h < H; ++h) dx[h] = pk_add(dx[h], dy[h]);
#pragma unroll
                for (int h = 0; h < H; ++h) dx[h] = pk_add(dx[h], dz[h]);
                double kd[GS];
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const int p0 = gi * GS + 2 * h;
                    md[p0] = vmin_f32(dx[h].x, md[p0]);              // min(d,td), :144
                    md[p0 + 1] = vmin_f32(dx[h].y, md[p0 + 1]);
                    kd[2 * h] = __hiloint2double(__float_as_int(md[p0]), (int)low[p0]);
                    kd[2 * h + 1] = __hiloint2double(__float_as_int(md[p0 + 1]), (int)low[p0 + 1]);
                }
#pragma unroll
                for (int st = 1; st < GS; st <<= 1)
#pragma unroll
                    for (int i = 0; i + st < GS; i += 2 * st)
                        asm("v_max_f64 %0, %1, %2" : "=v"(kd[i]) : "v"(kd[i]), "v"(kd[i + st]));
                gk[gi] = kd[0];
            };
#define PN2_PR_PAIR(q) asm("v_max_f64 %0, %1, %2" : "=v"(gp[q]) : "v"(gk[2 * (q)]), "v"(gk[2 * (q) + 1]))
#define PN2_PR_ONE(g) { update_group(std::integral_constant<int, g>()); PN2_PR_PAIR((g) / 2); }
#define PN2_PR_GROUP(g) if (rest & (1u << (g))) update_group(std::integral_constant<int, g>())
            // Scalar tests and branches, not vector work, are what a touched wave spends its time on (a taken or untaken
            // s_cmp + s_cbranch pair costs as much as four vector instructions on a lone wave; the update of a group is 13).
            // The LOWEST touched group -- the only one in most rounds -- is found by a binary search on its isolated bit
            // (three compare + branch pairs); further groups take nested bit tests (structured: hipcc turns a switch on
            // the bit number, or a loop over the set bits, into flag variables and copies of every slot register).
            if constexpr (ALL) {
                update_group(std::integral_constant<int, 0>()); update_group(std::integral_constant<int, 1>()); PN2_PR_PAIR(0);
                update_group(std::integral_constant<int, 2>()); update_group(std::integral_constant<int, 3>()); PN2_PR_PAIR(1);
                update_group(std::integral_constant<int, 4>()); update_group(std::integral_constant<int, 5>()); PN2_PR_PAIR(2);
                update_group(std::integral_constant<int, 6>()); update_group(std::integral_constant<int, 7>()); PN2_PR_PAIR(3);
            } else {
#if PN2_PR_DISPATCH == 2
            for (unsigned bits = mybits; bits != 0u;) {
                const unsigned lowbit = bits & (0u - bits);
                bits ^= lowbit;
                if (lowbit < 0x10u) {
                    if (lowbit < 0x04u) { if (lowbit == 0x01u) PN2_PR_ONE(0) else PN2_PR_ONE(1) }
                    else { if (lowbit == 0x04u) PN2_PR_ONE(2) else PN2_PR_ONE(3) }
                } else {
                    if (lowbit < 0x40u) { if (lowbit == 0x10u) PN2_PR_ONE(4) else PN2_PR_ONE(5) }
                    else { if (lowbit == 0x40u) PN2_PR_ONE(6) else PN2_PR_ONE(7) }
                }
            }
#elif PN2_PR_DISPATCH == 1
            const unsigned lowbit = mybits & (0u - mybits);
            const unsigned rest = mybits ^ lowbit;
            if (lowbit < 0x10u) {
                if (lowbit < 0x04u) { if (lowbit == 0x01u) PN2_PR_ONE(0) else PN2_PR_ONE(1) }
                else { if (lowbit == 0x04u) PN2_PR_ONE(2) else PN2_PR_ONE(3) }
            } else {
                if (lowbit < 0x40u) { if (lowbit == 0x10u) PN2_PR_ONE(4) else PN2_PR_ONE(5) }
                else { if (lowbit == 0x40u) PN2_PR_ONE(6) else PN2_PR_ONE(7) }
            }
            if (rest != 0u) {
                if (rest & 0x0fu) {
                    if (rest & 0x03u) { PN2_PR_GROUP(1); PN2_PR_PAIR(0); }
                    if (rest & 0x0cu) { PN2_PR_GROUP(2); PN2_PR_GROUP(3); PN2_PR_PAIR(1); }
                }
                if (rest & 0xf0u) {
                    if (rest & 0x30u) { PN2_PR_GROUP(4); PN2_PR_GROUP(5); PN2_PR_PAIR(2); }
                    if (rest & 0xc0u) { PN2_PR_GROUP(6); PN2_PR_GROUP(7); PN2_PR_PAIR(3); }
                }
            }
#else
#define PN2_PR_GROUP0(g) if (mybits & (1u << (g))) update_group(std::integral_constant<int, g>())
            if (mybits & 0x0fu) {
                if (mybits & 0x03u) { PN2_PR_GROUP0(0); PN2_PR_GROUP0(1); PN2_PR_PAIR(0); }
                if (mybits & 0x0cu) { PN2_PR_GROUP0(2); PN2_PR_GROUP0(3); PN2_PR_PAIR(1); }
            }
            if (mybits & 0xf0u) {
                if (mybits & 0x30u) { PN2_PR_GROUP0(4); PN2_PR_GROUP0(5); PN2_PR_PAIR(2); }
                if (mybits & 0xc0u) { PN2_PR_GROUP0(6); PN2_PR_GROUP0(7); PN2_PR_PAIR(3); }
            }
#undef PN2_PR_GROUP0
#endif
            }
#undef PN2_PR_ONE
#undef PN2_PR_PAIR
#undef PN2_PR_GROUP
            // lane key = max over the four cached pair keys (only the touched pairs were recomputed above)
            double k01, k23, kl;
            asm("v_max_f64 %0, %1, %2" : "=v"(k01) : "v"(gp[0]), "v"(gp[1]));
            asm("v_max_f64 %0, %1, %2" : "=v"(k23) : "v"(gp[2]), "v"(gp[3]));
            asm("v_max_f64 %0, %1, %2" : "=v"(kl) : "v"(k01), "v"(k23));
            wave_key = wave_max_f64_lane63(kl);
        }
        if (lane == 63) reinterpret_cast<double *>(slot)[w] = wave_key;
        __syncthreads();
        const double *dslot = reinterpret_cast<const double *>(slot);
        double key[W];
#pragma unroll
        for (int i = 0; i < W; ++i) key[i] = dslot[i];
        // the PREVIOUS round's index leaves here, under the latency of the key reads (behind the mirror read its address
        // arithmetic and exec juggling were 14 cycles of every round: scripts/ubench_xchg.hip, kinds 14 / 15)
        if (t == 0 && j > 1) {
            dst[j - 1] = kprev;
            if (PUBLISH)
                __hip_atomic_store(gtag + (j - 1), ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)kprev,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#if PN2_PR_SPEC_MIRROR == 4
        // lab: all four candidates' mirror rows are requested as soon as the keys are there, the tournament runs under their
        // latency, the winner's row is selected in registers
        static_assert(W == 4, "four waves");
        float4 cand[W];
#pragma unroll
        for (int i = 0; i < W; ++i) cand[i] = lds_rank[(unsigned)__double2loint(key[i])];
        double k01, k23, kw;
        asm("v_max_f64 %0, %1, %2" : "=v"(k01) : "v"(key[0]), "v"(key[1]));
        asm("v_max_f64 %0, %1, %2" : "=v"(k23) : "v"(key[2]), "v"(key[3]));
        asm("v_max_f64 %0, %1, %2" : "=v"(kw) : "v"(k01), "v"(k23));
        const bool lo_half = __double2loint(kw) == __double2loint(k01) && __double2hiint(kw) == __double2hiint(k01);
        const bool first = lo_half ? (__double2loint(kw) == __double2loint(key[0]) && __double2hiint(kw) == __double2hiint(key[0]))
                                   : (__double2loint(kw) == __double2loint(key[2]) && __double2hiint(kw) == __double2hiint(key[2]));
        const float4 sa = lo_half ? cand[0] : cand[2], sb = lo_half ? cand[1] : cand[3];
        const float4 s = first ? sa : sb;
        vstar = __int_as_float(__double2hiint(kw));
#elif PN2_PR_SPEC_MIRROR == 2
        // lab: semi-finals first, the two semi-final winners' rows requested, the final runs under their latency
        static_assert(W == 4, "four waves");
        double k01, k23, kw;
        asm("v_max_f64 %0, %1, %2" : "=v"(k01) : "v"(key[0]), "v"(key[1]));
        asm("v_max_f64 %0, %1, %2" : "=v"(k23) : "v"(key[2]), "v"(key[3]));
        const float4 sa = lds_rank[(unsigned)__double2loint(k01)], sb = lds_rank[(unsigned)__double2loint(k23)];
        asm("v_max_f64 %0, %1, %2" : "=v"(kw) : "v"(k01), "v"(k23));
        const bool first = __double2loint(kw) == __double2loint(k01) && __double2hiint(kw) == __double2hiint(k01);
        const float4 s = first ? sa : sb;
        vstar = __int_as_float(__double2hiint(kw));
#else
#pragma unroll
        for (int st = 1; st < W; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < W; i += 2 * st)
                asm("v_max_f64 %0, %1, %2" : "=v"(key[i]) : "v"(key[i]), "v"(key[i + st]));
        const unsigned win = (unsigned)__double2loint(key[0]);
        vstar = __int_as_float(__double2hiint(key[0]));
        const float4 s = lds_rank[win];                // same address in every lane: LDS broadcast
#endif
        sxy.x = s.x; sxy.y = s.y; syy.x = s.y; szk.x = s.z; szk.y = s.w;
        kprev = __float_as_int(s.w);
    };
    int j = 1;
#if PN2_PR_EARLY_ALL > 0
    {
        const int je = m - 1 < (PN2_PR_EARLY_ALL) ? m - 1 : (PN2_PR_EARLY_ALL);       // even count: the parities below stay aligned
        for (; j + 1 <= (je & ~1); j += 2) {
            round(j, 1, std::true_type());
            round(j + 1, 0, std::true_type());
        }
    }
#endif
    for (; j + 1 < m; j += 2) {
        round(j, 1, std::false_type());
        round(j + 1, 0, std::false_type());
    }
    if (j < m) round(j, 1, std::false_type());
    if (t == 0 && m > 1) {                              // the last round's index
        dst[m - 1] = kprev;
        if (PUBLISH)
            __hip_atomic_store(gtag + (m - 1), ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)kprev, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    fps_gather_epilogue<T>(m, src, dst, dxyz);
}

}  // namespace pn2
