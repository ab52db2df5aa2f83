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
// points are dealt to the slots SPATIALLY: a balanced kd-tree with G = 32 or 64 leaves is built once per cloud in LDS
// (histogram medians, below), leaf g becomes GROUP (wave g % 4, group g / 4 of that wave) = GS = 2 or 4 slots of all 64
// lanes, and every group keeps the tight bounding box of its 64 * GS points. Per round, lane l of every wave tests
// group l's box against the new sample (16 vector instructions for all groups at once); a group whose box lies
// farther from the sample than sqrt(v*) cannot change and is SKIPPED, a wave none of whose groups is touched skips
// its whole reduction and re-publishes its cached wave key. The tie rule is untouched: a slot's key is still
// (value bits : NS - 1 - rank) with rank = (k mod 512) * ceil(n / 512) + k / 512, it just lives in a register loaded
// once instead of being derived from the thread number, and the winner's (x, y, z, k) still comes from the LDS mirror
// kept in rank order.
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
// kd-tree. L = log2(G) levels; at every level each segment is cut at the median of its widest axis: 64-bin histogram
// along that axis (LDS atomics), the bin holding the median found by a wave prefix sum, points below / above go left /
// right, the points IN the median bin are dealt out by an LDS ticket so that both halves have exactly the same size.
// Which of them go where is timing dependent -- and irrelevant: any grouping gives the same samples, grouping only
// decides how much is skipped. Points never move during the build (they sit in registers with a segment number); one
// final scatter through LDS puts them into their slots.
#pragma once
#include "fps_body.h"

#include <math.h>

namespace pn2 {

constexpr int kPrT = 256;              // threads of the pruned tier
constexpr int kPrW = kPrT / PN2_WAVE;  // 4 waves: one per SIMD
constexpr int kPrBins = 64;            // histogram bins per median search (= one wave)

// LDS layout (bytes): [0,64) wave keys (2 parities x 4) | [64,256) reduction scratch | mirror / staging 16 * NS |
// hist G * 64 ints | segtab G float4 | split G int2 | cursor G ints | leafcur G ints | cell 2 x G x 8 floats | gbox G x 8 floats
__host__ __device__ constexpr size_t fps_pruned_lds_bytes(int P, int G)
{
    return 256 + (size_t)16 * kPrT * P + (size_t)G * (kPrBins * 4 + 16 + 8 + 4 + 4 + 2 * 32 + 32);
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

// P rank slots per thread, GS slots per group (both powers of two, GS even: the update runs on register pairs).
template <int P, int GS, bool PUBLISH>
__device__ __forceinline__ void fps_pruned_body(int n, int m, int Q, int cloud, const float *__restrict__ xyz,
                                                int *__restrict__ out, float *__restrict__ out_xyz,
                                                unsigned long long *__restrict__ tagged, char *smem, unsigned tag = 1u)
{
    constexpr int T = kPrT, W = kPrW, NS = T * P;
    constexpr int GW = P / GS;                    // groups per wave
    constexpr int G = W * GW;                     // groups = kd leaves = test lanes
    static_assert(G <= 64 && GS >= 2 && (GS & 1) == 0 && (G & (G - 1)) == 0, "group geometry");
    constexpr int LV = G == 64 ? 6 : G == 32 ? 5 : G == 16 ? 4 : 3;
    static_assert((1 << LV) == G, "G must be 8, 16, 32 or 64");

    unsigned long long *partial = reinterpret_cast<unsigned long long *>(smem);        // [2][W]
    float *scratch = reinterpret_cast<float *>(smem + 64);                              // 48 floats
    float4 *lds_rank = reinterpret_cast<float4 *>(smem + 256);                          // mirror [NS], first the staging copy
    char *tab = smem + 256 + (size_t)16 * NS;
    int *hist = reinterpret_cast<int *>(tab);                                           // [G][64]
    float4 *segtab = reinterpret_cast<float4 *>(tab + (size_t)G * kPrBins * 4);         // [G] {axis, lo, scale, bin width}
    int2 *split = reinterpret_cast<int2 *>(reinterpret_cast<char *>(segtab) + (size_t)G * 16);   // [G] {median bin, tickets that go left}
    int *cursor = reinterpret_cast<int *>(reinterpret_cast<char *>(split) + (size_t)G * 8);      // [G]
    int *leafcur = cursor + G;                                                          // [G]
    float *cell = reinterpret_cast<float *>(leafcur + G);                               // [2][G][8]: lo xyz, -, hi xyz, -
    float *gbox = cell + 2 * G * 8;                                                     // [G][8]

    const float *__restrict__ src = xyz + (size_t)cloud * n * 3;
    int *__restrict__ dst = out + (size_t)cloud * m;
    float *__restrict__ dxyz = out_xyz ? out_xyz + (size_t)cloud * m * 3 : nullptr;
    pn2_gu64 *gtag = PUBLISH ? (pn2_gu64 *)(tagged + (size_t)cloud * m) : nullptr;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);

    // ---- the items in natural order: item i = point i, or (i >= n) a padding item at point 0's position ----------------
    float px[P], py[P], pz[P];
    int seg[P], bin[P];
    float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int i = t + j * T;
        const int k = i < n ? i : 0;
        px[j] = src[(size_t)k * 3 + 0]; py[j] = src[(size_t)k * 3 + 1]; pz[j] = src[(size_t)k * 3 + 2];
        seg[j] = 0;
        lx = fminf(lx, px[j]); ly = fminf(ly, py[j]); lz = fminf(lz, pz[j]);
        hx = fmaxf(hx, px[j]); hy = fmaxf(hy, py[j]); hz = fmaxf(hz, pz[j]);
    }
    lx = pr_minmax_lane63<false>(lx); ly = pr_minmax_lane63<false>(ly); lz = pr_minmax_lane63<false>(lz);
    hx = pr_minmax_lane63<true>(hx); hy = pr_minmax_lane63<true>(hy); hz = pr_minmax_lane63<true>(hz);
    if (lane == 63) {
        scratch[w * 8 + 0] = lx; scratch[w * 8 + 1] = ly; scratch[w * 8 + 2] = lz;
        scratch[w * 8 + 4] = hx; scratch[w * 8 + 5] = hy; scratch[w * 8 + 6] = hz;
    }
    for (int i = t; i < G; i += T) leafcur[i] = 0;
    __syncthreads();
    if (t < 8 && (t & 3) != 3) {                   // root cell = the cloud's bounding box
        float v = scratch[t];
#pragma unroll
        for (int ww = 1; ww < W; ++ww) v = t < 4 ? fminf(v, scratch[ww * 8 + t]) : fmaxf(v, scratch[ww * 8 + t]);
        cell[t] = v;
    }
    __syncthreads();

    // ---- balanced kd-tree, one level per iteration ---------------------------------------------------------------------
    for (int lev = 0; lev < LV; ++lev) {
        const int nseg = 1 << lev;
        float *cur = cell + (lev & 1) * G * 8, *nxt = cell + ((lev & 1) ^ 1) * G * 8;
        if (t < nseg) {                                  // the segment's split axis: the widest side of its cell
            const float *c = cur + t * 8;
            const float ex = c[4] - c[0], ey = c[5] - c[1], ez = c[6] - c[2];
            int a = 0; float e = ex, lo = c[0];
            if (ey > e) { a = 1; e = ey; lo = c[1]; }
            if (ez > e) { a = 2; e = ez; lo = c[2]; }
            const bool ok = e > 0.0f && e < INFINITY;    // degenerate / non-finite cell: every point in bin 0, tickets split it
            segtab[t] = make_float4(__int_as_float(a), ok ? lo : 0.0f, ok ? (float)kPrBins / e : 0.0f, ok ? e * (1.0f / kPrBins) : 0.0f);
        }
        for (int i = t; i < nseg * kPrBins; i += T) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const float4 st = segtab[seg[j]];
            const int a = __float_as_int(st.x);
            const float c = a == 0 ? px[j] : a == 1 ? py[j] : pz[j];
            int b = (int)((c - st.y) * st.z);            // NaN -> 0
            b = b < 0 ? 0 : b > kPrBins - 1 ? kPrBins - 1 : b;
            bin[j] = b;
            atomicAdd(&hist[seg[j] * kPrBins + b], 1);
        }
        __syncthreads();
        for (int s = w; s < nseg; s += W) {              // one wave per segment: the bin that holds the median
            const int h = hist[s * kPrBins + lane];
            const int incl = pr_prefix_sum_incl(h);
            const int half = (NS >> lev) >> 1;
            const unsigned long long ge = __ballot(incl >= half);
            const int mb = __builtin_ctzll(ge);          // exists: incl of lane 63 = the segment's size >= half
            const int incl_mb = __builtin_amdgcn_readlane(incl, mb), h_mb = __builtin_amdgcn_readlane(h, mb);
            if (lane == 0) {
                split[s] = make_int2(mb, half - (incl_mb - h_mb));       // tickets 0 .. k-1 of the median bin go left
                cursor[s] = 0;
                const float4 st = segtab[s];
                const int a = __float_as_int(st.x);
                const float *c = cur + s * 8;
                float *l = nxt + (2 * s) * 8, *r = nxt + (2 * s + 1) * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) { l[q] = c[q]; r[q] = c[q]; }
                if (st.z > 0.0f) {                        // children overlap by the median bin
                    l[4 + a] = st.y + (float)(mb + 1) * st.w;
                    r[a] = st.y + (float)mb * st.w;
                }
            }
        }
        __syncthreads();
        int tk[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int2 sp = split[seg[j]];
            tk[j] = 0;
            if (bin[j] == sp.x) tk[j] = atomicAdd(&cursor[seg[j]], 1) - sp.y;      // >= 0: right
            else if (bin[j] > sp.x) tk[j] = 0;
            else tk[j] = -1;
        }
#pragma unroll
        for (int j = 0; j < P; ++j) seg[j] = 2 * seg[j] + (tk[j] >= 0 ? 1 : 0);
        // the next level rewrites hist / segtab behind its own barrier and split / cursor two barriers from here
    }

    // ---- scatter into the slots: leaf g -> wave g % W, group g / W; ticket r of the leaf -> lane r % 64, slot r / 64 -------
    {
        int r[P];
#pragma unroll
        for (int j = 0; j < P; ++j) r[j] = atomicAdd(&leafcur[seg[j]], 1);
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int g = seg[j];
            const int dt = (g & (W - 1)) * 64 + (r[j] & 63);
            const int dp = (g / W) * GS + (r[j] >> 6);
            lds_rank[dt * P + dp] = make_float4(px[j], py[j], pz[j], __int_as_float(t + j * T));
        }
    }
    __syncthreads();
    pn2_f2 xx[P / 2], yy[P / 2], zz[P / 2];
    float md[P];
    unsigned low[P];
    int kk[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float4 v = lds_rank[t * P + p];
        const int i = __float_as_int(v.w);
        const bool real = i < n;
        const int rank = (i & (kRefThreads - 1)) * Q + (i >> 9);      // tie rank of point i (fps_body.h)
        kk[p] = real ? i : 0;
        low[p] = real ? (unsigned)(NS - 1 - rank) : 0u;                // padding: value 0, lowest key (never wins over rank 0)
        md[p] = real ? 1e38f : 0.0f;                                   // tf_sampling_g.cu:118
        if (p & 1) { xx[p / 2].y = v.x; yy[p / 2].y = v.y; zz[p / 2].y = v.z; }
        else { xx[p / 2].x = v.x; yy[p / 2].x = v.y; zz[p / 2].x = v.z; }
    }
    __syncthreads();                                    // everybody has read the staging copy: the region becomes the mirror
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float X = (p & 1) ? xx[p / 2].y : xx[p / 2].x, Y = (p & 1) ? yy[p / 2].y : yy[p / 2].x,
                    Z = (p & 1) ? zz[p / 2].y : zz[p / 2].x;
        if (md[p] != 0.0f) lds_rank[low[p]] = make_float4(X, Y, Z, __int_as_float(kk[p]));
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
    float blx, bly, blz, bhx, bhy, bhz;
    {
        const float *o = gbox + (lane & (G - 1)) * 8;
        blx = o[0]; bly = o[1]; blz = o[2]; bhx = o[4]; bhy = o[5]; bhz = o[6];
    }

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
    double gk[GW];                                      // lane-level key of every group (cached across rounds)
#pragma unroll
    for (int gi = 0; gi < GW; ++gi) gk[gi] = 0.0;
    double wave_key = 0.0;                              // lane 63: the wave's key (cached while no group of the wave changes)

    auto round = [&](const int j, const int par) __attribute__((always_inline)) {
        // which groups can the new sample change? (all lanes, lane l = group l)
        const float thr = __fadd_rn(__fmul_rn(vstar, 1.00001f), 1e-30f);
        const float ax = __builtin_fmaxf(__builtin_fmaxf(__fsub_rn(blx, sxy.x), __fsub_rn(sxy.x, bhx)), 0.0f);
        const float ay = __builtin_fmaxf(__builtin_fmaxf(__fsub_rn(bly, sxy.y), __fsub_rn(sxy.y, bhy)), 0.0f);
        const float az = __builtin_fmaxf(__builtin_fmaxf(__fsub_rn(blz, szk.x), __fsub_rn(szk.x, bhz)), 0.0f);
        const float bd = __fadd_rn(__fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay)), __fmul_rn(az, az));
        const unsigned long long far_mask = __ballot(bd >= thr);          // NaN -> not far -> updated
        const unsigned mybits = (unsigned)(~far_mask >> (w * GW)) & ((1u << GW) - 1u);   // scalar
        unsigned long long *slot = partial + par * W;
        if (mybits != 0u) {                                               // wave-uniform
#pragma unroll
            for (int gi = 0; gi < GW; ++gi) {
                if (mybits & (1u << gi)) {                                // wave-uniform
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
                }
            }
            double kt[GW];
#pragma unroll
            for (int gi = 0; gi < GW; ++gi) kt[gi] = gk[gi];
#pragma unroll
            for (int st = 1; st < GW; st <<= 1)
#pragma unroll
                for (int i = 0; i + st < GW; i += 2 * st)
                    asm("v_max_f64 %0, %1, %2" : "=v"(kt[i]) : "v"(kt[i]), "v"(kt[i + st]));
            wave_key = wave_max_f64_lane63(kt[0]);
        }
        if (lane == 63) reinterpret_cast<double *>(slot)[w] = wave_key;
        __syncthreads();
        const double *dslot = reinterpret_cast<const double *>(slot);
        double key[W];
#pragma unroll
        for (int i = 0; i < W; ++i) key[i] = dslot[i];
#pragma unroll
        for (int st = 1; st < W; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < W; i += 2 * st)
                asm("v_max_f64 %0, %1, %2" : "=v"(key[i]) : "v"(key[i]), "v"(key[i + st]));
        const unsigned win = (unsigned)__double2loint(key[0]);
        vstar = __int_as_float(__double2hiint(key[0]));
        const float4 s = lds_rank[win];                // same address in every lane: LDS broadcast
        sxy.x = s.x; sxy.y = s.y; syy.x = s.y; szk.x = s.z; szk.y = s.w;
        if (t == 0) {
            const int k = __float_as_int(s.w);
            dst[j] = k;
            if (PUBLISH)
                __hip_atomic_store(gtag + j, ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)k, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    int j = 1;
    for (; j + 1 < m; j += 2) {
        round(j, 1);
        round(j + 1, 0);
    }
    if (j < m) round(j, 1);
    fps_gather_epilogue<T>(m, src, dst, dxyz);
}

}  // namespace pn2
