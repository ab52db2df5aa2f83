// topk.hip -- selection_sort (select_top_k) for gfx950.
//
// Replaces selection_sort_gpu / selectionSortLauncher (reference
// tf_ops/grouping/tf_grouping_g.cu:83-123, :129-132; CPU twin
// test/selection_sort.cpp:20-63). The op's contract is the FULL (b,m,n) pair
// of outputs: a copy of each distance row and an iota, on which k rounds of
// "swap the first minimum of the unsorted tail into slot s" have been applied.
// The swaps permute the tail, so the whole row -- not just its first k
// entries -- is reproduced exactly.
//
// Design. The reference gives a row to ONE thread (O(k*n) serial global-memory
// traffic per row). Here a 64-lane wave owns a row held in LDS: each round every
// lane scans a strided slice of the tail keeping its first minimum, the wave
// reduces (value, index) keys with a 64-bit min, and lane 0 swaps. Rows longer
// than the LDS tier fall back to a one-thread-per-row kernel.
#include "pn2_device.h"

#include <limits.h>

namespace pn2 {

constexpr int kSortMaxLdsN = 16384;   // 8 B per element -> 128 KiB

// map a float to a uint whose unsigned order equals the float `<` order;
// +0 and -0 compare equal under `<`, so both map to the same key.
__device__ __forceinline__ unsigned orderable(float f)
{
    if (f == 0.0f) f = 0.0f;   // folds -0 into +0
    const unsigned u = (unsigned)__float_as_int(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(64) void selection_sort_wave_kernel(int n, int k, const float *__restrict__ dist,
                                                                 int *__restrict__ outi, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *val = reinterpret_cast<float *>(smem);
    int *ind = reinterpret_cast<int *>(smem + sizeof(float) * (size_t)n);
    const size_t row = blockIdx.x;
    const int lane = threadIdx.x;
    const float *src = dist + row * n;
    for (int s = lane; s < n; s += 64) { val[s] = src[s]; ind[s] = s; }
    __syncthreads();
    const int rounds = min(k, n);
    for (int s = 0; s < rounds; ++s) {
        // first minimum of val[s..n): key = (orderable value, position), 64-bit min
        unsigned long long best = ~0ull;
        for (int t = s + lane; t < n; t += 64) {
            const unsigned long long key = ((unsigned long long)orderable(val[t]) << 32) | (unsigned)t;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)best, o, 64);
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), o, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            best = other < best ? other : best;
        }
        const int mn = (int)(unsigned)best;
        if (lane == 0 && mn != s) {
            const float tv = val[mn]; val[mn] = val[s]; val[s] = tv;
            const int ti = ind[mn]; ind[mn] = ind[s]; ind[s] = ti;
        }
        __syncthreads();
    }
    for (int s = lane; s < n; s += 64) { out[row * n + s] = val[s]; outi[row * n + s] = ind[s]; }
}

// knn_point without the (b, m, n) tensors. The reference builds the pairwise
// squared-distance matrix in TF (tf_grouping.py:57-65: tile, subtract, square, reduce_sum -- three
// (b,m,n,c)/(b,m,n) tensors, ~0.5 GB each at the metric shape), runs the selection sort above on it and
// slices the first k columns. Here a wave computes its query's distance row straight into LDS
// (((dx*dx)+(dy*dy))+(dz*dz), d = xyz1 - xyz2, no FMA: the values the elementwise graph produces) and
// reproduces the first k outputs of the swap rounds -- ties come out in the reference's swap-dependent
// order, not in index order -- WITHOUT scanning the whole row k times:
//   * only elements with value <= v_k (the k-th smallest) can ever be selected, and a swap only moves
//     the element sitting at slot s to the slot of the selected one. So the rounds can be replayed on the
//     candidate set C = {value <= tau} for any tau >= v_k, tracking each candidate's CURRENT slot: round s
//     picks the candidate with the smallest (value, current slot); if another candidate sits at slot s it
//     inherits the winner's slot. Non-candidates move around too, but nothing ever looks at them.
//   * tau: for k <= 40 the k-th smallest of the 64 per-lane minima (registers only); beyond, two 256-bin
//     histogram passes over the row (top 16 bits of the order-preserving key, v_k rounded up by < 0.4 %).
//     Either way |C| is k plus a handful on scattered data.
//   * |C| > kKnnCap (clouds of identical points) falls back to the literal rounds on the row.
constexpr int kKnnCap = 1024;

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o, 64);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o, 64);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        v = other < v ? other : v;
    }
    return v;
}

// wave-wide minimum of a positive double (+inf allowed) in VALU only (DPP row operations + v_min_f64), the
// result broadcast to every lane: the ds_bpermute butterfly this replaces cost ~1200 cycles per round
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double knn_dpp_min_step(double v)
{
    const int hi = __double2hiint(v), lo = __double2loint(v);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);   // no source: keep own value
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const double o = __hiloint2double(ohi, olo);
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(v), "v"(o));
    return r;
}
__device__ __forceinline__ double knn_wave_min_f64(double v)
{
    v = knn_dpp_min_step<0xB1, 0xf>(v);    // quad_perm:[1,0,3,2]
    v = knn_dpp_min_step<0x4E, 0xf>(v);    // quad_perm:[2,3,0,1]
    v = knn_dpp_min_step<0x141, 0xf>(v);   // row_half_mirror
    v = knn_dpp_min_step<0x140, 0xf>(v);   // row_mirror
    v = knn_dpp_min_step<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v = knn_dpp_min_step<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3: lane 63 holds the minimum
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// the 256-bin histogram in `hist` (lane l owns bins 4l..4l+3): bin in which the cumulative count reaches
// `want` (1-based), and the count before that bin
__device__ __forceinline__ void knn_find_bin(const int *hist, int lane, int want, int &bin, int &before)
{
    const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const int mine = h0 + h1 + h2 + h3;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const int excl = incl - mine;
    const bool here = excl < want && want <= incl;               // exactly one lane (want <= total)
    int b = 0, bef = excl;
    if (here) {
        if (want <= excl + h0) { b = 0; }
        else if (want <= excl + h0 + h1) { b = 1; bef = excl + h0; }
        else if (want <= excl + h0 + h1 + h2) { b = 2; bef = excl + h0 + h1; }
        else { b = 3; bef = excl + h0 + h1 + h2; }
    }
    const unsigned long long m = __ballot(here);
    const int src = m ? __builtin_ctzll(m) : 0;
    bin = __shfl(4 * lane + b, src);
    before = __shfl(bef, src);
}

__global__ __launch_bounds__(64) void knn_wave_kernel(int n, int m, int k, const float *__restrict__ xyz1,
                                                      const float *__restrict__ xyz2, float *__restrict__ oval,
                                                      int *__restrict__ oidx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *val = reinterpret_cast<float *>(smem);                                  // [n]  the distance row
    int *ind = reinterpret_cast<int *>(smem + sizeof(float) * (size_t)n);          // [n]  only for the fallback
    int *hist = ind + n;                                                           // [256]
    unsigned *ckey = reinterpret_cast<unsigned *>(hist + 256);                     // [kKnnCap] order-preserving value key
    int *cpos = reinterpret_cast<int *>(ckey + kKnnCap);                           // [kKnnCap] current slot
    int *cidx = cpos + kKnnCap;                                                    // [kKnnCap] original index
    const size_t row = blockIdx.x;                               // query number over all clouds
    const size_t cloud = row / m;
    const int lane = threadIdx.x;
    const float *pts = xyz1 + cloud * n * 3;
    const float qx = xyz2[row * 3 + 0], qy = xyz2[row * 3 + 1], qz = xyz2[row * 3 + 2];
    unsigned lmin = 0xffffffffu;                                  // smallest key among this lane's elements
    for (int s = lane; s < n; s += 64) {
        const float d = sqdist_key(pts[s * 3 + 0], pts[s * 3 + 1], pts[s * 3 + 2], qx, qy, qz);   // NaN sign cleared (pn2_device.h)
        val[s] = d;
        lmin = min(lmin, orderable(d));
    }
    const int rounds = min(k, n);

    // ---- tau >= v_k ---------------------------------------------------------------------------------------
    unsigned tau;
    if (rounds <= 40) {
        // the `rounds`-th smallest of the 64 per-lane minima: that many DISTINCT elements are <= it, so it
        // bounds v_k, and on scattered data it sits at global rank ~ -64 ln(1 - rounds/64) (44 for 32 of
        // 4096). No LDS traffic at all: the histogram passes below spend ~100 k cycles per query in
        // same-address LDS atomics (distances share a handful of exponents).
        int rank = 0;
        for (int j = 0; j < 64; ++j) {
            const unsigned other = (unsigned)__builtin_amdgcn_readlane((int)lmin, j);
            rank += (other < lmin || (other == lmin && j < lane)) ? 1 : 0;
        }
        const unsigned long long mask = __ballot(rank == rounds - 1);            // exactly one lane
        tau = (unsigned)__builtin_amdgcn_readlane((int)lmin, __builtin_ctzll(mask));
    } else {
    // upper edge of the 16-bit key prefix that holds the k-th smallest value
    for (int i = lane; i < 256; i += 64) hist[i] = 0;
    __syncthreads();
    for (int s = lane; s < n; s += 64) atomicAdd(&hist[orderable(val[s]) >> 24], 1);
    __syncthreads();
    int b1, before1;
    knn_find_bin(hist, lane, rounds, b1, before1);
    __syncthreads();
    for (int i = lane; i < 256; i += 64) hist[i] = 0;
    __syncthreads();
    for (int s = lane; s < n; s += 64) {
        const unsigned key = orderable(val[s]);
        if ((int)(key >> 24) == b1) atomicAdd(&hist[(key >> 16) & 255], 1);
    }
    __syncthreads();
    int b2, before2;
    knn_find_bin(hist, lane, rounds - before1, b2, before2);
    tau = ((unsigned)b1 << 24) | ((unsigned)b2 << 16) | 0xffffu;
    }

    // ---- candidates, in ascending slot order --------------------------------------------------------------
    int total = 0;
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const unsigned key = s < n ? orderable(val[s]) : 0xffffffffu;
        const bool in = s < n && key <= tau;
        const unsigned long long mask = __ballot(in);
        const int slot = total + __popcll(mask & ((1ull << lane) - 1ull));
        if (in && slot < kKnnCap) { ckey[slot] = __float_as_uint(val[s] == 0.0f ? 0.0f : val[s]); cpos[slot] = s; cidx[slot] = s; }
        total += __popcll(mask);
    }
    __syncthreads();

    if (total <= kKnnCap) {
        // ---- the swap rounds replayed on the candidates ----------------------------------------------------
        // keys (raw value bits : current slot) read as doubles: distances are >= +0, so the bit patterns
        // order like the values, stay below the fp64 exponent of Inf (0x7ff00000 > 0x7f800000), and a dead
        // candidate is +Inf -- the arg-min is a v_min_f64 chain, as in the FPS kernel
        const double dead = __hiloint2double(0x7ff00000, 0);
        for (int s = 0; s < rounds; ++s) {
            double best = dead;
            int bslot = 0;
            for (int c = lane; c < total; c += 64) {
                const double key = __hiloint2double((int)ckey[c], cpos[c]);
                if (key < best) { best = key; bslot = c; }
            }
            const double win = knn_wave_min_f64(best);            // (value, current slot): unique among live candidates
            const int q = __double2loint(win);                    // the winner's current slot
            if (best == win && best < dead) {                     // exactly one lane holds the winner
                const int orig = cidx[bslot];
                oval[row * k + s] = val[orig];
                oidx[row * k + s] = orig;
                ckey[bslot] = 0x7ff00000u;                        // dead: never wins again
                cpos[bslot] = 0;
            }
            __syncthreads();
            if (q != s) {                                         // the element sitting at slot s moves to slot q
                for (int c = lane; c < total; c += 64)
                    if (cpos[c] == s && ckey[c] != 0x7ff00000u) cpos[c] = q;
            }
            __syncthreads();
        }
    } else {
        // ---- too many ties: the literal rounds on the whole row --------------------------------------------
        for (int s = lane; s < n; s += 64) ind[s] = s;
        __syncthreads();
        for (int s = 0; s < rounds; ++s) {
            unsigned long long best = ~0ull;
            for (int t = s + lane; t < n; t += 64) {
                const unsigned long long key = ((unsigned long long)orderable(val[t]) << 32) | (unsigned)t;
                best = key < best ? key : best;
            }
            const int mn = (int)(unsigned)wave_min_u64(best);
            if (lane == 0 && mn != s) {
                const float tv = val[mn]; val[mn] = val[s]; val[s] = tv;
                const int ti = ind[mn]; ind[mn] = ind[s]; ind[s] = ti;
            }
            __syncthreads();
        }
        for (int s = lane; s < rounds; s += 64) {
            oval[row * k + s] = val[s];
            oidx[row * k + s] = ind[s];
        }
    }
}

// fallback for very long rows: the reference's own mapping (one thread per row)
__global__ __launch_bounds__(64) void selection_sort_serial_kernel(long long rows, int n, int k,
                                                                   const float *__restrict__ dist,
                                                                   int *__restrict__ outi, float *__restrict__ out)
{
    const long long row = (long long)blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const float *src = dist + row * n;
    float *v = out + row * n;
    int *vi = outi + row * n;
    for (int s = 0; s < n; ++s) { v[s] = src[s]; vi[s] = s; }
    for (int s = 0; s < k && s < n; ++s) {
        int mn = s;
        float mv = v[s];
        for (int t = s + 1; t < n; ++t) {
            const float c = v[t];
            if (c < mv) { mv = c; mn = t; }
        }
        if (mn != s) {
            v[mn] = v[s]; v[s] = mv;
            const int ti = vi[mn]; vi[mn] = vi[s]; vi[s] = ti;
        }
    }
}

}  // namespace pn2

extern "C" int pn2_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream)
{
    using namespace pn2;
    if (k <= 0) return PN2_E_ARG;                       // tf_grouping.cpp:113
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    const long long rows = (long long)b * m;
    if (rows == 0) return PN2_OK;
    if (!dist || !outi || !out) return PN2_E_NULL;
    if (rows > INT_MAX) return PN2_E_TOO_LARGE;
    hipStream_t st = as_stream(stream);
    if (n <= kSortMaxLdsN) {
        const size_t lds = 8 * (size_t)n;
        auto kern = selection_sort_wave_kernel;
        if (int rc = allow_dynamic_lds(kern, lds)) return rc;
        if (int rc = launch(kern, dim3((unsigned)rows), dim3(64), lds, st, n, k, dist, outi, out)) return rc;
    } else {
        if (int rc = launch(selection_sort_serial_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, rows, n,
                           k, dist, outi, out)) return rc;
    }
    return PN2_OK;
}

extern "C" int pn2_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                             void *stream)
{
    using namespace pn2;
    if (k <= 0) return PN2_E_ARG;                       // tf_grouping.cpp:113
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    const long long rows = (long long)b * m;
    if (rows == 0) return PN2_OK;
    if (!xyz1 || !xyz2 || !val || !idx) return PN2_E_NULL;
    if (rows > INT_MAX || k > n) return PN2_E_TOO_LARGE;   // k > n: tf.slice would fail in the reference as well
    if (n > kSortMaxLdsN - 2048) return PN2_E_TOO_LARGE;    // callers keep the matrix + pn2_selection_sort path
    const size_t lds = 8 * (size_t)n + sizeof(int) * (256 + 3 * (size_t)kKnnCap);
    auto kern = knn_wave_kernel;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    if (int rc = launch(kern, dim3((unsigned)rows), dim3(64), lds, as_stream(stream), n, m, k, xyz1, xyz2, val, idx)) return rc;
    return PN2_OK;
}
