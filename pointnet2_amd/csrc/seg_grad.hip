// seg_grad.hip -- gradients of group_point / three_interpolate as a SEGMENTED REDUCTION instead of a
// scatter of fp32 atomics. gfx950.
//
// The reference kernels (tf_grouping_g.cu:60-78; tf_interpolate.cpp:131-153 is the CPU twin of the same
// sum) add every element of grad_out into grad_points[idx] with an atomic. Measured on MI355X the
// atomic scatter runs at ~1.3 TB/s of algorithmic traffic (16 % of HBM peak) whatever the channel
// count, because the L2 atomic units, not the memory, are the limit. Here the index tensor is inverted
// first (a counting sort of the b*m*nsample references by target point: int atomics, independent of
// the channel count), then ONE lane group owns each output row and sums the grad_out rows that refer to
// it: coalesced row reads, every output element written exactly once, no float atomics, no zero-fill.
// Worth it from ~16 channels up (the inversion costs about as much as the atomic scatter of 8 channels).
//
// deterministic = 1: the order in which the counting sort fills a segment varies from run to run, so a
// plain fp32 sum would too. The reproducible mode sums each output element in 64-bit FIXED POINT with a
// scale chosen PER ELEMENT from the largest addend of its own segment (two passes over the segment, the
// second one out of L2): integer sums and maxima do not depend on the order, so the result is identical
// on every run, and it is within one rounding of the exact sum relative to the element's own largest
// addend (the atomics-based pn2_*_grad_det scale by the largest addend of the whole tensor).
#include "pn2_device.h"

#include <limits.h>
#include <math.h>

#include <algorithm>
#include <type_traits>

namespace pn2 {

constexpr int kSegThreads = 256;

static inline unsigned seg_grid(long long work, int per_block = kSegThreads)
{
    long long g = (work + per_block - 1) / per_block;
    if (g > 256 * 32) g = 256 * 32;
    return (unsigned)(g > 0 ? g : 1);
}

// workspace: start int[b * (rows + 1)] | cursor int[b * rows] | sorted int[b * rows] | list int[b * entries]
// sorted[row] != 0: the row's segment lists its entries in ascending order
struct SegWs {
    int *start, *cursor, *sorted, *list;
};
static inline SegWs seg_ws(void *ws, int b, int rows, long long entries)
{
    int *p = reinterpret_cast<int *>(ws);
    int *cursor = p + (size_t)b * (rows + 1);
    return {p, cursor, cursor + (size_t)b * rows, cursor + 2 * (size_t)b * rows};
}

__global__ __launch_bounds__(kSegThreads) void seg_count_kernel(long long total, long long entries, int rows,
                                                                const int *__restrict__ idx, int *__restrict__ cnt)
{
    for (long long e = (long long)blockIdx.x * kSegThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kSegThreads)
        atomicAdd(cnt + (e / entries) * rows + idx[e], 1);
}

// one workgroup per cloud: exclusive scan of the counts -> start[0..rows], cursor = start
__global__ __launch_bounds__(1024) void seg_scan_kernel(int rows, int *__restrict__ start, int *__restrict__ cursor)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    int *cnt = cursor + (size_t)blockIdx.x * rows;              // counts were accumulated in the cursor array
    int *st = start + (size_t)blockIdx.x * (rows + 1);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < rows; base += 1024) {
        const int r = base + t;
        const int v = r < rows ? cnt[r] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int off = carry_s;
        for (int i = 0; i < w; ++i) off += wsum[i];
        if (r < rows) {
            st[r] = off + incl - v;
            cnt[r] = off + incl - v;
        }
        __syncthreads();
        if (t == 1023) carry_s = off + incl;
        __syncthreads();
    }
    if (t == 0) st[rows] = carry_s;
}

__global__ __launch_bounds__(kSegThreads) void seg_fill_kernel(long long total, long long entries, int rows,
                                                               const int *__restrict__ idx, int *__restrict__ cursor,
                                                               int *__restrict__ list)
{
    for (long long e = (long long)blockIdx.x * kSegThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kSegThreads) {
        const long long i = e / entries;
        const int pos = atomicAdd(cursor + i * rows + idx[e], 1);
        list[i * entries + pos] = (int)(e - i * entries);
    }
}

// The whole inversion of one cloud in ONE workgroup with LDS counters (count, scan, fill): scattered
// 4-byte atomics at the L2 run at ~12 G/s on this chip (85 us for the 1 M references of the metric
// shape, per pass), LDS atomics do the same in a few microseconds. rows <= kSegLdsRows.
// SORT (the list fits in LDS beside the counters): the segments are filled in LDS and every segment of
// up to kSegSortMax entries is then sorted ascending: by one thread up to kSegSortThread entries (insertion
// sort; the average segment has m*nsample/n ~ 8 entries), by one wave beyond (rank sort: ball queries
// that overflow nsample return the lowest indices, so low-numbered points collect a reference from
// almost every centroid -- segments of 100+ entries are the rule, not the exception). A sorted segment is summed in exactly the order of the reference's CPU
// loop (tf_grouping.cpp / query_ball_point.cpp:72-85, tf_interpolate.cpp:131-153: ascending entry
// number), so the fp32 result is bit-identical to it AND the same on every run, in a single pass.
constexpr int kSegLdsRows = 24576;                                // 96 KiB of counters
constexpr int kSegSortThread = 32;                                // insertion sort by one thread up to here
constexpr int kSegSortMax = 1024;                                 // rank sort by one wave up to here
// Rank sort of one segment (distinct ints, in LDS) by a group of G lanes holding Q entries each: rank =
// number of smaller entries is a permutation; a lane ranks its entries against the whole segment (LDS
// reads, the same address across the group), and only then writes them back. Every lane of the wave
// must call it (a group without work passes len = 0): the trip count is the wave's maximum.
template <int G, int Q>
__device__ __forceinline__ void seg_rank_sort(int *llist, int beg, int len, int gl)
{
    int val[Q], rank[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int a = gl + G * q;
        val[q] = a < len ? llist[beg + a] : INT_MAX;
        rank[q] = 0;
    }
    for (int j = 0; __any(j < len); ++j) {
        const int other = j < len ? llist[beg + j] : INT_MAX;
#pragma unroll
        for (int q = 0; q < Q; ++q) rank[q] += other < val[q] ? 1 : 0;
    }
    asm volatile("" ::: "memory");                               // all reads of the wave precede its writes
#pragma unroll
    for (int q = 0; q < Q; ++q)
        if (gl + G * q < len) llist[beg + rank[q]] = val[q];
}

// LDSLIST (always with SORT): the list is built in LDS and leaves the workgroup as one coalesced copy -- scattered 4-byte
// stores straight to global memory are one request per lane (24 K of them per cloud at sem_seg FP4: ~10 us of the kernel's 20).
// RUNS: lanes of a wave that hold the SAME target in a row (a ball query's padding repeats its first hit up to nsample times,
// tf_grouping_g.cu:24-31) take ONE ticket for the run -- the head lane adds the run's length, the others derive their position
// from it; 64 lanes on one counter otherwise serialise in the LDS atomic unit.
// KEEP > 0: the cloud's entries are at most KEEP x 8192 -- a thread's share (KEEP batches of eight) stays in registers between the
// counting pass and the placing pass instead of being read again.
template <bool SORT, bool LDSLIST = SORT, bool RUNS = false, int KEEP = 0>
__global__ __launch_bounds__(1024) void seg_invert_lds_kernel(long long entries, int rows, const int *__restrict__ idx,
                                                              int *__restrict__ start, int *__restrict__ sorted,
                                                              int *__restrict__ list)
{
    extern __shared__ int smem_i[];
    int *cnt = smem_i;                                            // [rows]
    int *llist = smem_i + rows;                                   // [entries] when SORT
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int *my = idx + (size_t)blockIdx.x * entries;
    int *st = start + (size_t)blockIdx.x * (rows + 1);
    int *flags = sorted + (size_t)blockIdx.x * rows;
    int *out = list + (size_t)blockIdx.x * entries;
    for (int r = t; r < rows; r += 1024) cnt[r] = 0;
    if (t == 0) carry_s = 0;
    __syncthreads();
    // (both passes over idx fetch eight entries per thread before they touch the counters: a loop of load -> LDS atomic pays
    // the global latency per entry -- 24 per thread at sem_seg FP4, where the inversion was 17 us of the gradient's 45)
    constexpr int kInvU = 8;
    // head lane of this lane's run and the run's length (valid on the head): `same` = this lane continues the previous lane's run
    auto run_of = [&](bool same, int &hl, int &runlen) __attribute__((always_inline)) {
        const unsigned long long hm = __ballot(!same);
        const unsigned long long below = lane == 63 ? hm : (hm & ((2ull << lane) - 1ull));
        hl = 63 - __clzll((long long)below);
        const unsigned long long above = lane == 63 ? 0ull : (hm >> (lane + 1));
        runlen = above ? __ffsll((unsigned long long)above) : 64 - lane;
    };
    int keep[KEEP > 0 ? KEEP : 1][kInvU];
    auto fetch = [&](long long b0, int (&v)[kInvU], bool (&ok)[kInvU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kInvU; ++u) {
            const long long e = b0 + t + (long long)u * 1024;
            ok[u] = e < entries;
            v[u] = ok[u] ? my[e] : -1;
        }
    };
    auto count = [&](const int (&v)[kInvU], const bool (&ok)[kInvU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kInvU; ++u) {
            if (RUNS) {
                const int prev = __shfl_up(v[u], 1);
                int hl, runlen;
                run_of(lane != 0 && v[u] == prev && ok[u], hl, runlen);
                if (ok[u] && hl == lane) atomicAdd(&cnt[v[u]], runlen);
            } else if (ok[u]) {
                atomicAdd(&cnt[v[u]], 1);
            }
        }
    };
    if (KEEP > 0) {
#pragma unroll
        for (int bi = 0; bi < (KEEP > 0 ? KEEP : 1); ++bi) {
            bool ok[kInvU];
            fetch((long long)bi * 1024 * kInvU, keep[bi], ok);
            count(keep[bi], ok);
        }
    } else {
        for (long long b0 = 0; b0 < entries; b0 += 1024 * kInvU) {     // (uniform trip count: the run logic shuffles across the wave)
            int v[kInvU];
            bool ok[kInvU];
            fetch(b0, v, ok);
            count(v, ok);
        }
    }
    __syncthreads();
    for (int base = 0; base < rows; base += 1024) {
        const int r = base + t;
        const int v = r < rows ? cnt[r] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int off = carry_s;
        for (int i = 0; i < w; ++i) off += wsum[i];
        if (r < rows) {
            st[r] = off + incl - v;
            cnt[r] = off + incl - v;                              // becomes the fill cursor
        }
        __syncthreads();
        if (t == 1023) carry_s = off + incl;
        __syncthreads();
    }
    if (t == 0) st[rows] = carry_s;
    auto place = [&](long long b0, const int (&v)[kInvU]) __attribute__((always_inline)) {
        int pos[kInvU];
        bool ok[kInvU];
#pragma unroll
        for (int u = 0; u < kInvU; ++u) ok[u] = b0 + t + (long long)u * 1024 < entries;
#pragma unroll
        for (int u = 0; u < kInvU; ++u) {
            if (RUNS) {
                const int prev = __shfl_up(v[u], 1);
                int hl, runlen;
                run_of(lane != 0 && v[u] == prev && ok[u], hl, runlen);
                const int base = (ok[u] && hl == lane) ? atomicAdd(&cnt[v[u]], runlen) : 0;
                pos[u] = __shfl(base, hl) + (lane - hl);
            } else {
                pos[u] = ok[u] ? atomicAdd(&cnt[v[u]], 1) : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < kInvU; ++u) {
            if (!ok[u]) continue;
            const int e = (int)(b0 + t + (long long)u * 1024);
            if (LDSLIST) llist[pos[u]] = e;
            else out[pos[u]] = e;
        }
    };
    if (KEEP > 0) {
#pragma unroll
        for (int bi = 0; bi < (KEEP > 0 ? KEEP : 1); ++bi) place((long long)bi * 1024 * kInvU, keep[bi]);
    } else {
        for (long long b0 = 0; b0 < entries; b0 += 1024 * kInvU) {
            int v[kInvU];
            bool ok[kInvU];
            fetch(b0, v, ok);
            place(b0, v);
        }
    }
    if (!SORT) {
        for (int r = t; r < rows; r += 1024) flags[r] = 0;
        if (LDSLIST) {
            __syncthreads();
            for (long long e = t; e < entries; e += 1024) out[e] = llist[e];
        }
        return;
    }
    __syncthreads();                                              // cnt[r] is now the END of segment r
    constexpr int kMid = 128, kCap = 1536;                        // 16-lane groups up to kMid entries, waves beyond
    __shared__ int nmid, nlong;
    __shared__ int wmid[kCap], wlong[kCap / 4];
    if (t == 0) { nmid = 0; nlong = 0; }
    __syncthreads();
    for (int r = t; r < rows; r += 1024) {
        const int end = cnt[r], beg = r ? cnt[r - 1] : 0;
        const int len = end - beg;
        bool done = true;
        if (len <= kSegSortThread) {
            for (int a = beg + 1; a < end; ++a) {
                const int v = llist[a];
                int p = a - 1;
                while (p >= beg && llist[p] > v) { llist[p + 1] = llist[p]; --p; }
                llist[p + 1] = v;
            }
        } else if (len <= kMid) {
            const int slot = atomicAdd(&nmid, 1);
            done = slot < kCap;
            if (done) wmid[slot] = r;
        } else if (len <= kSegSortMax) {
            const int slot = atomicAdd(&nlong, 1);
            done = slot < kCap / 4;
            if (done) wlong[slot] = r;
        } else {
            done = false;
        }
        flags[r] = done;                                          // unsorted rows take the fixed-point two-pass sum
    }
    __syncthreads();
    const int n_mid = min(nmid, kCap), n_long = min(nlong, kCap / 4);
    for (int base = 0; base < n_mid; base += 64) {               // 64 groups of 16 lanes per trip
        const int li = base + (t >> 4);
        int beg = 0, len = 0;
        if (li < n_mid) { const int r = wmid[li]; beg = r ? cnt[r - 1] : 0; len = cnt[r] - beg; }
        seg_rank_sort<16, kMid / 16>(llist, beg, len, t & 15);
    }
    for (int base = 0; base < n_long; base += 16) {              // one wave per segment
        const int li = base + w;
        int beg = 0, len = 0;
        if (li < n_long) { const int r = wlong[li]; beg = r ? cnt[r - 1] : 0; len = cnt[r] - beg; }
        seg_rank_sort<64, kSegSortMax / 64>(llist, beg, len, lane);
    }
    __syncthreads();
    for (long long e = t; e < entries; e += 1024) out[e] = llist[e];
}

// fixed-point helpers (per element): 2^ex > |m|; shift k = 62 - logcount - ex
__device__ __forceinline__ int seg_shift(float maxabs, int logcount)
{
    const unsigned bits = __float_as_uint(maxabs);
    return 62 - logcount - ((int)(bits >> 23) - 126);
}

// One lane group (LPR lanes) per output row; lanes own 4 consecutive channels (VEC4) or 1 channel.
// SRC_DIV: source row of entry e is e / SRC_DIV (1 for group_point, 3 for three_interpolate);
// weight (may be null): addend = grad_out * weight[entry], rounded to fp32 like the reference.
// The segment's entry numbers are fetched LPR at a time (one per lane) and handed round with a
// shuffle, and the rows are read four at a time: a lane that first loads an entry number and then the
// row it names pays two dependent memory round trips per entry (335 us where this form needs ~150).
// WIDE (default mode, 16-byte rows wider than LPR float4s and at most twice that: c = 320 on 64 lanes): a lane carries its SECOND
// float4 (lane + LPR) in the same sweep of the segment -- as a second sweep it repeated every dependent batch of row loads for a
// quarter of the lanes (the long-row body has done so since it exists).
template <int LPR, bool VEC4, bool DET, int SRC_DIV, int NT, bool WIDE = false>
__device__ __forceinline__ void seg_reduce_body(unsigned blk, unsigned nblk, long long out_rows, int rows, long long entries, int c,
                                                const float *__restrict__ grad_out, const float *__restrict__ weight,
                                                const int *__restrict__ start, const int *__restrict__ sorted,
                                                const int *__restrict__ list, float *__restrict__ out, int long_from)
{
    constexpr int CH = VEC4 ? 4 : 1;
    constexpr int UNR = DET ? 4 : 8;                                 // row loads in flight per lane (default mode: medium rows are chains of batches)
    const long long group = ((long long)blk * NT + threadIdx.x) / LPR;
    const int lane = threadIdx.x & 63;
    const int gl = threadIdx.x % LPR, gbase = lane - gl;            // first lane of this group inside the wave
    const long long ngroups = (long long)nblk * NT / LPR;
    const long long trips = (out_rows + ngroups - 1) / ngroups;     // wave-uniform loop: shuffles need all lanes
    for (long long trip = 0; trip < trips; ++trip) {
        const long long row_raw = group + trip * ngroups;
        const bool row_ok = row_raw < out_rows;
        const long long row = row_ok ? row_raw : out_rows - 1;
        const long long i = row / rows;
        const int r = (int)(row - i * rows);
        const int beg = start[i * (rows + 1) + r];
        int end = row_ok ? start[i * (rows + 1) + r + 1] : beg;
        // rows of long_from entries or more belong to seg_reduce_long_kernel (default mode only; 0 = every row is this kernel's)
        const bool mine_row = long_from <= 0 || end - beg < long_from;
        if (!mine_row) end = beg;
        const int *seg = list + i * entries;
        const float *src = grad_out + (size_t)i * (entries / SRC_DIV) * c;
        const float *wsrc = weight ? weight + (size_t)i * entries : nullptr;
        const int len = end - beg;
        int logc = 0;
        while ((1 << logc) < len) ++logc;
        // a sorted segment summed in order is already reproducible (and equals the reference's CPU sum):
        // the fixed-point second pass is only for long, unsorted segments
        const bool two_pass = DET && !sorted[row];

        for (int cc0 = 0; __any(cc0 < c); cc0 += (WIDE ? 2 : 1) * LPR * CH) {
            const int cc = cc0 + gl * CH;
            const bool ch_ok = cc < c;
            const int ccl = ch_ok ? cc : 0;
            const int cc2 = cc + LPR * CH;                          // WIDE: the lane's second float4
            const bool ch2_ok = WIDE && cc2 < c;
            const int ccl2 = ch2_ok ? cc2 : 0;
            float acc2[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) acc2[q] = 0.0f;
            float acc[CH];
            unsigned mx[CH];
            long long fx[CH];
            int k[CH];
            double sc[CH];                                          // 2^k: the scaling is one exact fp64 multiply
#pragma unroll
            for (int q = 0; q < CH; ++q) { acc[q] = 0.0f; mx[q] = 0u; fx[q] = 0; k[q] = 0; sc[q] = 1.0; }
            // pass 0: plain sum (and, for DET, the largest |addend|); pass 1 (DET only): fixed-point sum
            for (int pass = 0; __any(pass < (two_pass ? 2 : 1)); ++pass) {
                const bool pass_on = pass < (two_pass ? 2 : 1);
                if (DET && pass == 1) {
#pragma unroll
                    for (int q = 0; q < CH; ++q) { k[q] = seg_shift(__uint_as_float(mx[q]), logc); sc[q] = ldexp(1.0, k[q]); }
                }
                const int plen = pass_on ? len : 0;
                for (int p0 = 0; __any(p0 < plen); p0 += LPR) {
                    const int mine = (p0 + gl < plen) ? seg[beg + p0 + gl] : 0;     // this lane's entry of the chunk
                    const int chunk = min(LPR, plen - p0);
                    for (int j0 = 0; __any(j0 < chunk); j0 += UNR) {
                        float a[UNR][CH];
                        float a2[WIDE ? UNR : 1][CH];
                        float wv[UNR];
                        bool ok[UNR];
#pragma unroll
                        for (int u = 0; u < UNR; ++u) {
                            const int j = j0 + u;
                            ok[u] = j < chunk;
                            const int e = __shfl(mine, gbase + (ok[u] ? j : 0));
                            const float *g = src + (size_t)(e / SRC_DIV) * c + ccl;
                            if (VEC4) {
                                const float4 v = *reinterpret_cast<const float4 *>(g);
                                a[u][0] = v.x; a[u][1] = v.y; a[u][2] = v.z; a[u][3] = v.w;
                                if (WIDE) {
                                    const float4 v2 = *reinterpret_cast<const float4 *>(src + (size_t)(e / SRC_DIV) * c + ccl2);
                                    a2[u][0] = v2.x; a2[u][1] = v2.y; a2[u][2] = v2.z; a2[u][3] = v2.w;
                                }
                            } else {
                                a[u][0] = g[0];
                            }
                            wv[u] = wsrc ? wsrc[e] : 1.0f;
                        }
#pragma unroll
                        for (int u = 0; u < UNR; ++u) {
                            if (!ok[u]) continue;
                            if (WIDE) {
#pragma unroll
                                for (int q = 0; q < CH; ++q)
                                    acc2[q] = __fadd_rn(acc2[q], wsrc ? __fmul_rn(a2[u][q], wv[u]) : a2[u][q]);
                            }
#pragma unroll
                            for (int q = 0; q < CH; ++q) {
                                const float ad = wsrc ? __fmul_rn(a[u][q], wv[u]) : a[u][q];
                                if (!DET || pass == 0) {
                                    acc[q] = __fadd_rn(acc[q], ad);
                                    if (DET) mx[q] = max(mx[q], __float_as_uint(ad) & 0x7fffffffu);
                                } else {
                                    fx[q] += __double2ll_rn((double)ad * sc[q]);
                                }
                            }
                        }
                    }
                }
            }
            if (row_ok && ch_ok && mine_row) {
                float res[CH];
#pragma unroll
                for (int q = 0; q < CH; ++q)                        // non-finite addends: the fp32 sum propagates them
                    res[q] = (!two_pass || mx[q] >= 0x7f800000u) ? acc[q] : (float)ldexp((double)fx[q], -k[q]);
                float *o = out + row * c + cc;
                if (VEC4) *reinterpret_cast<float4 *>(o) = make_float4(res[0], res[1], res[2], res[3]);
                else o[0] = res[0];
            }
            if (WIDE && row_ok && ch2_ok && mine_row)
                *reinterpret_cast<float4 *>(out + row * c + cc2) = make_float4(acc2[0], acc2[1], acc2[2], acc2[3]);
        }
    }
}


template <int LPR, bool VEC4, bool DET, int SRC_DIV>
__global__ __launch_bounds__(kSegThreads) void seg_reduce_kernel(long long out_rows, int rows, long long entries, int c,
                                                                 const float *__restrict__ grad_out,
                                                                 const float *__restrict__ weight,
                                                                 const int *__restrict__ start,
                                                                 const int *__restrict__ sorted,
                                                                 const int *__restrict__ list,
                                                                 float *__restrict__ out, int long_from)
{
    seg_reduce_body<LPR, VEC4, DET, SRC_DIV, kSegThreads>(blockIdx.x, gridDim.x, out_rows, rows, entries, c, grad_out, weight, start, sorted,
                                                           list, out, long_from);
}

// ---- long segments (round 6) -------------------------------------------------------------------------------------------------
// A ball query that finds fewer than nsample points pads its list with the FIRST hit (tf_grouping_g.cu:24-31), and the first
// hit is the in-ball point of lowest index: low-numbered points collect a reference from almost every centroid PLUS the
// padding of every short list -- segments of hundreds to thousands of entries beside an average of 8-16. One lane group summing
// such a segment alone is a serial chain of row loads the whole launch waits for: group_point's gradient at cls_ssg L2
// (c = 128) took 104 us for 144 MB, at cls_msg L2 (c = 320, nsample 128) 741 us for 700 MB (profiles/r06). In the default
// (not bit-reproducible) mode the rows of kSegLongFrom entries or more are therefore left out by the row-per-lane-group body and summed
// here by a WHOLE workgroup each: its 512 / LPR lane groups take every (512 / LPR)-th batch of four entries, the partial rows
// meet in LDS and are added in group order (a fixed order: the result does not depend on timing, only the association differs
// from the serial sum). The reproducible mode keeps the one-group serial sum, whose order is the reference CPU loop's.
constexpr int kSegLongFrom = 64;
// ... or fewer where the average row is short: a lane group keeps four row loads in flight, so a 63-entry row is sixteen dependent
// batches -- behind the balanced long rows, those rows were the end of the launch. Twice the average row, between 32 and 64
// (cls_ssg L2, 16 per row: 38.0 -> 32.8 us with 32; sem_seg FP4's interpolation, 24 per row with little spread: 33.4 -> 32.5 with
// 48, 37.6 with 32 -- a whole workgroup on a row of 32 entries is mostly idle lanes).
inline int seg_long_from(long long entries, int rows)
{
    const long long twice = 2 * entries / (rows > 0 ? rows : 1);
    return (int)(twice < 32 ? 32 : twice > kSegLongFrom ? kSegLongFrom : twice);
}
constexpr int kSegLongThreads = 512;

template <int LPR, int SRC_DIV>
__device__ __forceinline__ void seg_reduce_long_body(unsigned blk, unsigned nblk, long long out_rows, int rows, long long entries, int c,
                                                     const float *__restrict__ grad_out, const float *__restrict__ weight,
                                                     const int *__restrict__ start, const int *__restrict__ list,
                                                     float *__restrict__ out, int long_from)
{
    constexpr int NG = kSegLongThreads / LPR;                      // lane groups per workgroup
    __shared__ float4 part[kSegLongThreads];                       // [NG][LPR]
    __shared__ int row_beg[kSegLongThreads], row_len[kSegLongThreads];
    __shared__ unsigned long long long_mask[kSegLongThreads / 64];
    const int t = threadIdx.x, g = t / LPR, gl = t % LPR, lane = t & 63, wv = t >> 6;
    // the workgroup looks at rows blockIdx.x + k gridDim.x (long rows sit at the low indices of every cloud: interleaved, they
    // spread over the workgroups), 512 of them per pass, one per thread
    for (long long base = blk; base < out_rows; base += (long long)nblk * kSegLongThreads) {
        const long long my = base + (long long)t * nblk;
        int beg = 0, len = 0;
        if (my < out_rows) {
            const long long i = my / rows;
            const int r = (int)(my - i * rows);
            beg = start[i * (rows + 1) + r];
            len = start[i * (rows + 1) + r + 1] - beg;
        }
        const bool is_long = len >= long_from;
        row_beg[t] = beg; row_len[t] = len;
        const unsigned long long m = __ballot(is_long);
        if (lane == 0) long_mask[wv] = m;
        __syncthreads();
        for (int w2 = 0; w2 < kSegLongThreads / 64; ++w2) {
            unsigned long long todo = long_mask[w2];                // workgroup-uniform
            while (todo) {
                const int bit = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int q = w2 * 64 + bit;                        // the thread slot that looked at this row
                const long long row = base + (long long)q * nblk;
                const long long i = row / rows;
                const int rbeg = row_beg[q], rlen = row_len[q];
                const int *seg = list + i * entries + rbeg;
                const float *src = grad_out + (size_t)i * (entries / SRC_DIV) * c;
                const float *wsrc = weight ? weight + (size_t)i * entries : nullptr;
                // a lane owns the float4s gl and gl + LPR of the row (the second one where the row is wider than LPR float4s:
                // c = 320 on 64 lanes -- as a second pass over the segment it cost the whole pass again for a quarter of the
                // lanes); rows wider than 2 LPR float4s take more sweeps of the segment
                const int per = c / 4;
                for (int f0 = 0; f0 < per; f0 += 2 * LPR) {
                    const int fa = f0 + gl, fb = f0 + LPR + gl;
                    const bool oka = fa < per, okb = fb < per;
                    const int ca = (oka ? fa : 0) * 4, cb = (okb ? fb : 0) * 4;
                    const bool wide = f0 + LPR < per;               // workgroup-uniform: somebody owns a second float4
                    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
                    // eight row loads in flight per lane either way: two batches of four entries (one float4 each), or one
                    // batch with both float4s of the lane
                    auto sweep = [&](auto widec) __attribute__((always_inline)) {
                        constexpr bool WIDE = decltype(widec)::value;
                        constexpr int NE = 8;                     // (wide rows too: a 1000-entry row of c = 320 is the end of the launch -- 32 entries per batch made it 31 dependent batches)
                        // (fetching the NEXT batch's entry numbers ahead of this batch's rows -- one round trip per batch instead of two
                        // -- was measured: c = 320 172 -> 163 us, c = 128 38.5 -> 42.7: not kept)
                        for (int p = g * 4; p < rlen; p += NG * NE) {
                            int e[NE];
                            float wq[NE];
                            float4 va[NE], vb[WIDE ? NE : 1];
#pragma unroll
                            for (int u = 0; u < NE; ++u) {
                                const int pe = p + (u >> 2) * NG * 4 + (u & 3);
                                e[u] = seg[pe < rlen ? pe : rlen - 1];
                            }
#pragma unroll
                            for (int u = 0; u < NE; ++u) {
                                const float *rowp = src + (size_t)(e[u] / SRC_DIV) * c;
                                va[u] = *reinterpret_cast<const float4 *>(rowp + ca);
                                if (WIDE) vb[u] = *reinterpret_cast<const float4 *>(rowp + cb);
                                wq[u] = wsrc ? wsrc[e[u]] : 1.0f;
                            }
#pragma unroll
                            for (int u = 0; u < NE; ++u) {
                                const int pe = p + (u >> 2) * NG * 4 + (u & 3);
                                if (pe < rlen) {
                                    float4 x = va[u];
                                    if (wsrc) { x.x = __fmul_rn(x.x, wq[u]); x.y = __fmul_rn(x.y, wq[u]); x.z = __fmul_rn(x.z, wq[u]); x.w = __fmul_rn(x.w, wq[u]); }
                                    acc0.x = __fadd_rn(acc0.x, x.x); acc0.y = __fadd_rn(acc0.y, x.y); acc0.z = __fadd_rn(acc0.z, x.z); acc0.w = __fadd_rn(acc0.w, x.w);
                                    if (WIDE) {
                                        float4 y = vb[u];
                                        if (wsrc) { y.x = __fmul_rn(y.x, wq[u]); y.y = __fmul_rn(y.y, wq[u]); y.z = __fmul_rn(y.z, wq[u]); y.w = __fmul_rn(y.w, wq[u]); }
                                        acc1.x = __fadd_rn(acc1.x, y.x); acc1.y = __fadd_rn(acc1.y, y.y); acc1.z = __fadd_rn(acc1.z, y.z); acc1.w = __fadd_rn(acc1.w, y.w);
                                    }
                                }
                            }
                        }
                    };
                    if (wide) sweep(std::true_type()); else sweep(std::false_type());
                    for (int half = 0; half < (wide ? 2 : 1); ++half) {
                        float4 mine = acc0;
                        if (half) mine = acc1;
                        part[g * LPR + gl] = mine;
                        __syncthreads();
                        const bool okh = half ? okb : oka;
                        if (g == 0 && okh) {
                            float4 sum = part[gl];
#pragma unroll 4
                            for (int k = 1; k < NG; ++k) {
                                const float4 o = part[k * LPR + gl];
                                sum.x = __fadd_rn(sum.x, o.x); sum.y = __fadd_rn(sum.y, o.y); sum.z = __fadd_rn(sum.z, o.z); sum.w = __fadd_rn(sum.w, o.w);
                            }
                            *reinterpret_cast<float4 *>(out + row * c + (half ? cb : ca)) = sum;
                        }
                        __syncthreads();
                    }
                }
            }
        }
        __syncthreads();                                            // row_beg / row_len / long_mask are rewritten by the next pass
    }
}


// ONE launch for both (default mode): blocks [0, nlong) sum the long rows -- dependent chains of row loads, dispatched first --,
// the other blocks the short rows at streaming rate beside them (as two launches: 18 + 38 us at cls_ssg L2, 107 + 147 at
// cls_msg L2; the long-row kernel alone uses a fraction of the memory system)
template <int LPR, int SRC_DIV, bool WIDE = false>
__global__ __launch_bounds__(kSegLongThreads) void seg_reduce_split_kernel(unsigned nlong, long long out_rows, int rows, long long entries,
                                                                           int c, const float *__restrict__ grad_out,
                                                                           const float *__restrict__ weight,
                                                                           const int *__restrict__ start,
                                                                           const int *__restrict__ sorted,
                                                                           const int *__restrict__ list, float *__restrict__ out, int long_from)
{
    if (blockIdx.x < nlong)
        seg_reduce_long_body<LPR, SRC_DIV>(blockIdx.x, nlong, out_rows, rows, entries, c, grad_out, weight, start, list, out, long_from);
    else {
        // Workgroups go to the eight XCDs round-robin, each XCD has an L2 of its own, and a gradient row of three_interpolate is
        // referenced by THREE target rows: target rows dealt to workgroups in launch order put those three readers on three XCDs
        // (PMC traffic 2.7 x algorithmic at sem_seg FP4). Every eighth workgroup -- one XCD -- therefore takes one CONTIGUOUS
        // eighth of the target rows, i.e. whole clouds (the short part's workgroup count is a multiple of 8).
        const unsigned s = blockIdx.x - nlong, n = gridDim.x - nlong;
        seg_reduce_body<LPR, true, false, SRC_DIV, kSegLongThreads, WIDE>((s & 7u) * (n >> 3) + (s >> 3), n, out_rows, rows, entries, c, grad_out,
                                                                     weight, start, sorted, list, out, long_from);
    }
}

// Workgroups of the long-row part = the stride of the rows a workgroup looks at (row blk + k wg, k = 0, 1, ...). The long rows are
// the LOW point numbers of every cloud, so what matters is the stride's residue modulo the rows per cloud: with a common factor
// every cloud's row r goes to the same few workgroups (wg = rows = 512: thirty workgroups did all the work, 1758 us), and with a
// residue near 0 -- 1021 = 2 x 512 - 3, the rule until the last session of round 6 -- a workgroup's rows walk down three point
// numbers at a time, so the workgroups that start at a low number get ALL their ~16 rows long and the others none (cls_ssg L2:
// ~200 of 1021 workgroups summed every long row, 42 us for 144 MB). The residue is therefore put at the golden section of the
// rows per cloud (consecutive k land far apart and fill in evenly), coprime with it.
// wide: rows of 64 lanes (c > 128): half as many lane groups per workgroup, a long row takes twice the batches -- twice the
// workgroups (c = 320 at cls_msg L2: 172 -> 159 us; at c = 128 the extra workgroups cost more than they balance: 38.5 -> 40.2).
static long long seg_long_blocks(long long out_rows, int rows, bool wide)
{
    const int per = wide ? 8 : 16;                                    // rows looked at per workgroup
    long long cap = (out_rows + per - 1) / per;
    const long long most = (wide || rows > 1021) ? 2045 : 1021;
    if (cap > most) cap = most;
    if (cap < 1) cap = 1;
    double tg = 0.381966 * rows;
    while (tg > (double)cap && tg > 2.0) tg *= 0.618034;              // fewer workgroups than that: the next golden fraction of the rows
    const long long target = (long long)tg;
    for (long long d = 0; d < rows; ++d)
        for (int sgn = 0; sgn < 2; ++sgn) {
            const long long r = sgn ? target - d : target + d;
            if (r <= 0 || r >= rows || r > cap || std::__gcd(r, (long long)rows) != 1) continue;
            return (cap - r) / rows * rows + r;                       // the largest q rows + r within the cap
        }
    long long wg = cap | 1;                                           // (rows per cloud of 1 or 2: nothing to spread)
    while (wg > 1 && std::__gcd((long long)rows, wg) > 1) wg -= 2;
    return wg;
}

template <bool DET, int SRC_DIV>
static int launch_reduce(long long out_rows, int rows, long long entries, int c, const float *grad_out, const float *weight,
                         const SegWs &w, float *out, hipStream_t st)
{
    // 16-byte row accesses need 16-byte aligned bases (a C-ABI caller may pass a sub-allocated pointer)
    const bool vec4 = (c & 3) == 0 && ((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    const int per = vec4 ? c / 4 : c;                             // lanes a row can use
    int lpr = 1;
    while (lpr < per && lpr < 64) lpr <<= 1;
    const long long threads = out_rows * lpr;
    // default mode, 16-byte rows, at least 16 lanes per row: the long rows go to seg_reduce_long_kernel (second launch below)
    const bool split = !DET && vec4 && lpr >= 16;
#define PN2_SEG_CASE(L)                                                                                              \
    if (lpr == L) {                                                                                                  \
        if (split && L >= 16) {                                                                                      \
            /* ~16 rows looked at per long-row workgroup, at most 1021 of them. The long rows are the LOW point numbers of every \
               cloud: workgroup w looks at rows w, w + wg, w + 2 wg, ..., so wg must not share a factor with the rows per cloud -- \
               with wg = rows = 512 every cloud's row r went to workgroup r and thirty workgroups did all the work (1758 us) */ \
            const long long wg = seg_long_blocks(out_rows, rows, L >= 64);                                           \
            const unsigned ga = (seg_grid(threads, kSegLongThreads) + 7u) & ~7u;   /* a multiple of 8: seg_reduce_split_kernel's XCD map */ \
            if (L == 64 && per > 64 && per <= 128)              /* c in (256, 512]: both float4s of a lane in one sweep */ \
                return launch((seg_reduce_split_kernel<64, SRC_DIV, true>), dim3((unsigned)wg + ga), dim3(kSegLongThreads), 0, st, \
                              (unsigned)wg, out_rows, rows, entries, c, grad_out, weight, w.start, w.sorted, w.list, out, seg_long_from(entries, rows)); \
            return launch((seg_reduce_split_kernel<(L >= 16 ? L : 16), SRC_DIV>), dim3((unsigned)wg + ga), dim3(kSegLongThreads), 0, st, \
                          (unsigned)wg, out_rows, rows, entries, c, grad_out, weight, w.start, w.sorted, w.list, out, seg_long_from(entries, rows)); \
        }                                                                                                            \
        if (vec4)                                                                                                    \
            return launch((seg_reduce_kernel<L, true, DET, SRC_DIV>), dim3(seg_grid(threads)), dim3(kSegThreads), 0, st, \
                          out_rows, rows, entries, c, grad_out, weight, w.start, w.sorted, w.list, out, 0);         \
        return launch((seg_reduce_kernel<L, false, DET, SRC_DIV>), dim3(seg_grid(threads)), dim3(kSegThreads), 0, st, \
                      out_rows, rows, entries, c, grad_out, weight, w.start, w.sorted, w.list, out, 0);             \
    }
    PN2_SEG_CASE(1) PN2_SEG_CASE(2) PN2_SEG_CASE(4) PN2_SEG_CASE(8) PN2_SEG_CASE(16) PN2_SEG_CASE(32) PN2_SEG_CASE(64)
#undef PN2_SEG_CASE
    return PN2_E_TOO_LARGE;
}

// invert idx (b clouds x `entries` references into `rows` targets), then reduce
template <int SRC_DIV>
static int seg_grad(int b, int rows, long long entries, int c, const float *grad_out, const int *idx, const float *weight,
                    float *out, void *ws, int deterministic, hipStream_t st)
{
    SegWs w = seg_ws(ws, b, rows, entries);
    const long long total = (long long)b * entries;
    if (rows <= kSegLdsRows && b >= 4) {
        // enough clouds to spread over CUs: the whole inversion of a cloud in one workgroup, LDS counters
        const size_t with_list = sizeof(int) * ((size_t)rows + (size_t)entries);
        const bool fits = with_list <= 144 * 1024;                    // + 8 KiB static for the long-row lists
        const bool sort = deterministic && fits;
        const size_t lds = fits ? with_list : sizeof(int) * (size_t)rows;
        constexpr bool RUNS = SRC_DIV == 1;                           // group_point's idx: padded ball-query lists
        auto kern = sort ? seg_invert_lds_kernel<true, true, RUNS> : fits ? seg_invert_lds_kernel<false, true, RUNS> : seg_invert_lds_kernel<false, false, RUNS>;
        if (!sort && fits) {                                          // default mode: the entries stay in registers between the passes
            const long long nb = (entries + 8191) / 8192;
            if (nb == 1) kern = seg_invert_lds_kernel<false, true, RUNS, 1>;
            else if (nb == 2) kern = seg_invert_lds_kernel<false, true, RUNS, 2>;
            else if (nb == 3) kern = seg_invert_lds_kernel<false, true, RUNS, 3>;
            else if (nb == 4) kern = seg_invert_lds_kernel<false, true, RUNS, 4>;
        }
        if (int rc = allow_dynamic_lds(kern, lds)) return rc;
        if (int rc = launch(kern, dim3(b), dim3(1024), lds, st, entries, rows, idx, w.start, w.sorted, w.list)) return rc;
    } else {
    if (int rc = clear_async(w.cursor, 2 * sizeof(int) * (size_t)b * rows, st)) return rc;   // counters and sorted flags
    if (int rc = launch(seg_count_kernel, dim3(seg_grid(total)), dim3(kSegThreads), 0, st, total, entries, rows, idx, w.cursor)) return rc;
    if (int rc = launch(seg_scan_kernel, dim3(b), dim3(1024), 0, st, rows, w.start, w.cursor)) return rc;
    if (int rc = launch(seg_fill_kernel, dim3(seg_grid(total)), dim3(kSegThreads), 0, st, total, entries, rows, idx, w.cursor, w.list)) return rc;
    }
    const long long out_rows = (long long)b * rows;
    return deterministic ? launch_reduce<true, SRC_DIV>(out_rows, rows, entries, c, grad_out, weight, w, out, st)
                         : launch_reduce<false, SRC_DIV>(out_rows, rows, entries, c, grad_out, weight, w, out, st);
}

}  // namespace pn2

// Host logic of the default mode's long-row part for a shape (no device work): what launch_reduce computes. Tests simulate the
// assignment of the low point numbers -- where the long rows are -- to workgroups from it (tests/test_seg_plan_model.py).
extern "C" int pn2_seg_grad_plan(int rows, long long entries, int c, long long out_rows, int *long_from, int *long_blocks)
{
    using namespace pn2;
    if (rows <= 0 || entries < 0 || c <= 0 || out_rows <= 0) return PN2_E_SHAPE;
    const int per = (c & 3) == 0 ? c / 4 : c;
    int lpr = 1;
    while (lpr < per && lpr < 64) lpr <<= 1;
    if (long_from) *long_from = seg_long_from(entries, rows);
    if (long_blocks) *long_blocks = (int)seg_long_blocks(out_rows, rows, lpr >= 64);
    return PN2_OK;
}

extern "C" long long pn2_seg_grad_ws_bytes(int b, int rows, long long entries)
{
    if (b <= 0 || rows <= 0 || entries < 0) return 16;
    return (long long)sizeof(int) * ((long long)b * (rows + 1) + 2ll * b * rows + (long long)b * entries) + 16;
}

extern "C" int pn2_group_point_grad_seg(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                                        float *grad_points, void *ws, int deterministic, void *stream)
{
    using namespace pn2;
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!grad_points || !ws) return PN2_E_NULL;
    const long long entries = (long long)m * nsample;
    if (entries > INT_MAX || (long long)b * n > INT_MAX) return PN2_E_TOO_LARGE;
    hipStream_t st = as_stream(stream);
    if (entries == 0) {
        return clear_async(grad_points, sizeof(float) * (size_t)b * n * c, st);
    }
    if (!grad_out || !idx) return PN2_E_NULL;
    return seg_grad<1>(b, n, entries, c, grad_out, idx, nullptr, grad_points, ws, deterministic, st);
}

extern "C" int pn2_three_interpolate_grad_seg(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                              const float *weight, float *grad_points, void *ws, int deterministic,
                                              void *stream)
{
    using namespace pn2;
    if (b < 0 || n < 0 || c <= 0 || m <= 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!grad_points || !ws) return PN2_E_NULL;
    const long long entries = (long long)n * 3;
    if (entries > INT_MAX || (long long)b * m > INT_MAX) return PN2_E_TOO_LARGE;
    hipStream_t st = as_stream(stream);
    if (entries == 0) {
        return clear_async(grad_points, sizeof(float) * (size_t)b * m * c, st);
    }
    if (!grad_out || !idx || !weight) return PN2_E_NULL;
    return seg_grad<3>(b, m, entries, c, grad_out, idx, weight, grad_points, ws, deterministic, st);
}
