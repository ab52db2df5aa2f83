// ball_query_body.h -- device body of the ball-query sweep (shared by ball_query.hip and sa_fused.hip).
// See ball_query.hip for the design notes and the measurement history.
#pragma once
#include "pn2_device.h"

#include <math.h>

namespace pn2 {

typedef unsigned long long __attribute__((address_space(1))) pn2_gu64b;

constexpr int kBqThreads = 512;
constexpr int kBqWaves = kBqThreads / PN2_WAVE;
constexpr int kBqQpw = 2;              // queries swept together by one wave (each LDS read serves both)
constexpr int kBqMaxLdsPoints = 9600;  // 16 B/point + row buffers must stay under 160 KiB

// popcount(mask) + acc as two VALU ops on a VGPR accumulator. On gfx9-class CUs a scalar op costs a
// SIMD issue slot of 4 cycles (a VALU op 2), and the first version of this kernel was bound by its
// ~60 scalar ops per trip (rocprofv3: 64 us at the metric shape); the hit counters therefore live in
// VGPRs (wave-uniform values) and only the exec-mask updates of the predicated stores stay scalar.
__device__ __forceinline__ int vbcnt_acc(unsigned long long mask, int acc)
{
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2\n\tv_bcnt_u32_b32 %0, %3, %0"
        : "=&v"(r)
        : "s"((unsigned)mask), "v"(acc), "s"((unsigned)(mask >> 32)));
    return r;
}

// A query's hits are kept as a BITMAP while the cloud is swept: the ballot of chunk c (64 candidates)
// is the c-th 64-bit word, parked in lane c of two VGPRs (one v_cndmask per half word, no
// LDS traffic, no exec-mask juggling). One "window" is 64 chunks = 4096 candidates; bq_flush turns
// the window's bitmap into the ordered index list: a wave-wide prefix sum of the word popcounts gives
// every lane its first output slot, then each lane peels its set bits in ascending order.
// The ordered-compaction-per-chunk version of this kernel spent more issue slots on appending hits
// (7 VALU + 3 SALU per chunk and query) than on the distances themselves.
// Park the two queries' ballots of chunk c in lane c of their bitmap registers (`sel` = lane id == c,
// one v_cmp per chunk shared by the four half words). v_writelane_b32 would need two scalar operands
// (value + lane select), which gfx9's single constant-bus port only allows through M0.
__device__ __forceinline__ void bq_park(unsigned &a_lo, unsigned &a_hi, unsigned &b_lo, unsigned &b_hi,
                                        unsigned long long ma, unsigned long long mb, bool sel)
{
    a_lo = sel ? (unsigned)ma : a_lo;
    a_hi = sel ? (unsigned)(ma >> 32) : a_hi;
    b_lo = sel ? (unsigned)mb : b_lo;
    b_hi = sel ? (unsigned)(mb >> 32) : b_hi;
}

__device__ __forceinline__ int wave_prefix_sum_incl(int v)
{
    // Hillis-Steele over DPP row shifts, then the two cross-row broadcasts
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); v += t;    // row_shr:1
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true); v += t;    // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); v += t;    // row_shr:4
    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true); v += t;    // row_shr:8
    t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); v += t;   // row_bcast:15 -> rows 1,3
    t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); v += t;   // row_bcast:31 -> rows 2,3
    return v;
}

// Append the window's hits (bitmap words mlo/mhi, lane c = chunk c) to rowbuf in ascending index order.
// `done` = hits already in rowbuf (wave-uniform); returns the new total (may exceed nsample).
__device__ __forceinline__ int bq_flush(unsigned mlo, unsigned mhi, int window_base, int done, int nsample,
                                        int *rowbuf, int lane)
{
    const int pc = __popc(mlo) + __popc(mhi);
    const int incl = wave_prefix_sum_incl(pc);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int pos = done + incl - pc;                       // first output slot of this lane's word
    unsigned long long word = ((unsigned long long)mhi << 32) | mlo;
    const int kbase = window_base + lane * 64;
    while (__any(word != 0ull && pos < nsample)) {
        if (word != 0ull && pos < nsample) rowbuf[pos] = kbase + __builtin_ctzll(word);
        word &= word - 1ull;
        ++pos;
    }
    return done + total;
}

template <bool LDS_CLOUD, bool FUSE>
__device__ __forceinline__ void bq_emit(size_t row, int nsample, int cnt, const int *rowbuf,
                                        const float4 *cloud, const float *__restrict__ data, float qx, float qy,
                                        float qz, int *__restrict__ idx, int *__restrict__ pts_cnt,
                                        float *__restrict__ grouped, int subtract, int lane)
{
    const int first = cnt > 0 ? rowbuf[0] : 0;   // the first hit pads the row; zeros when the ball is empty
    for (int l = lane; l < nsample; l += 64) {
        const int v = (l < cnt) ? rowbuf[l] : first;
        if (idx) idx[row * nsample + l] = v;
        if (FUSE && grouped) {
            float gx, gy, gz;
            if (LDS_CLOUD) {
                const float4 p = cloud[v];
                gx = p.x; gy = p.y; gz = p.z;
            } else {
                gx = data[(size_t)v * 3 + 0]; gy = data[(size_t)v * 3 + 1]; gz = data[(size_t)v * 3 + 2];
            }
            if (subtract) { gx = __fsub_rn(gx, qx); gy = __fsub_rn(gy, qy); gz = __fsub_rn(gz, qz); }
            float *o = grouped + (row * nsample + l) * 3;
            o[0] = gx; o[1] = gy; o[2] = gz;
        }
    }
    if (lane == 0 && pts_cnt) pts_cnt[row] = cnt;
}

// Wait until the FPS workgroup of the same launch has published sample j of this cloud and return its
// index (consumer side of the R2 granule hand-off: ONE 8-byte agent-scope relaxed load per poll, the
// tag travels with the data, no fence). The producers of a launch are running before any consumer polls
// (sa_fused.hip: producers are the first b blocks, and the launch is refused unless they all fit), so the wait always ends; the spin is nevertheless
// bounded (~10 s; the longest legal chain, n = m = 8192, takes < 10 ms) and returns -1 instead of hanging
// the GPU should the device ever stall a producer for that long. The caller records the failure in the
// launch's status word and gives up its queries: an error the host can read, not a trap that would take
// the whole HIP context down.
__device__ __forceinline__ int bq_poll_sample(const unsigned long long *tagged, unsigned tag = 1u)
{
    const pn2_gu64b *g = (const pn2_gu64b *)tagged;
    for (unsigned it = 0;; ++it) {
        const unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == tag) return (int)(unsigned)v;
#ifndef PN2_POLL_SLEEP
#define PN2_POLL_SLEEP 16
#endif
        __builtin_amdgcn_s_sleep(PN2_POLL_SLEEP);
#ifndef PN2_POLL_LIMIT
#define PN2_POLL_LIMIT (1u << 23)          /* ~10 s; the lab build of tests/test_overlap_status_gpu.py sets 2 to force the give-up path */
#endif
        if (it > PN2_POLL_LIMIT) return -1;
    }
}

// One workgroup's share of the ball queries of cloud `bi`: queries [q0, q1).
// POLL: the query points are not read from xyz2; they are the FPS samples of the same launch, taken
// from the tagged index stream as soon as they exist (and written to new_xyz on the way).
// STAGED: the LDS copy of the cloud already exists (multi-radius kernel: staged once, swept per radius);
// row_stride: ints between two row buffers (0 = nsample; the multi-radius kernel passes its largest nsample
// so that waves working on different radii never share a buffer).
template <bool LDS_CLOUD, bool FUSE, bool POLL, int NT = kBqThreads, bool STAGED = false>
__device__ __forceinline__ bool bq_block_body(int n, int m, int nsample, float thr, int bi, int q0, int q1,
                                              const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                              const unsigned long long *__restrict__ tagged,
                                              float *__restrict__ new_xyz, int *__restrict__ idx,
                                              int *__restrict__ pts_cnt, float *__restrict__ grouped, int subtract,
                                              char *smem, unsigned tag = 1u, int row_stride = 0,
                                              unsigned *status = nullptr)
{
    if (row_stride == 0) row_stride = nsample;
    float4 *cloud = reinterpret_cast<float4 *>(smem);                                   // [n] when LDS_CLOUD
    int *rowbuf_all = reinterpret_cast<int *>(smem + (LDS_CLOUD ? sizeof(float4) * (size_t)((n + 127) & ~127) : 0));

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *__restrict__ data = xyz1 + (size_t)bi * n * 3;
    int *rowbuf0 = rowbuf_all + (w * kBqQpw + 0) * row_stride;
    int *rowbuf1 = rowbuf_all + (w * kBqQpw + 1) * row_stride;

    // LDS copy of the cloud, padded to a multiple of 128 with points at +inf: a padded candidate's
    // distance is +inf (or NaN), never < thr, so the sweep needs no bounds predicate.
    const int npad = (n + 127) & ~127;
    if (LDS_CLOUD && !STAGED) {
        for (int k = t; k < npad; k += NT) {
            if (k < n) {
                const float *p = data + (size_t)k * 3;
                cloud[k] = make_float4(p[0], p[1], p[2], 0.0f);
            } else {
                cloud[k] = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
            }
        }
        __syncthreads();
    }

    for (int j = q0 + w * kBqQpw; j < q1; j += (NT / PN2_WAVE) * kBqQpw) {
        const bool two = (j + 1) < q1;                       // wave-uniform
        const size_t row0 = (size_t)bi * m + j;
        const size_t row1 = row0 + (two ? 1 : 0);
        float ax, ay, az, bx, by, bz;
        if (POLL) {
            const int ka = bq_poll_sample(tagged + row0, tag);
            const int kb = bq_poll_sample(tagged + row1, tag);
            if (ka < 0 || kb < 0) {                              // wave-uniform; see bq_poll_sample
                if (lane == 0 && status) atomicExch(status, 1u);
                return false;
            }
            const float4 pa = cloud[ka], pb = cloud[kb];          // POLL implies LDS_CLOUD
            ax = pa.x; ay = pa.y; az = pa.z; bx = pb.x; by = pb.y; bz = pb.z;
            if (lane == 0) {
                float *o = new_xyz + row0 * 3;
                o[0] = ax; o[1] = ay; o[2] = az;
                if (two) { o[3] = bx; o[4] = by; o[5] = bz; }
            }
        } else {
            ax = xyz2[row0 * 3 + 0]; ay = xyz2[row0 * 3 + 1]; az = xyz2[row0 * 3 + 2];
            bx = xyz2[row1 * 3 + 0]; by = xyz2[row1 * 3 + 1]; bz = xyz2[row1 * 3 + 2];
        }
        // hit counters: wave-uniform values kept in VGPRs (see vbcnt_acc); bitmap words: lane c = chunk c
        int cnt0 = 0, cnt1 = two ? 0 : nsample;
        asm volatile("v_mov_b32 %0, %0" : "+v"(cnt0));
        asm volatile("v_mov_b32 %0, %0" : "+v"(cnt1));
        int done0 = 0, done1 = 0;                            // hits already flushed to the row buffers
        for (int wbase = 0; wbase < n; wbase += 4096) {      // one window = 64 chunks of 64 candidates
            unsigned m0lo = 0u, m0hi = 0u, m1lo = 0u, m1hi = 0u;
            const int wend = min(n, wbase + 4096);
            // 128 candidates per trip: two chunks x two queries = four independent distance chains per lane
            for (int base = wbase; base < wend; base += 128) {
                const int kA = base + lane, kB = base + 64 + lane;
                float pax, pay, paz, pbx, pby, pbz;
                if (LDS_CLOUD) {
                    const float4 pa = cloud[kA];
                    const float4 pb = cloud[kB];
                    pax = pa.x; pay = pa.y; paz = pa.z; pbx = pb.x; pby = pb.y; pbz = pb.z;
                } else {
                    const float inf = INFINITY;
                    const float *pa = data + (size_t)min(kA, n - 1) * 3;
                    const float *pb = data + (size_t)min(kB, n - 1) * 3;
                    pax = kA < n ? pa[0] : inf; pay = pa[1]; paz = pa[2];
                    pbx = kB < n ? pb[0] : inf; pby = pb[1]; pbz = pb[2];
                }
                // reference operand order: (x2-x1) with x2 the query (query_ball_point.cpp:26-32)
                const float sA0 = sqdist(ax, ay, az, pax, pay, paz);
                const float sB0 = sqdist(ax, ay, az, pbx, pby, pbz);
                const float sA1 = sqdist(bx, by, bz, pax, pay, paz);
                const float sB1 = sqdist(bx, by, bz, pbx, pby, pbz);
                const unsigned long long mA0 = __ballot(sA0 < thr), mB0 = __ballot(sB0 < thr);
                const unsigned long long mA1 = __ballot(sA1 < thr), mB1 = __ballot(sB1 < thr);
                const int cA = (base - wbase) >> 6;          // chunk number inside the window (scalar)
                bq_park(m0lo, m0hi, m1lo, m1hi, mA0, mA1, lane == cA);
                bq_park(m0lo, m0hi, m1lo, m1hi, mB0, mB1, lane == cA + 1);
                cnt0 = vbcnt_acc(mB0, vbcnt_acc(mA0, cnt0));
                cnt1 = vbcnt_acc(mB1, vbcnt_acc(mA1, cnt1));
                // stop as soon as both rows are full (reference: break at cnt == nsample, :23-24)
                if (__builtin_amdgcn_readfirstlane(min(cnt0, cnt1)) >= nsample) break;
            }
            if (done0 < nsample) done0 = bq_flush(m0lo, m0hi, wbase, done0, nsample, rowbuf0, lane);
            if (done1 < nsample && two) done1 = bq_flush(m1lo, m1hi, wbase, done1, nsample, rowbuf1, lane);
            if (__builtin_amdgcn_readfirstlane(min(cnt0, cnt1)) >= nsample) break;
        }
        cnt0 = __builtin_amdgcn_readfirstlane(min(cnt0, nsample));
        cnt1 = __builtin_amdgcn_readfirstlane(min(cnt1, nsample));
        // LDS ops of one wave execute in order; only the compiler must not reorder
        asm volatile("" ::: "memory");
        bq_emit<LDS_CLOUD, FUSE>(row0, nsample, cnt0, rowbuf0, cloud, data, ax, ay, az, idx, pts_cnt, grouped,
                                 subtract, lane);
        if (two)
            bq_emit<LDS_CLOUD, FUSE>(row1, nsample, cnt1, rowbuf1, cloud, data, bx, by, bz, idx, pts_cnt,
                                     grouped, subtract, lane);
        asm volatile("" ::: "memory");
    }
    return true;
}


// ------------------------------------------------------------------------------------------------
// Cell-list variant. The brute-force sweep above evaluates every (query, candidate) pair until the
// row is full; at the metric shape that is ~2400 of 4096 candidates per query for ~40 hits. Here each
// workgroup first bins its cloud into a uniform grid (counting sort in LDS, cells of edge >= the
// search radius), and a query only visits the <= 3x3 runs of x-adjacent cells around it. The result
// is still index-exact: the hit predicate is the same fp32 expression on the same operands, every
// in-ball point lies in a visited cell (see bq_cell / the reach margin), and the ORDER comes from the
// hit bitmap (indexed by original point number), not from the visiting order.
constexpr int kBqGridMax = 12;                                      // cells per axis
constexpr int kBqCellsMax = kBqGridMax * kBqGridMax * kBqGridMax;   // 1728 -> 6.75 KiB of cell ends
constexpr int kBqMinCells = 64;        // coarser grids prune nothing: the sweep with its early exit wins
constexpr int kBqMiscBytes = 1024;     // reduction scratch of the binning pass

struct BqGrid {
    float ox, oy, oz, ix, iy, iz;
    int gx, gy, gz;
    int nb;                                // index blocks of the cell list (bq_build_grid<NT, true>): 1, or 2 = sorted by (k >= n / 2, cell)
};

// Monotone non-decreasing in v (every fp32 operation below is), which is all the pruning needs:
// cell(a) <= cell(p) <= cell(b) whenever a <= p <= b. NaN lands in cell 0; it can never be a hit.
__device__ __forceinline__ int bq_cell(float v, float o, float inv, int g)
{
    const float f = __fmul_rn(__fsub_rn(v, o), inv);
    return (f >= 0.0f) ? (int)fminf(f, (float)(g - 1)) : 0;
}

// Wave-wide min / max that leave the result in lane 63: DPP row shifts + the two row broadcasts
// (6 VALU ops per value; the ds_bpermute butterfly this replaces cost an LDS round trip per step).
// A lane without a DPP source keeps its own value (old = v, bound_ctrl off).
template <bool MAX>
__device__ __forceinline__ float wave_minmax_lane63(float v)
{
#define PN2_BQ_STEP(ctrl, rmask)                                                                              \
    {                                                                                                         \
        const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, \
                                                                   rmask, 0xf, false));                       \
        v = MAX ? fmaxf(v, o) : fminf(v, o);                                                                  \
    }
    PN2_BQ_STEP(0x111, 0xf)   // row_shr:1
    PN2_BQ_STEP(0x112, 0xf)   // row_shr:2
    PN2_BQ_STEP(0x114, 0xf)   // row_shr:4
    PN2_BQ_STEP(0x118, 0xf)   // row_shr:8   -> lane 15 of each row holds the row result
    PN2_BQ_STEP(0x142, 0xa)   // row_bcast:15 -> rows 1, 3
    PN2_BQ_STEP(0x143, 0xc)   // row_bcast:31 -> rows 2, 3: lane 63 holds the wave result
#undef PN2_BQ_STEP
    return v;
}

__device__ __forceinline__ int wave_max_i32_lane63(int v)
{
#define PN2_BQ_STEP(ctrl, rmask) v = max(v, __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, 0xf, false));
    PN2_BQ_STEP(0x111, 0xf) PN2_BQ_STEP(0x112, 0xf) PN2_BQ_STEP(0x114, 0xf) PN2_BQ_STEP(0x118, 0xf)
    PN2_BQ_STEP(0x142, 0xa) PN2_BQ_STEP(0x143, 0xc)
#undef PN2_BQ_STEP
    return v;
}

// LDS layout of the cell-list kernel:
//   sorted points float4[n] | cell table int[kBqCellsMax + 4] | per wave {G bitmaps, G rows} | scratch
// G = 64 / LPQ queries are in flight per wave, LPQ lanes each.
constexpr int kBqTabInts = kBqCellsMax + 4;                         // tab[0] = 0, tab[c + 1] = end of cell c
__host__ __device__ __forceinline__ size_t bq_cells_wave_bytes(int n, int nsample, int lpq)
{
    const size_t nwin = ((size_t)n + 4095) / 4096;
    const size_t g = (size_t)(64 / lpq);
    // per wave: G bitmaps, G row buffers (padded to 16 bytes together), G emit headers (float4)
    return g * nwin * 512 + sizeof(int) * ((g * (size_t)nsample + 3) & ~(size_t)3) + g * 16;
}
__host__ __device__ __forceinline__ size_t bq_cells_lds_bytes(int n, int nsample, int lpq, int nthreads)
{
    return sizeof(float4) * (size_t)n + sizeof(int) * (size_t)kBqTabInts +
           (size_t)(nthreads / 64) * bq_cells_wave_bytes(n, nsample, lpq) + kBqMiscBytes;
}

constexpr int kBqCellsMaxPoints = 8192;                            // 16 points per thread in registers

// Count one point per lane into its cell. Points of one wave instruction that share a cell are common
// (duplicated points, the reference's dropout-to-first-point augmentation: up to 87 % of a cloud in ONE
// cell) and same-address LDS atomics serialise, so the cell of the first lane is peeled with a single
// aggregated atomic; the rest go one per lane. Nothing is returned: the atomics are fire-and-forget.
__device__ __forceinline__ void bq_cell_count(int *cellend, int c, bool valid, int lane)
{
    const unsigned long long act = __ballot(valid);
    if (act == 0ull) return;
    const int leader = __builtin_ctzll(act);
    const int c0 = __builtin_amdgcn_readlane(c, leader);
    const bool mine = valid && c == c0;
    const unsigned long long mask = __ballot(mine);
    if (lane == leader) atomicAdd(&cellend[c0], __popcll(mask));
    if (valid && !mine) atomicAdd(&cellend[c], 1);
}

// Bin the cloud (n <= kBqCellsMaxPoints). Returns false (block-uniform) when the grid would be too
// coarse to prune or one cell holds a large share of the cloud (a sweep with its early exit is the
// better tool for both); `sorted` has not been written in that case.
// pos_tab (optional, n entries): point k's position in `sorted` -- the overlapped launch's consumers look the
// query point up by its index (the FPS sample) without a trip to global memory.
// BLK (round 6): CROWDED BALLS. The output is the nsample SMALLEST indices of a ball. When a ball holds several times nsample
// points (a uniform cube at r = 0.2: 137 of 4096 for nsample 32) the list's 27 cells are ~885 candidates and the sweep reads
// ~960 until its 32nd hit -- neither prunes. With the cloud sorted by (index block, cell), block = k >= n / 2, a query walks the
// first block's cells (~440 candidates, ~68 hits) and never reads the second: every hit below n / 2 is in the first block, so
// nsample hits there ARE the answer. Decided per workgroup after the count pass from the occupancy: expected hits per ball =
// (n / non-empty cells) x ball volume / cell volume >= 3 nsample (and both blocks' ends fit the table: <= 864 cells); the two
// halves of the count table are merged otherwise and everything is as without BLK. crowd_nsample <= 0: never.
template <int NT, bool BLK = false>
__device__ __forceinline__ bool bq_build_grid(int n, float reach, const float *__restrict__ data, float4 *sorted,
                                              int *cellend, float *misc, BqGrid &g, unsigned short *pos_tab = nullptr,
                                              int crowd_nsample = 0)
{
    constexpr int kBqPtsPerThread = kBqCellsMaxPoints / NT, kBqWavesT = NT / 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float inf = INFINITY;
    // the thread's points live in registers for all three passes: one trip to memory
    float px[kBqPtsPerThread], py[kBqPtsPerThread], pz[kBqPtsPerThread];
#pragma unroll
    for (int i = 0; i < kBqPtsPerThread; ++i) {
        px[i] = py[i] = pz[i] = 0.0f;
        if (i * NT < n) {                                        // block-uniform: slots past n cost nothing
            const int k = t + i * NT;
            const float *p = data + (size_t)(k < n ? k : 0) * 3;
            px[i] = p[0]; py[i] = p[1]; pz[i] = p[2];
        }
    }
    float lx = inf, ly = inf, lz = inf, hx = -inf, hy = -inf, hz = -inf;
    // copies of point 0 (the reference's dropout augmentation sets up to 87.5 % of a cloud to its first point,
    // provider.py:227-233): counted on the way, so that such a cloud leaves for the sweep BEFORE the counting sort
    // instead of after it (the crowded-cell test below would send it there anyway, 7 us later)
    const float p0x = data[0], p0y = data[1], p0z = data[2];
    int dups = 0;
#pragma unroll
    for (int i = 0; i < kBqPtsPerThread; ++i) {
        if (t + i * NT < n) {
            const float x = px[i], y = py[i], z = pz[i];
            dups += (x == p0x && y == p0y && z == p0z) ? 1 : 0;
            if (fabsf(x) < inf) { lx = fminf(lx, x); hx = fmaxf(hx, x); }     // non-finite coordinates never hit
            if (fabsf(y) < inf) { ly = fminf(ly, y); hy = fmaxf(hy, y); }
            if (fabsf(z) < inf) { lz = fminf(lz, z); hz = fmaxf(hz, z); }
        }
    }
    lx = wave_minmax_lane63<false>(lx); ly = wave_minmax_lane63<false>(ly); lz = wave_minmax_lane63<false>(lz);
    hx = wave_minmax_lane63<true>(hx); hy = wave_minmax_lane63<true>(hy); hz = wave_minmax_lane63<true>(hz);
    dups = wave_prefix_sum_incl(dups);                           // lane 63: the wave's count
    // scratch layout: float[6][16] wave partials (read back as float4: 96 scalar LDS reads per thread
    // were a visible part of the binning time), then int[16] wave sums and int[16] wave maxima, int[16] copy counts
    int *dupw = reinterpret_cast<int *>(misc + 8 * 16);
    if (lane == 63) {
        misc[0 * 16 + w] = lx; misc[1 * 16 + w] = ly; misc[2 * 16 + w] = lz;
        misc[3 * 16 + w] = hx; misc[4 * 16 + w] = hy; misc[5 * 16 + w] = hz;
        dupw[w] = dups;
    }
    __syncthreads();
    {
        int total = 0;
        const int4 *d4 = reinterpret_cast<const int4 *>(dupw);
#pragma unroll
        for (int v = 0; v < kBqWavesT / 4; ++v) {
            const int4 a = d4[v];
            total += a.x + a.y + a.z + a.w;
        }
        if (total > max(128, n / 16)) return false;             // block-uniform: one cell would hold all of them
    }
    lx = ly = lz = inf;
    hx = hy = hz = -inf;
    {
        const float4 *m4 = reinterpret_cast<const float4 *>(misc);
#pragma unroll
        for (int v = 0; v < kBqWavesT / 4; ++v) {
            const float4 a = m4[0 * 4 + v], b = m4[1 * 4 + v], c = m4[2 * 4 + v];
            const float4 d = m4[3 * 4 + v], e = m4[4 * 4 + v], f = m4[5 * 4 + v];
            lx = fminf(fminf(lx, fminf(a.x, a.y)), fminf(a.z, a.w));
            ly = fminf(fminf(ly, fminf(b.x, b.y)), fminf(b.z, b.w));
            lz = fminf(fminf(lz, fminf(c.x, c.y)), fminf(c.z, c.w));
            hx = fmaxf(fmaxf(hx, fmaxf(d.x, d.y)), fmaxf(d.z, d.w));
            hy = fmaxf(fmaxf(hy, fmaxf(e.x, e.y)), fmaxf(e.z, e.w));
            hz = fmaxf(fmaxf(hz, fmaxf(f.x, f.y)), fmaxf(f.z, f.w));
        }
    }
    if (!(lx <= hx)) lx = hx = 0.0f;
    if (!(ly <= hy)) ly = hy = 0.0f;
    if (!(lz <= hz)) lz = hz = 0.0f;
    // cell edge: at least the reach (so a ball spans <= 3 cells per axis), at least extent/kBqGridMax
    const float ex = hx - lx, ey = hy - ly, ez = hz - lz;
    // reach < 0 (three_nn's cell list, interpolate.hip): no radius is given -- the edge is |reach| times the mean spacing of
    // the points in their bounding box, cbrt(volume / n); a flat or degenerate cloud (volume 0) gets extent / kBqGridMax
    if (reach < 0.0f) reach = -reach * cbrtf(fmaxf(ex * ey * ez, 0.0f) / (float)n);
    const float edge_min = reach * 1.0001f;
    const float cx = fmaxf(edge_min, ex * (1.0f / kBqGridMax));
    const float cy = fmaxf(edge_min, ey * (1.0f / kBqGridMax));
    const float cz = fmaxf(edge_min, ez * (1.0f / kBqGridMax));
    auto dim = [](float ext, float edge) {
        const float q = ext / edge;                              // NaN (inf/inf, 0/0) -> 1 cell
        return (q >= 0.0f) ? min(kBqGridMax, (int)fminf(q, (float)kBqGridMax) + 1) : 1;
    };
    g.ox = lx; g.oy = ly; g.oz = lz;
    g.ix = 1.0f / cx; g.iy = 1.0f / cy; g.iz = 1.0f / cz;
    g.gx = dim(ex, cx); g.gy = dim(ey, cy); g.gz = dim(ez, cz);
    g.nb = 1;
    const int ncells1 = g.gx * g.gy * g.gz;
    if (ncells1 < kBqMinCells) return false;
    const bool two = BLK && crowd_nsample > 0 && 2 * ncells1 + 1 <= kBqTabInts - 1;   // block-uniform: count per (block, cell)
    const int nhalf = n >> 1;

    for (int c = t; c < (two ? 2 * ncells1 : ncells1); c += NT) cellend[c] = 0;
    __syncthreads();
    int cell[kBqPtsPerThread];
#pragma unroll
    for (int i = 0; i < kBqPtsPerThread; ++i) {
        cell[i] = 0;
        if (i * NT < n) {                                        // block-uniform
            cell[i] = (bq_cell(pz[i], g.oz, g.iz, g.gz) * g.gy + bq_cell(py[i], g.oy, g.iy, g.gy)) * g.gx +
                      bq_cell(px[i], g.ox, g.ix, g.gx);
            if (BLK && two && t + i * NT >= nhalf) cell[i] += ncells1;
            bq_cell_count(cellend, cell[i], t + i * NT < n, lane);
        }
    }
    __syncthreads();
    int *wsum = reinterpret_cast<int *>(misc + 6 * 16);
    int *wmax = wsum + 16;
    if (BLK && two) {
        // occupancy of the cells (both halves together) -> are the balls crowded? Otherwise the halves are merged.
        int *wne = reinterpret_cast<int *>(misc + 9 * 16);
        const int per1 = (ncells1 + NT - 1) / NT;
        const int a0 = min(ncells1, t * per1), a1 = min(ncells1, a0 + per1);
        int ne = 0;
        for (int c = a0; c < a1; ++c) ne += (cellend[c] + cellend[ncells1 + c]) > 0 ? 1 : 0;
        ne = wave_prefix_sum_incl(ne);
        if (lane == 63) wne[w] = ne;
        __syncthreads();
        int nonempty = 0;
#pragma unroll
        for (int v = 0; v < kBqWavesT; ++v) nonempty += wne[v];
        const float ball = 4.18879f * reach * reach * reach * g.ix * g.iy * g.iz;        // ball volume / cell volume
        const float hits = (float)n / (float)max(nonempty, 1) * ball;
        const bool crowded = hits >= 3.0f * (float)crowd_nsample;
        if (__builtin_amdgcn_readfirstlane((int)crowded)) {
            g.nb = 2;
        } else {
            for (int c = a0; c < a1; ++c) cellend[c] += cellend[ncells1 + c];
#pragma unroll
            for (int i = 0; i < kBqPtsPerThread; ++i)
                if (t + i * NT >= nhalf) cell[i] -= (i * NT < n) ? ncells1 : 0;
            __syncthreads();
        }
    }
    const int ncells = g.nb * ncells1;                           // entries of the table: (block, cell)
    // exclusive scan of the counts: a contiguous slab of cells per thread, wave scan, wave offsets
    const int per = (ncells + NT - 1) / NT;
    const int c0 = min(ncells, t * per), c1 = min(ncells, c0 + per);
    int sum = 0, big = 0;
    for (int c = c0; c < c1; ++c) {
        const int v = cellend[c];
        sum += v;
        big = max(big, v);
    }
    const int incl = wave_prefix_sum_incl(sum);
    big = wave_max_i32_lane63(big);
    if (lane == 63) { wsum[w] = incl; wmax[w] = big; }
    __syncthreads();
    int run = incl - sum;
    {
        const int4 *s4 = reinterpret_cast<const int4 *>(wsum), *x4 = reinterpret_cast<const int4 *>(wmax);
#pragma unroll
        for (int v = 0; v < kBqWavesT / 4; ++v) {
            const int4 a = s4[v], b = x4[v];
            run += (4 * v + 0 < w ? a.x : 0) + (4 * v + 1 < w ? a.y : 0) + (4 * v + 2 < w ? a.z : 0) + (4 * v + 3 < w ? a.w : 0);
            big = max(max(big, max(b.x, b.y)), max(b.z, b.w));
        }
    }
    // a crowded cell means crowded balls (the sweep's early exit is at its best) and a contended scatter
    if (big > max(128, n / 16)) return false;                    // block-uniform
    for (int c = c0; c < c1; ++c) {
        const int cnt = cellend[c];
        cellend[c] = run;                                        // start of the cell: the scatter cursor
        run += cnt;
    }
    __syncthreads();
    int pos[kBqPtsPerThread];
#pragma unroll
    for (int i = 0; i < kBqPtsPerThread; ++i)
        if (t + i * NT < n) pos[i] = atomicAdd(&cellend[cell[i]], 1);
#pragma unroll
    for (int i = 0; i < kBqPtsPerThread; ++i) {
        const int k = t + i * NT;
        if (k < n) {
            sorted[pos[i]] = make_float4(px[i], py[i], pz[i], __int_as_float(k));
            if (pos_tab) pos_tab[k] = (unsigned short)pos[i];
        }
    }
    __syncthreads();                                             // cellend[c] is now the END of cell c
    return true;
}

// Inclusive prefix sum inside aligned groups of LPQ lanes (DPP row shifts; rows are 16 lanes).
template <int LPQ>
__device__ __forceinline__ int group_prefix_sum_incl(int v, int sub)
{
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); v += (LPQ >= 16 || sub >= 1) ? t : 0;   // row_shr:1
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true); v += (LPQ >= 16 || sub >= 2) ? t : 0;   // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); v += (LPQ >= 16 || sub >= 4) ? t : 0;   // row_shr:4
    if (LPQ >= 16) { t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true); v += t; }              // row_shr:8
    if (LPQ >= 32) { t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); v += t; }             // row_bcast:15
    if (LPQ >= 64) { t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); v += t; }             // row_bcast:31
    return v;
}

// The queries of one workgroup against its binned cloud. A wave works on G = 64 / LPQ queries at a
// time, LPQ lanes each: the per-query arithmetic (cell ranges, run bounds, prefix sums) is then done
// by all lanes at once for G queries instead of being wave-uniform work, which is what bounds the
// one-query-per-wave form (~500 issue slots per query, most of them scalar-like bookkeeping).
// A query's candidates are the <= 3x3 runs of x-adjacent cells around it; its LPQ lanes walk them run
// by run. Hits go into the query's bitmap (bit k = point k) with LDS atomic ORs; the bitmap is then
// turned into the ascending index list by a group prefix sum of popcounts and a bit-peeling loop.
template <int NT, int LPQ, bool FUSE, bool POLL, bool BLK = false>
__device__ __forceinline__ bool bq_cells_query_loop(int n, int m, int nsample, float thr, float radius, float reach,
                                                    int bi, int q0, int q1, const BqGrid &g,
                                                    const float *__restrict__ data,
                                                    const float *__restrict__ xyz2,
                                                    const unsigned long long *__restrict__ tagged,
                                                    float *__restrict__ new_xyz, int *__restrict__ idx,
                                                    int *__restrict__ pts_cnt, float *__restrict__ grouped,
                                                    int subtract, const float4 *sorted, const int *tab,
                                                    char *wave_area, size_t wave_stride = 0, unsigned tag = 1u,
                                                    unsigned *status = nullptr, const unsigned short *pos_tab = nullptr)
{
    constexpr int G = 64 / LPQ;
    constexpr int CW = 64 / LPQ;                                 // 64-bit bitmap words per lane and window
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = lane / LPQ, sub = lane % LPQ;
    const int nwin = (n + 4095) / 4096;
    const int gwords = nwin * 128;                               // 32-bit words of one query's bitmap
    // wave_stride: bytes between the private areas of two waves (the multi-radius kernel passes the size its
    // largest nsample needs, so that waves working on different radii never overlap)
    if (wave_stride == 0) wave_stride = bq_cells_wave_bytes(n, nsample, LPQ);
    unsigned *bm = reinterpret_cast<unsigned *>(wave_area + (size_t)w * wave_stride);
    unsigned *bmq = bm + grp * gwords;
    int *rowflat = reinterpret_cast<int *>(bm + G * gwords);      // the wave's G row buffers, back to back
    int *rowbuf = rowflat + grp * nsample;
    float4 *qhdr = reinterpret_cast<float4 *>(rowflat + ((G * nsample + 3) & ~3));   // per query: centroid + hit count (flat emit)
    const bool pow2 = (nsample & (nsample - 1)) == 0;
    const int lg = 31 - __builtin_clz((unsigned)nsample);
    for (int i = lane; i < G * gwords; i += 64) bm[i] = 0u;
    // |q| beyond this and q -/+ reach may round past an in-ball point: such queries visit everything
    const float lim = radius * 4096.0f;

    float nqx = 0.f, nqy = 0.f, nqz = 0.f;                       // next query, loaded one trip ahead
    if (!POLL) {
        const int j = min(q0 + w * G + grp, q1 - 1);
        const float *q = xyz2 + ((size_t)bi * m + j) * 3;
        nqx = q[0]; nqy = q[1]; nqz = q[2];
    }
    for (int jb = q0 + w * G; jb < q1; jb += (NT / 64) * G) {
        const int j = jb + grp;
        const bool qvalid = j < q1;
        const size_t row = (size_t)bi * m + (qvalid ? j : q1 - 1);
        float qx, qy, qz;
        if (POLL) {
            const int ka = bq_poll_sample(tagged + row, tag);
            if (__any(ka < 0)) {                                 // see bq_poll_sample: report, give up
                if (lane == 0 && status) atomicExch(status, 1u);
                return false;
            }
            const float4 qp = sorted[pos_tab[ka]];               // POLL implies pos_tab: exact copy of xyz[ka]
            qx = qp.x; qy = qp.y; qz = qp.z;
            if (sub == 0 && qvalid) {
                float *o = new_xyz + row * 3;
                o[0] = qx; o[1] = qy; o[2] = qz;
            }
        } else {
            qx = nqx; qy = nqy; qz = nqz;
            if (jb + (NT / 64) * G < q1) {
                const int jn = min(j + (NT / 64) * G, q1 - 1);
                const float *q = xyz2 + ((size_t)bi * m + jn) * 3;
                nqx = q[0]; nqy = q[1]; nqz = q[2];
            }
        }
        const int cx0 = bq_cell(__fsub_rn(qx, reach), g.ox, g.ix, g.gx), cx1 = bq_cell(__fadd_rn(qx, reach), g.ox, g.ix, g.gx);
        const int cy0 = bq_cell(__fsub_rn(qy, reach), g.oy, g.iy, g.gy), cy1 = bq_cell(__fadd_rn(qy, reach), g.oy, g.iy, g.gy);
        const int cz0 = bq_cell(__fsub_rn(qz, reach), g.oz, g.iz, g.gz), cz1 = bq_cell(__fadd_rn(qz, reach), g.oz, g.iz, g.gz);
        const bool near_origin = fabsf(qx) <= lim && fabsf(qy) <= lim && fabsf(qz) <= lim;   // false for NaN
        // Up to nine candidate runs per query. NARROW (the ball spans <= 3 cells in y and z: always the case
        // when the grid was built for this radius): run r = the x-adjacent cells cx0..cx1 of row (cy0 + r%3,
        // cz0 + r/3). WIDE (a radius larger than the cell edge -- the multi-radius kernel bins once, at the
        // smallest radius): run r = the whole y-range cy0..cy1, every x, of slab cz0 + r; more candidates per
        // run, still one contiguous piece of the sorted array each. Anything else visits the whole array --
        // slow, never wrong (also the path of far-away and NaN queries).
        const int sy = cy1 - cy0, sz = cz1 - cz0;                // spans - 1
        const bool ordered = cy1 >= cy0 && cz1 >= cz0 && cx1 >= cx0;
        const bool narrow = ordered && sy <= 2 && sz <= 2;
        const bool wide = ordered && !narrow && sz <= 8;
        const bool everything = !near_origin || !(narrow || wide);
        const int dx1 = cx1 - cx0 + 1;
        // BLK: the list is sorted by (index block, cell); a query walks block 0 and goes on to block 1 only while it holds fewer
        // than nsample hits (all of a ball's hits below n / 2 are in block 0, so nsample of them are the nsample smallest)
        const int nblk = BLK ? __builtin_amdgcn_readfirstlane(g.nb) : 1;
        const int ncell1 = g.gx * g.gy * g.gz;
        int hc = 0;                                              // BLK: this lane's hits so far
        bool more = true;                                        // BLK: this query still needs candidates
        for (int blk = 0; blk < nblk; ++blk) {
        const int toff = BLK ? blk * ncell1 : 0;
        int rb[9], re[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int ry = r % 3, rz = r / 3;
            const bool rvalid = qvalid && more && !everything && (narrow ? (ry <= sy && rz <= sz) : (r <= sz));
            const int nfirst = ((cz0 + rz) * g.gy + (cy0 + ry)) * g.gx + cx0;          // narrow: cells [nfirst, nfirst + dx1)
            const int wfirst = ((cz0 + r) * g.gy + cy0) * g.gx;                         // wide: cells [wfirst, wlast)
            const int wlast = ((cz0 + r) * g.gy + cy1) * g.gx + g.gx;
            const int cfirst = rvalid ? (narrow ? nfirst : wfirst) : 0;
            const int clast = rvalid ? (narrow ? nfirst + dx1 : wlast) : 0;
            const int b0 = tab[toff + cfirst], e0 = tab[toff + clast];
            rb[r] = rvalid ? b0 : 0;
            re[r] = rvalid ? e0 : 0;
        }
        if (everything && qvalid && more) { rb[0] = BLK ? tab[toff] : 0; re[0] = BLK ? tab[toff + ncell1] : n; }
        // The nine runs are walked as ONE candidate stream per query (flat position f -> run by a compare
        // chain against the running lengths; all lanes of a group hold the same bounds). Walking run by
        // run in lockstep left three quarters of the lane slots empty (runs are ~10 points, a group has 8
        // lanes, and the trip count is the maximum over the wave's 8 queries). Four reads in flight per lane.
        int cum[10], adj[9];
        cum[0] = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            cum[r + 1] = cum[r] + (re[r] - rb[r]);
            adj[r] = rb[r] - cum[r];
        }
        const int ncand = cum[9];
        for (int f0 = sub; __any(f0 < ncand); f0 += 4 * LPQ) {
            float4 p[4];
            bool act[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + u * LPQ;
                act[u] = f < ncand;
                int sel = adj[0];
#pragma unroll
                for (int r = 1; r < 9; ++r) sel = (f >= cum[r]) ? adj[r] : sel;
                p[u] = sorted[act[u] ? f + sel : 0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // reference operand order: (x2-x1) with x2 the query (query_ball_point.cpp:26-32)
                const float s = sqdist(qx, qy, qz, p[u].x, p[u].y, p[u].z);
                const int k = __float_as_int(p[u].w);
                if (act[u] && s < thr) {
                    atomicOr(&bmq[k >> 5], 1u << (k & 31));
                    if (BLK) ++hc;
                }
            }
        }
        if (BLK && blk + 1 < nblk) {                             // hits of the group so far (wave-uniform trip: every lane takes part)
            const int incl = group_prefix_sum_incl<LPQ>(hc, sub);
            more = __shfl(incl, lane | (LPQ - 1)) < nsample;
            if (!__any(more && qvalid)) break;
        }
        }
        asm volatile("" ::: "memory");                           // LDS ops of one wave execute in order
        int done = 0;                                            // hits of this query so far (group-uniform)
        for (int win = 0; win < nwin; ++win) {
            unsigned long long *wp = reinterpret_cast<unsigned long long *>(bmq + win * 128) + sub * CW;
            unsigned long long wd[CW];
            int ps[CW + 1];
            ps[0] = 0;
#pragma unroll
            for (int i = 0; i < CW; ++i) {
                wd[i] = wp[i];
                wp[i] = 0ull;
                ps[i + 1] = ps[i] + __popcll(wd[i]);
            }
            const int pc = ps[CW];
            const int incl = group_prefix_sum_incl<LPQ>(pc, sub);
            const int total = __shfl(incl, lane | (LPQ - 1));
            const int start = done + incl - pc;
            // every word knows its first output slot, so the words peel independently: three bits each in
            // straight-line code (covers almost every word at the usual hit densities), then a loop for
            // whatever is left
#pragma unroll
            for (int i = 0; i < CW; ++i) {
                const int kbase = win * 4096 + (sub * CW + i) * 64;
                int pos = start + ps[i];
                unsigned long long word = wd[i];
#pragma unroll
                for (int step = 0; step < 3; ++step) {
                    if (word != 0ull && pos < nsample) rowbuf[pos] = kbase + __builtin_ctzll(word);
                    word &= word - 1ull;
                    ++pos;
                }
                wd[i] = word;
                ps[i] = pos;
            }
#pragma unroll
            for (int i = 0; i < CW; ++i) {
                const int kbase = win * 4096 + (sub * CW + i) * 64;
                while (__any(wd[i] != 0ull && ps[i] < nsample)) {
                    if (wd[i] != 0ull) {
                        if (ps[i] < nsample) rowbuf[ps[i]] = kbase + __builtin_ctzll(wd[i]);
                        wd[i] &= wd[i] - 1ull;
                        ++ps[i];
                    }
                }
            }
            done += total;
        }
        const int cnt = min(done, nsample);
        asm volatile("" ::: "memory");
        if (pow2) {
            // FLAT emit: the wave's G queries are consecutive rows, so their idx / grouped rows are ONE contiguous
            // block of G * nsample entries. Lane e handles entry e: every store instruction of the wave writes
            // 256 (idx) / 768 (grouped) contiguous bytes, whatever nsample is -- the per-query form below has
            // each group of LPQ lanes write its own 4 * LPQ-byte piece of a different row.
            if (sub == 0) qhdr[grp] = make_float4(qx, qy, qz, __int_as_float(cnt));
            asm volatile("" ::: "memory");
            const int nvalid = min(G, q1 - jb);                      // valid queries are a prefix of the wave's G
            const size_t row0 = (size_t)bi * m + jb;
            const int total = nvalid << lg;
            for (int e = lane; e < total; e += 64) {
                const int gq = e >> lg, l = e & (nsample - 1);
                const float4 hd = qhdr[gq];
                const int c = __float_as_int(hd.w);
                const int v = (l < c) ? rowflat[e] : (c > 0 ? rowflat[gq << lg] : 0);   // pad with the first hit; zeros when empty
                if (idx) idx[row0 * nsample + e] = v;
                if (FUSE && grouped) {
                    float gx = data[(size_t)v * 3 + 0], gy = data[(size_t)v * 3 + 1], gz = data[(size_t)v * 3 + 2];
                    if (subtract) { gx = __fsub_rn(gx, hd.x); gy = __fsub_rn(gy, hd.y); gz = __fsub_rn(gz, hd.z); }
                    float *o = grouped + (row0 * nsample + e) * 3;
                    o[0] = gx; o[1] = gy; o[2] = gz;
                }
            }
            if (pts_cnt && lane < nvalid) pts_cnt[row0 + lane] = __float_as_int(qhdr[lane].w);
        } else if (qvalid) {
            const int first = cnt > 0 ? rowbuf[0] : 0;           // the first hit pads the row; zeros when empty
            for (int l = sub; l < nsample; l += LPQ) {
                const int v = (l < cnt) ? rowbuf[l] : first;
                if (idx) idx[row * nsample + l] = v;
                if (FUSE && grouped) {
                    float gx = data[(size_t)v * 3 + 0], gy = data[(size_t)v * 3 + 1], gz = data[(size_t)v * 3 + 2];
                    if (subtract) { gx = __fsub_rn(gx, qx); gy = __fsub_rn(gy, qy); gz = __fsub_rn(gz, qz); }
                    float *o = grouped + (row * nsample + l) * 3;
                    o[0] = gx; o[1] = gy; o[2] = gz;
                }
            }
            if (sub == 0 && pts_cnt) pts_cnt[row] = cnt;
        }
        asm volatile("" ::: "memory");
    }
    return true;
}

// Cell-list workgroup body with the brute-force sweep as the block-uniform fallback (coarse grid or
// crowded cells). `radius` is the caller's radius; reach = radius * 1.001f exceeds the largest
// |coordinate difference| of any hit (<= radius * (1 + 4 ulp)) by 0.1 %.
// q_stride > 0: the workgroup is PERSISTENT -- after [q0, q1) it goes on to [q0 + q_stride, q1 + q_stride), ... up to m with
// the cloud staged and binned ONCE (the overlapped launch's consumers: a few workgroups per cloud walk the query ranges in
// the order the samples are published).
template <int NT, int LPQ, bool FUSE, bool POLL, bool BLK = false>
__device__ __forceinline__ void bq_cells_block_body(int n, int m, int nsample, float thr, float radius, int bi, int q0,
                                                    int q1, const float *__restrict__ xyz1,
                                                    const float *__restrict__ xyz2,
                                                    const unsigned long long *__restrict__ tagged,
                                                    float *__restrict__ new_xyz, int *__restrict__ idx,
                                                    int *__restrict__ pts_cnt, float *__restrict__ grouped,
                                                    int subtract, char *smem, unsigned tag = 1u,
                                                    unsigned *status = nullptr, int q_stride = 0)
{
    float4 *sorted = reinterpret_cast<float4 *>(smem);
    int *tab = reinterpret_cast<int *>(smem + sizeof(float4) * (size_t)n);
    char *wave_area = reinterpret_cast<char *>(tab + kBqTabInts);
    float *misc = reinterpret_cast<float *>(wave_area + (size_t)(NT / 64) * bq_cells_wave_bytes(n, nsample, LPQ));
    // POLL (the overlapped launch's consumers): index -> sorted position table behind the scratch (2 bytes per point)
    unsigned short *pos_tab = POLL ? reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(misc) + kBqMiscBytes) : nullptr;
    const float *__restrict__ data = xyz1 + (size_t)bi * n * 3;
    const float reach = radius * 1.001f;
    const int qlen = q1 - q0;
    BqGrid g;
    if (threadIdx.x == 0) tab[0] = 0;
    if (bq_build_grid<NT, BLK>(n, reach, data, sorted, tab + 1, misc, g, pos_tab, BLK ? nsample : 0)) {
        for (int a = q0; a < m; a += q_stride) {
            if (!bq_cells_query_loop<NT, LPQ, FUSE, POLL, BLK>(n, m, nsample, thr, radius, reach, bi, a, min(a + qlen, m), g, data, xyz2, tagged,
                                                       new_xyz, idx, pts_cnt, grouped, subtract, sorted, tab, wave_area, 0, tag,
                                                       status, pos_tab)) return;
            if (q_stride <= 0) break;
        }
    } else {
        __syncthreads();                                         // scratch was read by everyone before it is reused
        if (!bq_block_body<true, FUSE, POLL, NT>(n, m, nsample, thr, bi, q0, q1, xyz1, xyz2, tagged, new_xyz, idx, pts_cnt,
                                                 grouped, subtract, smem, tag, 0, status)) return;
        if (q_stride > 0)
            for (int a = q0 + q_stride; a < m; a += q_stride)   // the LDS copy of the cloud stays
                if (!bq_block_body<true, FUSE, POLL, NT, true>(n, m, nsample, thr, bi, a, min(a + qlen, m), xyz1, xyz2, tagged, new_xyz,
                                                               idx, pts_cnt, grouped, subtract, smem, tag, 0, status)) return;
    }
}

}  // namespace pn2
