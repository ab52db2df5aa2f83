// ball_query.hip -- query_ball_point (+ optional fused group of xyz) for gfx950.
//
// Replaces query_ball_point_gpu / queryBallPointLauncher (reference
// tf_ops/grouping/tf_grouping_g.cu:3-36, :125-128; CPU twin
// tf_ops/grouping/test/query_ball_point.cpp:19-47). Index-exact:
//   * the reference predicate max(sqrtf(s),1e-20f) < radius is evaluated as
//     s < s*, with s* the smallest fp32 whose predicate is false, found on the
//     host by bisection with the host's correctly rounded sqrtf
//     (pn2_ball_threshold). sqrtf is monotone, so the two are identical for
//     every s, and no device sqrt (whatever its rounding) is involved;
//   * s itself is the fp32 value ((dx*dx)+(dy*dy))+(dz*dz), no FMA.
//
// Design (DESIGN.md "ball query"). The reference walks the cloud with ONE
// thread per query. Here one 64-lane wave owns a query and sweeps the cloud 64
// candidates at a time in ascending index order: the cloud is staged once per
// workgroup in LDS as float4 (one ds_read_b128 per lane per step), the hit
// mask is a v_cmp (ballot), v_mbcnt gives each hit its ordered output slot,
// and the sweep stops (wave-uniformly) as soon as nsample hits are in. A wave
// sweeps TWO queries at once over 128 candidates per trip, so every LDS read
// feeds two distance chains and four independent chains are in flight per lane
// (the one-query, one-chunk first version was latency bound: 123 us at the
// metric shape). The result row is assembled in an LDS row buffer and leaves the CU as ONE
// coalesced store per query (the reference scatters 4-byte stores). With
// FUSE the same epilogue also emits xyz1[idx]-centroid straight from the LDS
// copy of the cloud (pointnet_util.py:44-46 needs three ops and two extra
// passes over idx for this).
//
// Two kernels share the exact predicate and the output stage. The SWEEP kernel above is O(m*n) with
// an early exit; the CELL-LIST kernel (ball_query_cells_kernel, device code in the second half of
// ball_query_body.h) bins the cloud per workgroup and visits ~27 cells per query: 57 -> 24 us at the
// metric shape. bq_use_cells() picks by shape; the cell-list kernel falls back to the sweep body by
// itself when the data make the grid useless (coarse grid, crowded cells).
#include "ball_query_body.h"

#include <limits.h>
#include <math.h>
#include <string.h>

namespace pn2 {

template <bool LDS_CLOUD, bool FUSE>
__global__ __launch_bounds__(kBqThreads) void ball_query_kernel(int n, int m, int nsample, float thr, int qpb,
                                                                const float *__restrict__ xyz1,
                                                                const float *__restrict__ xyz2, int *__restrict__ idx,
                                                                int *__restrict__ pts_cnt,
                                                                float *__restrict__ grouped, int subtract)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int q0 = blockIdx.x * qpb;
    bq_block_body<LDS_CLOUD, FUSE, false>(n, m, nsample, thr, blockIdx.y, q0, min(q0 + qpb, m), xyz1, xyz2, nullptr,
                                          nullptr, idx, pts_cnt, grouped, subtract, smem);
}

template <bool LDS_CLOUD, bool FUSE>
static int launch_bq(int b, int n, int m, float thr, int nsample, const float *xyz1, const float *xyz2, int *idx,
                     int *pts_cnt, float *grouped, int subtract, hipStream_t st)
{
    // aim at ~2 workgroups per CU over the whole launch; each stages the cloud once
    constexpr int kGran = kBqWaves * kBqQpw;
    long long total = (long long)b * m;
    int qpb = (int)((total + 511) / 512);
    qpb = ((qpb + kGran - 1) / kGran) * kGran;
    if (qpb < kGran) qpb = kGran;
    if (qpb > m) qpb = ((m + kGran - 1) / kGran) * kGran;
    const int gx = (m + qpb - 1) / qpb;
    const size_t lds = (LDS_CLOUD ? sizeof(float4) * (size_t)((n + 127) & ~127) : 0) + sizeof(int) * (size_t)nsample * kGran;
    if (lds > 160 * 1024) return PN2_E_TOO_LARGE;
    auto kern = ball_query_kernel<LDS_CLOUD, FUSE>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    if (int rc = launch(kern, dim3(gx, b), dim3(kBqThreads), lds, st, n, m, nsample, thr, qpb, xyz1, xyz2, idx,
                       pts_cnt, grouped, subtract)) return rc;
    return PN2_OK;
}

// Cell-list kernel (ball_query_body.h, second half): the grid is built per workgroup, so a workgroup
// takes a larger share of the queries than in the sweep kernel to amortise the binning pass.
template <int NT, int LPQ, bool FUSE>
__global__ __launch_bounds__(NT, NT / 256) void ball_query_cells_kernel(int b, int n, int m, int nsample, float thr,
                                                                      float radius, int qpb, int parts,
                                                                      const float *__restrict__ xyz1,
                                                                      const float *__restrict__ xyz2,
                                                                      int *__restrict__ idx,
                                                                      int *__restrict__ pts_cnt,
                                                                      float *__restrict__ grouped, int subtract)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int cloud, part;
    decode_cloud_block(blockIdx.x, parts, b, cloud, part);
    const int q0 = part * qpb;
    bq_cells_block_body<NT, LPQ, FUSE, false, true>(n, m, nsample, thr, radius, cloud, q0, min(q0 + qpb, m), xyz1, xyz2,   // BLK: crowded balls walk the first half of the indices first (ball_query_body.h)
                                     nullptr, nullptr, idx, pts_cnt, grouped, subtract, smem);
}

static size_t bq_sweep_lds_bytes(int n, int nsample, int nthreads = kBqThreads)
{
    return sizeof(float4) * (size_t)((n + 127) & ~127) + sizeof(int) * (size_t)nsample * (nthreads / 64) * kBqQpw;
}

// Per-call dispatch options (pn2_query_ball_group_xyz_ex; the reference-shaped entry points pass {0, 0}):
// kernel 0 = automatic, 1 = sweep kernel only, 2 = cell-list kernel whenever it fits, 3 = same with
// 512-thread workgroups only; qpb = queries per workgroup of the cell-list kernel, 0 = automatic.
struct BqOpts { int kernel, qpb; };

// Geometry of the cell-list kernel: workgroup size and lanes per query. The query loop is branchy,
// LDS-latency-bound code that wants 16 waves on a CU: two 512-thread workgroups when two fit in the
// LDS (small clouds), one 1024-thread workgroup otherwise; and a wave should carry as many queries
// as the LDS holds bitmaps for. Returns false when nothing fits (the sweep kernel is used).
struct BqCellsGeom { int nthreads, lpq; };
static bool bq_cells_pick(int n, int nsample, int kernel, BqCellsGeom &g)
{
    static const BqCellsGeom order[] = {{512, 8}, {512, 16}, {1024, 8}, {1024, 16}, {512, 8}, {512, 16}, {512, 32}};
    for (int i = 0; i < 7; ++i) {
        const BqCellsGeom &c = order[i];
        if (kernel == 3 && c.nthreads != 512) continue;
        const size_t cap = (i < 2 && kernel != 3) ? 80 * 1024 : 160 * 1024;   // first two: only if two workgroups fit
        if (bq_cells_lds_bytes(n, nsample, c.lpq, c.nthreads) <= cap &&
            bq_sweep_lds_bytes(n, nsample, c.nthreads) <= cap) { g = c; return true; }
    }
    return false;
}

// The cell-list kernel pays a binning pass per workgroup and only prunes when the grid is fine; measured
// on MI355X (scripts/bq_probe.py) it wins from about 2048 points per cloud and a launch that fills the GPU.
static bool bq_use_cells(int b, int n, int m, int kernel)
{
    if (n > kBqCellsMaxPoints || n < 64 || kernel == 1) return false;
    if (kernel >= 2) return true;
    return n >= 2048 && (long long)b * m >= 8192;
}

template <int NT, int LPQ, bool FUSE>
static int launch_bq_cells(int b, int n, int m, float thr, float radius, int nsample, const float *xyz1,
                           const float *xyz2, int *idx, int *pts_cnt, float *grouped, int subtract, int opt_qpb,
                           hipStream_t st)
{
    constexpr int kGran = (NT / 64) * (64 / LPQ);               // queries per workgroup trip
    long long total = (long long)b * m;
    int qpb = opt_qpb > 0 ? opt_qpb : (int)((total + 255) / 256);
    qpb = ((qpb + kGran - 1) / kGran) * kGran;
    if (qpb < kGran) qpb = kGran;
    if (qpb > m) qpb = ((m + kGran - 1) / kGran) * kGran;
    const int gx = (m + qpb - 1) / qpb;
    const size_t a = bq_cells_lds_bytes(n, nsample, LPQ, NT), s = bq_sweep_lds_bytes(n, nsample, NT);
    const size_t lds = a > s ? a : s;                       // the fallback inside the kernel uses the sweep layout
    auto kern = ball_query_cells_kernel<NT, LPQ, FUSE>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    if (int rc = launch(kern, dim3((unsigned)gx * b), dim3(NT), lds, st, b, n, m, nsample, thr, radius, qpb, gx,
                       xyz1, xyz2, idx, pts_cnt, grouped, subtract)) return rc;
    return PN2_OK;
}

template <bool FUSE>
static int dispatch_bq_cells(BqCellsGeom g, int b, int n, int m, float thr, float radius, int nsample,
                             const float *xyz1, const float *xyz2, int *idx, int *pts_cnt, float *grouped,
                             int subtract, int qpb, hipStream_t st)
{
#define PN2_BQ_CASE(NT, LPQ) \
    if (g.nthreads == NT && g.lpq == LPQ) \
        return launch_bq_cells<NT, LPQ, FUSE>(b, n, m, thr, radius, nsample, xyz1, xyz2, idx, pts_cnt, grouped, subtract, qpb, st)
    PN2_BQ_CASE(1024, 8);
    PN2_BQ_CASE(1024, 16);
    PN2_BQ_CASE(512, 8);
    PN2_BQ_CASE(512, 16);
    PN2_BQ_CASE(512, 32);
#undef PN2_BQ_CASE
    return PN2_E_TOO_LARGE;
}

static int ball_query_common(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                             int subtract, int *idx, int *pts_cnt, float *grouped, bool fuse, BqOpts opt, void *stream)
{
    if (opt.kernel < 0 || opt.kernel > 3 || opt.qpb < 0) return PN2_E_ARG;
    if (!(radius > 0.0f) || nsample <= 0) return PN2_E_ARG;   // tf_grouping.cpp:71,74
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    if (b == 0 || m == 0) return PN2_OK;
    if (!xyz1 || !xyz2) return PN2_E_NULL;
    if (fuse ? !grouped : !idx) return PN2_E_NULL;
    if ((long long)b * n * 3 > INT_MAX || (long long)b * m * nsample * 3 > (1ll << 40)) return PN2_E_TOO_LARGE;
    if (b > 65535) return PN2_E_TOO_LARGE;
    const float thr = pn2_ball_threshold(radius);
    hipStream_t st = as_stream(stream);
    const bool lds = n <= kBqMaxLdsPoints &&
                     sizeof(float4) * (size_t)((n + 127) & ~127) + sizeof(int) * (size_t)nsample * kBqWaves * kBqQpw <= 160 * 1024;
    BqCellsGeom geom;
    if (bq_use_cells(b, n, m, opt.kernel) && bq_cells_pick(n, nsample, opt.kernel, geom))
        return fuse ? dispatch_bq_cells<true>(geom, b, n, m, thr, radius, nsample, xyz1, xyz2, idx, pts_cnt, grouped, subtract, opt.qpb, st)
                    : dispatch_bq_cells<false>(geom, b, n, m, thr, radius, nsample, xyz1, xyz2, idx, pts_cnt, nullptr, 0, opt.qpb, st);
    if (fuse)
        return lds ? launch_bq<true, true>(b, n, m, thr, nsample, xyz1, xyz2, idx, pts_cnt, grouped, subtract, st)
                   : launch_bq<false, true>(b, n, m, thr, nsample, xyz1, xyz2, idx, pts_cnt, grouped, subtract, st);
    return lds ? launch_bq<true, false>(b, n, m, thr, nsample, xyz1, xyz2, idx, pts_cnt, nullptr, 0, st)
               : launch_bq<false, false>(b, n, m, thr, nsample, xyz1, xyz2, idx, pts_cnt, nullptr, 0, st);
}

}  // namespace pn2

// Smallest fp32 s* for which max(sqrtf(s*),1e-20f) < radius is FALSE.
// The predicate is monotone non-increasing in s >= 0 (sqrtf is correctly rounded,
// hence monotone), so  in-ball <=> s < s*.  Bisection over the fp32 bit patterns.
extern "C" float pn2_ball_threshold(float radius)
{
    if (!(radius > 0.0f) || !(1e-20f < radius)) return 0.0f;   // nothing is ever inside
    uint32_t lo = 0u;           // s=+0: predicate true
    uint32_t hi = 0x7f800000u;  // s=+inf: predicate false
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        float s;
        memcpy(&s, &mid, sizeof s);
        const float r = sqrtf(s);
        const float d = (r < 1e-20f) ? 1e-20f : r;
        if (d < radius) lo = mid; else hi = mid;
    }
    float out;
    memcpy(&out, &hi, sizeof out);
    return out;
}

extern "C" int pn2_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                    const float *xyz2, int *idx, int *pts_cnt, void *stream)
{
    return pn2::ball_query_common(b, n, m, radius, nsample, xyz1, xyz2, 0, idx, pts_cnt, nullptr, false, {0, 0}, stream);
}

extern "C" int pn2_query_ball_group_xyz(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                        const float *xyz2, int subtract_centroid, int *idx, int *pts_cnt,
                                        float *grouped_xyz, void *stream)
{
    return pn2::ball_query_common(b, n, m, radius, nsample, xyz1, xyz2, subtract_centroid, idx, pts_cnt,
                                  grouped_xyz, true, {0, 0}, stream);
}

// The same operator with the kernel choice as per-call arguments (tests force every kernel at every shape,
// scripts/bq_probe.py tunes the dispatch): kernel 0 automatic, 1 sweep, 2 cell list, 3 cell list with
// 512-thread workgroups; cells_qpb = queries per workgroup of the cell-list kernel (0 automatic).
// grouped_xyz == NULL: plain query_ball_point. Stateless like everything else in the library.
extern "C" int pn2_query_ball_group_xyz_ex(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                           const float *xyz2, int subtract_centroid, int *idx, int *pts_cnt,
                                           float *grouped_xyz, int kernel, int cells_qpb, void *stream)
{
    return pn2::ball_query_common(b, n, m, radius, nsample, xyz1, xyz2, grouped_xyz ? subtract_centroid : 0, idx, pts_cnt,
                                  grouped_xyz, grouped_xyz != nullptr, {kernel, cells_qpb}, stream);
}
