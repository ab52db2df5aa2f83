// ball_query_msg.hip -- multi-radius ball query + grouping of xyz: ONE staging / binning of the cloud
// serves every radius of a multi-scale-grouping level.
//
// Reference: pointnet_sa_module_msg, utils/pointnet_util.py:175-186 -- for every radius of the level the
// loop calls query_ball_point(radius, nsample, xyz, new_xyz) and group_point(xyz, idx) and subtracts the
// centroid, i.e. it rescans the cloud once per radius (three launches of query_ball_point_gpu,
// tf_grouping_g.cu:3-36, over the same xyz / new_xyz). pointnet2_cls_msg.py:27 runs radii
// (0.1, 0.2, 0.4) x nsample (16, 32, 128) over 4096-point clouds.
//
// Here a workgroup owns (cloud, query range) for ALL radii. It bins the cloud once into the cell list
// of ball_query_body.h -- cells sized for the SMALLEST radius -- and then runs the query loop once per
// radius against the same LDS-resident, cell-sorted array: a larger radius simply visits a larger block
// of cells (the "wide" runs of bq_cells_query_loop). When the cell list does not apply (small clouds,
// coarse grid, crowded cells) the cloud is staged once in index order and swept per radius. Both paths
// use the device bodies of the single-radius kernels, so every output is bit-identical to separate
// pn2_query_ball_point / pn2_query_ball_group_xyz calls (tests/test_configs_gpu.py).
#include "ball_query_body.h"

#include <limits.h>
#include <math.h>

namespace pn2 {

constexpr int kBqMaxScales = 4;

struct BqScale {
    float thr, radius;
    int nsample, lpq;                     // lanes per query of this radius' pass (as many queries per wave as the LDS holds)
    int *idx, *cnt;
    float *grouped;
};
struct BqScales {
    int count;
    float bin_radius;                     // the cell list is built for this radius (see pn2_query_ball_group_xyz_msg)
    BqScale s[kBqMaxScales];
};

template <int NT>
__global__ __launch_bounds__(NT, NT / 256) void ball_query_msg_kernel(int b, int n, int m, int qpb, int parts,
                                                                    int use_cells, int max_ns, int wave_stride_b, BqScales sc,
                                                                    const float *__restrict__ xyz1,
                                                                    const float *__restrict__ xyz2, int subtract)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int cloud, part;
    decode_cloud_block(blockIdx.x, parts, b, cloud, part);
    const int q0 = part * qpb, q1 = min(q0 + qpb, m);
    const float *__restrict__ data = xyz1 + (size_t)cloud * n * 3;

    // cell-list layout (ball_query_body.h), wave areas sized for the largest nsample of the level
    float4 *sorted = reinterpret_cast<float4 *>(smem);
    int *tab = reinterpret_cast<int *>(smem + sizeof(float4) * (size_t)n);
    char *wave_area = reinterpret_cast<char *>(tab + kBqTabInts);
    const size_t wave_stride = (size_t)wave_stride_b;             // largest per-wave area over the radii
    float *misc = reinterpret_cast<float *>(wave_area + (size_t)(NT / 64) * wave_stride);

    bool cells = false;
    BqGrid g;
    if (use_cells) {                                             // block-uniform
        if (threadIdx.x == 0) tab[0] = 0;
        cells = bq_build_grid<NT>(n, sc.bin_radius * 1.001f, data, sorted, tab + 1, misc, g);
    }
    // Radii still to be served by the index-ordered sweep: all of them without a cell list; with one, the radii whose block of
    // visited cells covers half of the grid or more (round 6). There the list prunes nothing -- on a uniform cube binned at 0.2
    // a ball of radius 0.4 reaches 5 of the 5 cells of every axis, 4096 candidates per query -- while the sweep stops at the
    // nsample-th hit (crowded balls: ~500 candidates for 128 hits); on the sphere-surface clouds of pointnet2_cls_msg.py:27 the
    // same radius reaches 5 of 10 cells per axis and stays on the list. uniform-cube level (0.1, 0.2, 0.4): 108 us -> see
    // profiles/r06/bq_msg_wide.txt. The cloud is restaged in index order once, behind the list passes.
    unsigned todo = (1u << sc.count) - 1u;
    if (cells) {
        for (int i = 0; i < sc.count; ++i) {
            const BqScale &s = sc.s[i];
            const float reach = s.radius * 1.001f;
            const float fx = fminf(1.0f, (2.0f * reach * g.ix + 1.0f) / (float)g.gx);
            const float fy = fminf(1.0f, (2.0f * reach * g.iy + 1.0f) / (float)g.gy);
            const float fz = fminf(1.0f, (2.0f * reach * g.iz + 1.0f) / (float)g.gz);
            if (__builtin_amdgcn_readfirstlane((int)(fx * fy * fz >= 0.5f))) continue;   // block-uniform: the grid is the workgroup's
#define PN2_MSG_LOOP(LPQ)                                                                                           \
    bq_cells_query_loop<NT, LPQ, true, false>(n, m, s.nsample, s.thr, s.radius, s.radius * 1.001f, cloud, q0, q1, g, data, \
                                              xyz2, nullptr, nullptr, s.idx, s.cnt, s.grouped, subtract, sorted, tab,  \
                                              wave_area, wave_stride)
            if (s.lpq == 8) PN2_MSG_LOOP(8);                      // block-uniform
            else if (s.lpq == 16) PN2_MSG_LOOP(16);
            else PN2_MSG_LOOP(32);
#undef PN2_MSG_LOOP
            todo &= ~(1u << i);
        }
        if (!todo) return;
    }
    // index-ordered LDS copy, padded to a multiple of 128 with points at +inf (never hit), swept per radius
    __syncthreads();                                             // the binning / list passes may still be reading
    float4 *cloud_lds = reinterpret_cast<float4 *>(smem);
    const int npad = (n + 127) & ~127;
    for (int k = threadIdx.x; k < npad; k += NT) {
        if (k < n) {
            const float *p = data + (size_t)k * 3;
            cloud_lds[k] = make_float4(p[0], p[1], p[2], 0.0f);
        } else {
            cloud_lds[k] = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
        }
    }
    __syncthreads();
    for (int i = 0; i < sc.count; ++i) {
        if (!((todo >> i) & 1u)) continue;
        const BqScale &s = sc.s[i];
        bq_block_body<true, true, false, NT, true>(n, m, s.nsample, s.thr, cloud, q0, q1, xyz1, xyz2, nullptr, nullptr, s.idx,
                                                   s.cnt, s.grouped, subtract, smem, 1u, max_ns);
    }
}

static size_t msg_sweep_bytes(int n, int max_ns, int nthreads)
{
    return sizeof(float4) * (size_t)((n + 127) & ~127) + sizeof(int) * (size_t)max_ns * (nthreads / 64) * kBqQpw;
}

// Queries per workgroup: about one workgroup per CU when the cloud is binned (the binning pass is per
// workgroup), two for the sweep; a multiple of the smallest trip (two queries per wave).
static int msg_qpb(int b, int m, int nthreads, int use_cells)
{
    const int gran = (nthreads / 64) * kBqQpw;
    const long long total = (long long)b * m;
    int qpb = (int)((total + (use_cells ? 255 : 511)) / (use_cells ? 256 : 512));
    qpb = ((qpb + gran - 1) / gran) * gran;
    if (qpb > m) qpb = ((m + gran - 1) / gran) * gran;
    return qpb;
}

// Lanes per query for every radius: the smallest of {8, 16, 32} (most queries per wave) that (1) fits the LDS
// left beside the sorted cloud and the cell table, (2) does not carry more queries per workgroup trip than the
// workgroup owns (waves * 64 / lpq <= qpb: otherwise waves idle), and (3) is at least 16 for a radius beyond
// the binning radius (long "wide" runs). Returns the per-wave stride, 0 if nothing fits.
static size_t msg_pick_lpq(int n, int nthreads, size_t cap, int qpb, BqScales &sc)
{
    const size_t fixed = sizeof(float4) * (size_t)n + sizeof(int) * (size_t)kBqTabInts + kBqMiscBytes;
    if (fixed >= cap) return 0;
    const size_t budget = (cap - fixed) / (size_t)(nthreads / 64);
    size_t stride = 0;
    for (int i = 0; i < sc.count; ++i) {
        int lpq_min = sc.s[i].radius > sc.bin_radius * 1.01f ? 16 : 8;
        while (lpq_min < 32 && (nthreads / 64) * (64 / lpq_min) > qpb) lpq_min *= 2;
        int lpq = 0;
        for (int cand : {8, 16, 32})
            if (cand >= lpq_min && bq_cells_wave_bytes(n, sc.s[i].nsample, cand) <= budget) { lpq = cand; break; }
        if (!lpq) return 0;
        sc.s[i].lpq = lpq;
        const size_t bytes = bq_cells_wave_bytes(n, sc.s[i].nsample, lpq);
        stride = bytes > stride ? bytes : stride;
    }
    return stride;
}

template <int NT>
static int launch_msg(int b, int n, int m, int use_cells, int max_ns, int qpb, size_t wave_stride, const BqScales &sc,
                      const float *xyz1, const float *xyz2, int subtract, hipStream_t st)
{
    const int parts = (m + qpb - 1) / qpb;
    const size_t cells = sizeof(float4) * (size_t)n + sizeof(int) * (size_t)kBqTabInts + (size_t)(NT / 64) * wave_stride + kBqMiscBytes;
    const size_t sweep = msg_sweep_bytes(n, max_ns, NT);
    const size_t lds = cells > sweep ? cells : sweep;
    auto kern = ball_query_msg_kernel<NT>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    return launch(kern, dim3((unsigned)parts * b), dim3(NT), lds, st, b, n, m, qpb, parts, use_cells, max_ns, (int)wave_stride, sc,
                  xyz1, xyz2, subtract);
}

}  // namespace pn2

extern "C" int pn2_query_ball_group_xyz_msg(int b, int n, int m, int nscales, const float *radii, const int *nsamples,
                                            const float *xyz1, const float *xyz2, int subtract_centroid, int *const *idx,
                                            int *const *pts_cnt, float *const *grouped_xyz, void *stream)
{
    using namespace pn2;
    if (nscales <= 0 || nscales > kBqMaxScales || !radii || !nsamples) return PN2_E_ARG;
    if (b < 0 || n <= 0 || m < 0) return PN2_E_SHAPE;
    for (int i = 0; i < nscales; ++i)
        if (!(radii[i] > 0.0f) || nsamples[i] <= 0) return PN2_E_ARG;   // tf_grouping.cpp:71,74
    BqScales sc;
    sc.count = nscales;
    int max_ns = 0;
    for (int i = 0; i < nscales; ++i) {
        sc.s[i] = {pn2_ball_threshold(radii[i]), radii[i], nsamples[i], 8, idx ? idx[i] : nullptr,
                   pts_cnt ? pts_cnt[i] : nullptr, grouped_xyz ? grouped_xyz[i] : nullptr};
        if (!sc.s[i].idx && !sc.s[i].grouped) return PN2_E_NULL;
        max_ns = nsamples[i] > max_ns ? nsamples[i] : max_ns;
        if ((long long)b * m * nsamples[i] * 3 > (1ll << 40)) return PN2_E_TOO_LARGE;
    }
    // Cell size: the SECOND smallest radius when there are three or more (else the smallest). Cells sized for
    // the smallest radius make every larger radius walk "wide" runs (whole y-ranges of a z-slab, every x): at
    // pointnet2_cls_msg.py:27's (0.1, 0.2, 0.4) binning at 0.2 keeps the 0.1 and 0.2 passes on <= 3 x 3 short
    // runs and only the 0.4 pass wide (measured: 91 us binned at 0.1 with 8 lanes per query, 56 at 0.1 with 16).
    {
        float r1 = radii[0], r2 = INFINITY;                      // smallest, second smallest
        for (int i = 1; i < nscales; ++i) {
            if (radii[i] < r1) { r2 = r1; r1 = radii[i]; }
            else if (radii[i] < r2) r2 = radii[i];
        }
        sc.bin_radius = (nscales >= 3) ? r2 : r1;
    }
    if (b == 0 || m == 0) return PN2_OK;
    if (!xyz1 || !xyz2) return PN2_E_NULL;
    if ((long long)b * n * 3 > INT_MAX) return PN2_E_TOO_LARGE;
    if (n > kBqCellsMaxPoints) return PN2_E_TOO_LARGE;          // callers launch the single-radius operator per scale
    // the binning pass is amortised over every radius of the level: worth it from mid-sized clouds on
    const int use_cells = n >= 1024 && (long long)b * m * nscales >= 8192;
    hipStream_t st = as_stream(stream);
    // geometry: two 512-thread workgroups per CU when everything fits in half the LDS, else one of 1024, else one of 512
    struct Geom { int nt; size_t cap; };
    static const Geom order[] = {{512, 80 * 1024}, {1024, 160 * 1024}, {512, 160 * 1024}};
    for (const Geom &g : order) {
        if (msg_sweep_bytes(n, max_ns, g.nt) > g.cap) continue;
        const int qpb = msg_qpb(b, m, g.nt, use_cells);
        const size_t stride = msg_pick_lpq(n, g.nt, g.cap, qpb, sc);
        if (!stride) continue;
        if (g.nt == 1024) return launch_msg<1024>(b, n, m, use_cells, max_ns, qpb, stride, sc, xyz1, xyz2, subtract_centroid, st);
        return launch_msg<512>(b, n, m, use_cells, max_ns, qpb, stride, sc, xyz1, xyz2, subtract_centroid, st);
    }
    return PN2_E_TOO_LARGE;
}
