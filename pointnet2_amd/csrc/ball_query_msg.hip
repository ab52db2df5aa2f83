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

namespace pn2 {

constexpr int kBqMaxScales = 4;

struct BqScale {
    float thr, radius;
    int nsample;
    int *idx, *cnt;
    float *grouped;
};
struct BqScales {
    int count;
    BqScale s[kBqMaxScales];
};

template <int NT, int LPQ>
__global__ __launch_bounds__(NT, NT / 256) void ball_query_msg_kernel(int b, int n, int m, int qpb, int parts,
                                                                    int use_cells, int max_ns, BqScales sc,
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
    const size_t wave_stride = bq_cells_wave_bytes(n, max_ns, LPQ);
    float *misc = reinterpret_cast<float *>(wave_area + (size_t)(NT / 64) * wave_stride);

    bool cells = false;
    BqGrid g;
    if (use_cells) {                                             // block-uniform
        float rmin = sc.s[0].radius;
        for (int i = 1; i < sc.count; ++i) rmin = fminf(rmin, sc.s[i].radius);
        if (threadIdx.x == 0) tab[0] = 0;
        cells = bq_build_grid<NT>(n, rmin * 1.001f, data, sorted, tab + 1, misc, g);
    }
    if (cells) {
        for (int i = 0; i < sc.count; ++i) {
            const BqScale &s = sc.s[i];
            bq_cells_query_loop<NT, LPQ, true, false>(n, m, s.nsample, s.thr, s.radius, s.radius * 1.001f, cloud, q0, q1, g,
                                                     data, xyz2, nullptr, nullptr, s.idx, s.cnt, s.grouped, subtract, sorted,
                                                     tab, wave_area, wave_stride);
        }
        return;
    }
    // index-ordered LDS copy, padded to a multiple of 128 with points at +inf (never hit), swept per radius
    __syncthreads();                                             // the binning pass may still be reading its scratch
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
        const BqScale &s = sc.s[i];
        bq_block_body<true, true, false, NT, true>(n, m, s.nsample, s.thr, cloud, q0, q1, xyz1, xyz2, nullptr, nullptr, s.idx,
                                                   s.cnt, s.grouped, subtract, smem, 1u, max_ns);
    }
}

static size_t msg_lds_bytes(int n, int max_ns, int lpq, int nthreads)
{
    const size_t cells = sizeof(float4) * (size_t)n + sizeof(int) * (size_t)kBqTabInts +
                         (size_t)(nthreads / 64) * bq_cells_wave_bytes(n, max_ns, lpq) + kBqMiscBytes;
    const size_t sweep = sizeof(float4) * (size_t)((n + 127) & ~127) +
                         sizeof(int) * (size_t)max_ns * (nthreads / 64) * kBqQpw;
    return cells > sweep ? cells : sweep;
}

template <int NT, int LPQ>
static int launch_msg(int b, int n, int m, int use_cells, int max_ns, const BqScales &sc, const float *xyz1,
                      const float *xyz2, int subtract, hipStream_t st)
{
    // queries per workgroup trip: the cell-list loop carries 64 / LPQ queries per wave, the sweep two
    const int kGran = (NT / 64) * (use_cells ? 64 / LPQ : kBqQpw);
    const long long total = (long long)b * m;
    // about one workgroup per CU when the cloud is binned (the binning pass is per workgroup); the sweep
    // stages cheaply and wants two
    int qpb = (int)((total + (use_cells ? 255 : 511)) / (use_cells ? 256 : 512));
    qpb = ((qpb + kGran - 1) / kGran) * kGran;
    if (qpb > m) qpb = ((m + kGran - 1) / kGran) * kGran;
    const int parts = (m + qpb - 1) / qpb;
    const size_t lds = msg_lds_bytes(n, max_ns, LPQ, NT);
    auto kern = ball_query_msg_kernel<NT, LPQ>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    return launch(kern, dim3((unsigned)parts * b), dim3(NT), lds, st, b, n, m, qpb, parts, use_cells, max_ns, sc, xyz1,
                  xyz2, subtract);
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
        sc.s[i] = {pn2_ball_threshold(radii[i]), radii[i], nsamples[i], idx ? idx[i] : nullptr,
                   pts_cnt ? pts_cnt[i] : nullptr, grouped_xyz ? grouped_xyz[i] : nullptr};
        if (!sc.s[i].idx && !sc.s[i].grouped) return PN2_E_NULL;
        max_ns = nsamples[i] > max_ns ? nsamples[i] : max_ns;
        if ((long long)b * m * nsamples[i] * 3 > (1ll << 40)) return PN2_E_TOO_LARGE;
    }
    if (b == 0 || m == 0) return PN2_OK;
    if (!xyz1 || !xyz2) return PN2_E_NULL;
    if ((long long)b * n * 3 > INT_MAX) return PN2_E_TOO_LARGE;
    if (n > kBqCellsMaxPoints) return PN2_E_TOO_LARGE;          // callers launch the single-radius operator per scale
    // the binning pass is amortised over every radius of the level: worth it from mid-sized clouds on
    const int use_cells = n >= 1024 && (long long)b * m * nscales >= 8192;
    hipStream_t st = as_stream(stream);
    // geometry: as many queries per wave as the LDS holds bitmaps and rows for (see bq_cells_pick)
    struct Geom { int nt, lpq; size_t cap; };
    static const Geom order[] = {{512, 8, 80 * 1024},   {512, 16, 80 * 1024},  {1024, 8, 160 * 1024}, {1024, 16, 160 * 1024},
                                 {512, 8, 160 * 1024},  {512, 16, 160 * 1024}, {512, 32, 160 * 1024}};
    for (const Geom &g : order) {
        if (msg_lds_bytes(n, max_ns, g.lpq, g.nt) > g.cap) continue;
#define PN2_MSG_CASE(NT, LPQ) \
        if (g.nt == NT && g.lpq == LPQ) return launch_msg<NT, LPQ>(b, n, m, use_cells, max_ns, sc, xyz1, xyz2, subtract_centroid, st)
        PN2_MSG_CASE(1024, 8);
        PN2_MSG_CASE(1024, 16);
        PN2_MSG_CASE(512, 8);
        PN2_MSG_CASE(512, 16);
        PN2_MSG_CASE(512, 32);
#undef PN2_MSG_CASE
    }
    return PN2_E_TOO_LARGE;
}
