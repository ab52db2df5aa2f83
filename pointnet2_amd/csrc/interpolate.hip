// interpolate.hip -- three_nn / three_interpolate / three_interpolate_grad for gfx950.
//
// Replaces the reference's CPU-only loops threenn_cpu, threeinterpolate_cpu,
// threeinterpolate_grad_cpu (tf_ops/3d_interpolation/tf_interpolate.cpp:60-103,
// :107-127, :131-153). In the reference these ops force a device->host->device
// round trip inside every feature-propagation layer.
//   * three_nn is index- and bit-exact: same fp32 squared distance
//     ((dx*dx)+(dy*dy))+(dz*dz), same strict-< cascade scanning the known points
//     in ascending order (order key (d,k)), +inf / index 0 for missing
//     neighbours (the reference's (float)1e40).
//   * three_interpolate is bit-exact too: (p1*w1 + p2*w2) + p3*w3, no FMA.
//   * the gradient accumulates with fp32 atomics (order not fixed).
//
// Design (DESIGN.md "three_nn"). FOUR lanes (a DPP quad) per unknown point: the known points are
// staged through LDS in float4 tiles (index in .w), lane q of the quad takes every fourth candidate,
// and at the end the quad merges its four triples with two quad_perm exchanges. The running top-3 is a
// v_min_f64 / v_max_f64 network on (d : k) keys -- the order key the reference's single ascending
// strict-< scan implies -- so visiting order is irrelevant and there are no compares or selects.
// The first version used one lane per point: at the largest FP layer (8 x 8192 unknown points) that
// is one wave per SIMD, a pure latency chain (115 us); the quad split quadruples the waves in flight.
#include "ball_query_body.h"

#include <limits.h>
#include <math.h>

namespace pn2 {

constexpr int kNnThreads = 256;             // 64 unknown points x 4 lanes
constexpr int kNnPoints = kNnThreads / 4;
constexpr int kNnTile = 2048;               // known points per LDS tile (32 KiB), a multiple of 16

// Top-3 as a min/max network on 64-bit keys (d bits : k). Read as fp64 the pattern is positive and
// finite for every fp32 d >= 0 including +inf and NaN payloads (exponent field < 0x7FF), and ordered
// exactly like the reference's scan order key (d, k): smaller d first, then smaller k. Inserting x
// into the ascending triple is five v_min_f64/v_max_f64, no compares, no selects:
//     t1 = min(b1,x); x = max(b1,x); t2 = min(b2,x); x = max(b2,x); t3 = min(b3,x)
// The empty slot is (+inf : 0): a candidate with d = +inf has a key >= it and never enters, exactly as
// `inf < 1e40` is false in the reference (tf_interpolate.cpp:74); NaN keys are larger still.
__device__ __forceinline__ double nn_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double nn_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void nn_insert(double x, double &b1, double &b2, double &b3)
{
    const double t1 = nn_min(b1, x);
    x = nn_max(b1, x);
    const double t2 = nn_min(b2, x);
    x = nn_max(b2, x);
    b3 = nn_min(b3, x);
    b1 = t1;
    b2 = t2;
}

template <int CTRL>
__device__ __forceinline__ double quad_xchg_d(double v)
{
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// merge the partner lane's triple (quad_perm CTRL) into this lane's
template <int CTRL>
__device__ __forceinline__ void nn_merge(double &b1, double &b2, double &b3)
{
    const double o1 = quad_xchg_d<CTRL>(b1), o2 = quad_xchg_d<CTRL>(b2), o3 = quad_xchg_d<CTRL>(b3);
    nn_insert(o1, b1, b2, b3);
    nn_insert(o2, b1, b2, b3);
    nn_insert(o3, b1, b2, b3);
}

__global__ __launch_bounds__(kNnThreads) void three_nn_kernel(int n, int m, const float *__restrict__ xyz1,
                                                              const float *__restrict__ xyz2,
                                                              float *__restrict__ dist, int *__restrict__ idx)
{
    __shared__ float4 tile[kNnTile];
    const int bi = blockIdx.y;
    const int sub = threadIdx.x & 3;                       // lane of the quad
    const int j = blockIdx.x * kNnPoints + (threadIdx.x >> 2);
    const bool live = j < n;
    const float *u = xyz1 + ((size_t)bi * n + (live ? j : 0)) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float *__restrict__ known = xyz2 + (size_t)bi * m * 3;

    const double empty = __hiloint2double(0x7F800000, 0);  // (+inf : 0) = the reference's (float)1e40, index 0 (:67)
    double b1 = empty, b2 = empty, b3 = empty;
    for (int base = 0; base < m; base += kNnTile) {
        const int cnt = min(kNnTile, m - base);
        const int cnt16 = (cnt + 15) & ~15;                // whole trips of 16 candidates per quad
        __syncthreads();
        for (int k = threadIdx.x; k < cnt16; k += kNnThreads) {
            if (k < cnt) {
                const float *p = known + (size_t)(base + k) * 3;
                tile[k] = make_float4(p[0], p[1], p[2], __int_as_float(base + k));   // .w = the index, bit for bit
            } else {
                tile[k] = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(0));   // pad: d = +inf, never enters
            }
        }
        __syncthreads();
        // 16 candidates per quad and trip, lane q takes 4c+q (c = 0..3): the four lanes of a quad read
        // four CONSECUTIVE float4 per instruction -- different LDS banks. (Giving each lane a contiguous
        // quarter of the tile put the quad on one bank: a 4-way conflict on every read made the kernel
        // LDS bound, 241 us instead of 64 at 32 x 8192 x 1024.) Order of visit is irrelevant: keyed network.
        for (int k = sub; k < cnt16; k += 16) {
            const float4 p0 = tile[k], p1 = tile[k + 4], p2 = tile[k + 8], p3 = tile[k + 12];
            // (x2-x1)..., x2 the known point (tf_interpolate.cpp:69-73)
            const float d0 = sqdist_key(p0.x, p0.y, p0.z, ux, uy, uz);
            const float d1 = sqdist_key(p1.x, p1.y, p1.z, ux, uy, uz);
            const float d2 = sqdist_key(p2.x, p2.y, p2.z, ux, uy, uz);
            const float d3 = sqdist_key(p3.x, p3.y, p3.z, ux, uy, uz);
            nn_insert(__hiloint2double(__float_as_int(d0), __float_as_int(p0.w)), b1, b2, b3);
            nn_insert(__hiloint2double(__float_as_int(d1), __float_as_int(p1.w)), b1, b2, b3);
            nn_insert(__hiloint2double(__float_as_int(d2), __float_as_int(p2.w)), b1, b2, b3);
            nn_insert(__hiloint2double(__float_as_int(d3), __float_as_int(p3.w)), b1, b2, b3);
        }
    }
    nn_merge<0xB1>(b1, b2, b3);                            // quad_perm:[1,0,3,2]
    nn_merge<0x4E>(b1, b2, b3);                            // quad_perm:[2,3,0,1]
    if (live && sub == 0) {
        float *od = dist + ((size_t)bi * n + j) * 3;
        int *oi = idx + ((size_t)bi * n + j) * 3;
        od[0] = __int_as_float(__double2hiint(b1)); od[1] = __int_as_float(__double2hiint(b2));
        od[2] = __int_as_float(__double2hiint(b3));
        oi[0] = __double2loint(b1); oi[1] = __double2loint(b2); oi[2] = __double2loint(b3);
    }
}


// ---- three_nn with a cell list (round 6) ---------------------------------------------------------
// The sweep above tests every unknown point against every known point: 67 M pair tests at sem_seg's last feature-propagation
// level (8 x 8192 unknown, 1024 known), 28 us. But the three nearest known points of an unknown point lie within about one
// spacing of the known set. Here a workgroup bins its cloud's known points ONCE into a grid (the ball query's counting sort,
// ball_query_body.h: cell edge = kNnCellFactor x the mean spacing cbrt(box volume / m)), every unknown point -- four lanes,
// as above -- visits the 3 x 3 runs of three x-adjacent cells around its own cell (35-50 candidates instead of 1024), and
// keeps the same keyed top-3 network, so the visiting order is irrelevant.
// EXACT: a known point OUTSIDE the visited block differs from the unknown point's cell by two or more cells along some axis,
// so it is farther away than the distance `margin` from the unknown point to the nearest face of the block that has cells
// behind it. If the third-best squared distance found is below margin'^2 (margin' = margin less 0.1 % and a coordinate-scale
// epsilon: both evaluations' rounding errors are six orders of magnitude smaller), no outside point can enter the triple --
// the result is the sweep's, bit for bit. Unknown points that fail the test (fewer than three candidates, a sparse
// neighbourhood, a position outside the known points' box) are collected and answered by a SWEEP of all known points, one
// wave per point; a workgroup whose cloud cannot be binned (fewer than 64 cells, a crowded cell) sweeps all of its points.
constexpr float kNnCellFactor = 1.3f;       // simulation (profiles/r06/three_nn_cells.txt): 1.2 -> 2 % of Poisson-sampled points fall back, 35 candidates; 1.6 -> 0 %, 77
constexpr int kNnCellsThreads = 512;

// the partner's key through a DPP move whose unwritten lanes receive the EMPTY key (+inf : 0) -- never 0.0, which would win
// every minimum -- so that a lane without a partner merges nothing
template <int CTRL, int ROWS>
__device__ __forceinline__ double nn_dpp_or_empty(double v)
{
    const int hi = __builtin_amdgcn_update_dpp(0x7F800000, __double2hiint(v), CTRL, ROWS, 0xf, false);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWS>
__device__ __forceinline__ void nn_merge_dpp(double &b1, double &b2, double &b3)
{
    const double o1 = nn_dpp_or_empty<CTRL, ROWS>(b1), o2 = nn_dpp_or_empty<CTRL, ROWS>(b2), o3 = nn_dpp_or_empty<CTRL, ROWS>(b3);
    nn_insert(o1, b1, b2, b3);
    nn_insert(o2, b1, b2, b3);
    nn_insert(o3, b1, b2, b3);
}

template <int NT>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 4) void three_nn_cells_kernel(int n, int m, int rows_per_part, int parts, int b, float factor,
                                                            const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                            float *__restrict__ dist, int *__restrict__ idx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *sorted = reinterpret_cast<float4 *>(smem);                                  // [m] known points in cell order, .w = index
    int *tab = reinterpret_cast<int *>(smem + sizeof(float4) * (size_t)((m + 3) & ~3)); // tab[c] = start of cell c, tab[c + 1] = its end
    float *misc = reinterpret_cast<float *>(tab + kBqTabInts);
    int *fail = reinterpret_cast<int *>(reinterpret_cast<char *>(misc) + kBqMiscBytes); // [0] = count, [1 ..] = rows to sweep
    int cloud, part;
    decode_cloud_block(blockIdx.x, parts, b, cloud, part);
    const int rb = part * rows_per_part, re = min(rb + rows_per_part, n);
    const float *__restrict__ known = xyz2 + (size_t)cloud * m * 3;
    const float *__restrict__ unk = xyz1 + (size_t)cloud * n * 3;
    float *__restrict__ od = dist + (size_t)cloud * n * 3;
    int *__restrict__ oi = idx + (size_t)cloud * n * 3;
    const int t = threadIdx.x, lane = t & 63, sub = t & 3;
    const double empty = __hiloint2double(0x7F800000, 0);  // (+inf : 0) = the reference's (float)1e40, index 0 (:67)

    if (t == 0) { tab[0] = 0; fail[0] = 0; }
    BqGrid g;
    const bool binned = bq_build_grid<NT>(m, -factor, known, sorted, tab + 1, misc, g);   // block-uniform; ends with a barrier when true
    if (!binned) {
        __syncthreads();
        for (int k = t; k < m; k += NT) {
            const float *p = known + (size_t)k * 3;
            sorted[k] = make_float4(p[0], p[1], p[2], __int_as_float(k));
        }
        for (int r = rb + t; r < re; r += NT) fail[1 + (r - rb)] = r;
        if (t == 0) fail[0] = re - rb;
    } else {
        const float ex = 1.0f / g.ix, ey = 1.0f / g.iy, ez = 1.0f / g.iz;               // cell edges
        const float scale = fmaxf(fmaxf(fabsf(g.ox) + ex * g.gx, fabsf(g.oy) + ey * g.gy), fabsf(g.oz) + ez * g.gz);
        for (int r0 = rb; r0 < re; r0 += NT / 4) {
            const int j = r0 + (t >> 2);
            const bool live = j < re;
            const float *u = unk + (size_t)(live ? j : rb) * 3;
            const float ux = u[0], uy = u[1], uz = u[2];               // (requested before the binning instead: no gain, and spills at 512 threads)
            const int cx = bq_cell(ux, g.ox, g.ix, g.gx), cy = bq_cell(uy, g.oy, g.iy, g.gy), cz = bq_cell(uz, g.oz, g.iz, g.gz);
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.gx - 1);
            int rs[9], rl[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int y = cy + (q % 3) - 1, z = cz + (q / 3) - 1;
                const bool ok = y >= 0 && y < g.gy && z >= 0 && z < g.gz;
                const int row = ok ? (z * g.gy + y) * g.gx : 0;
                const int s0 = tab[row + x0], e0 = tab[row + x1 + 1];
                rs[q] = s0; rl[q] = ok ? e0 : s0;
            }
            double b1 = empty, b2 = empty, b3 = empty;
            // A run holds 4-9 candidates, i.e. one or two per lane of the quad: the first TWO candidates of all nine runs are
            // requested at once (eighteen LDS reads in flight, one latency) and inserted without a branch -- a lane whose run
            // is shorter inserts the empty key, which the network ignores; only runs longer than eight take the loop behind.
            // (Run by run with a loop each, every run paid its own LDS latency and loop overhead: 4 us per 128 unknown points
            // against 2 now, scripts/three_nn_probe.py.)
            float4 pa[9], pb[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int ka = rs[q] + sub, kb = ka + 4;
                pa[q] = sorted[ka < rl[q] ? ka : 0];
                pb[q] = sorted[kb < rl[q] ? kb : 0];
            }
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int ka = rs[q] + sub, kb = ka + 4;
                const float da = sqdist_key(pa[q].x, pa[q].y, pa[q].z, ux, uy, uz);      // (x2-x1)..., x2 the known point (tf_interpolate.cpp:69-73)
                const float db = sqdist_key(pb[q].x, pb[q].y, pb[q].z, ux, uy, uz);
                const double ea = __hiloint2double(__float_as_int(da), __float_as_int(pa[q].w));
                const double eb = __hiloint2double(__float_as_int(db), __float_as_int(pb[q].w));
                nn_insert(ka < rl[q] ? ea : empty, b1, b2, b3);
                nn_insert(kb < rl[q] ? eb : empty, b1, b2, b3);
            }
#pragma unroll
            for (int q = 0; q < 9; ++q)
                for (int k = rs[q] + sub + 8; k < rl[q]; k += 4) {
                    const float4 p = sorted[k];
                    const float d = sqdist_key(p.x, p.y, p.z, ux, uy, uz);
                    nn_insert(__hiloint2double(__float_as_int(d), __float_as_int(p.w)), b1, b2, b3);
                }
            nn_merge<0xB1>(b1, b2, b3);
            nn_merge<0x4E>(b1, b2, b3);
            // distance to the nearest face of the visited block that has cells behind it
            float margin = INFINITY;
            if (cx - 1 > 0) margin = fminf(margin, ux - (g.ox + (float)(cx - 1) * ex));
            if (cx + 1 < g.gx - 1) margin = fminf(margin, (g.ox + (float)(cx + 2) * ex) - ux);
            if (cy - 1 > 0) margin = fminf(margin, uy - (g.oy + (float)(cy - 1) * ey));
            if (cy + 1 < g.gy - 1) margin = fminf(margin, (g.oy + (float)(cy + 2) * ey) - uy);
            if (cz - 1 > 0) margin = fminf(margin, uz - (g.oz + (float)(cz - 1) * ez));
            if (cz + 1 < g.gz - 1) margin = fminf(margin, (g.oz + (float)(cz + 2) * ez) - uz);
            margin = margin * 0.999f - 1e-5f * scale;
            const float d3 = __int_as_float(__double2hiint(b3));
            const bool sure = margin > 0.0f && d3 < margin * margin;                      // NaN anywhere -> not sure -> swept
            if (live && sub == 0) {
                if (sure) {
                    od[(size_t)j * 3 + 0] = __int_as_float(__double2hiint(b1)); od[(size_t)j * 3 + 1] = __int_as_float(__double2hiint(b2));
                    od[(size_t)j * 3 + 2] = d3;
                    oi[(size_t)j * 3 + 0] = __double2loint(b1); oi[(size_t)j * 3 + 1] = __double2loint(b2); oi[(size_t)j * 3 + 2] = __double2loint(b3);
                } else {
                    fail[1 + atomicAdd(&fail[0], 1)] = j;
                }
            }
        }
    }
    __syncthreads();
    // the rows the cell list could not answer: a sweep of every known point, one wave per row (lane l takes l, l + 64, ...)
    const int nf = fail[0];
    for (int f = t >> 6; f < nf; f += NT / 64) {
        const int j = fail[1 + f];
        const float ux = unk[(size_t)j * 3 + 0], uy = unk[(size_t)j * 3 + 1], uz = unk[(size_t)j * 3 + 2];
        double b1 = empty, b2 = empty, b3 = empty;
        for (int k = lane; k < m; k += 64) {
            const float4 p = sorted[k];
            const float d = sqdist_key(p.x, p.y, p.z, ux, uy, uz);
            nn_insert(__hiloint2double(__float_as_int(d), __float_as_int(p.w)), b1, b2, b3);
        }
        // the wave's 64 triples -> lane 63: quads, half rows, rows (every lane of a merged group holds the group's triple, so
        // a mirror pairs disjoint groups), then the two row broadcasts (lanes they do not write merge the empty key)
        nn_merge<0xB1>(b1, b2, b3);
        nn_merge<0x4E>(b1, b2, b3);
        nn_merge_dpp<0x141, 0xf>(b1, b2, b3);              // row_half_mirror
        nn_merge_dpp<0x140, 0xf>(b1, b2, b3);              // row_mirror
        nn_merge_dpp<0x142, 0xa>(b1, b2, b3);              // row_bcast:15 -> rows 1, 3
        nn_merge_dpp<0x143, 0xc>(b1, b2, b3);              // row_bcast:31 -> rows 2, 3
        if (lane == 63) {
            od[(size_t)j * 3 + 0] = __int_as_float(__double2hiint(b1)); od[(size_t)j * 3 + 1] = __int_as_float(__double2hiint(b2));
            od[(size_t)j * 3 + 2] = __int_as_float(__double2hiint(b3));
            oi[(size_t)j * 3 + 0] = __double2loint(b1); oi[(size_t)j * 3 + 1] = __double2loint(b2); oi[(size_t)j * 3 + 2] = __double2loint(b3);
        }
    }
}

static size_t three_nn_cells_lds(int m, int rows_per_part)
{
    return sizeof(float4) * (size_t)((m + 3) & ~3) + sizeof(int) * (size_t)kBqTabInts + kBqMiscBytes + sizeof(int) * (size_t)(rows_per_part + 4);
}

constexpr int kThreads = 256;

static inline unsigned grid_for(long long work)
{
    long long g = (work + kThreads - 1) / kThreads;
    const long long cap = 256ll * 32;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ float interp3(float p1, float p2, float p3, float w1, float w2, float w3)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(p1, w1), __fmul_rn(p2, w2)), __fmul_rn(p3, w3));   // :122
}

// c % 4 == 0: one float4 of the output row per lane
__global__ __launch_bounds__(kThreads) void three_interpolate_v4_kernel(long long chunks, int m, int n, int c4,
                                                                        const float4 *__restrict__ points,
                                                                        const int *__restrict__ idx,
                                                                        const float *__restrict__ weight,
                                                                        float4 *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < chunks; e += (long long)gridDim.x * kThreads) {
        const long long r = e / c4;            // r = i*n + j
        const int l = (int)(e - r * c4);
        const long long i = r / n;
        const int *q = idx + r * 3;
        const float *wq = weight + r * 3;
        const float w1 = wq[0], w2 = wq[1], w3 = wq[2];
        const float4 *base = points + i * m * c4;
        const float4 a = base[(long long)q[0] * c4 + l];
        const float4 b = base[(long long)q[1] * c4 + l];
        const float4 c = base[(long long)q[2] * c4 + l];
        float4 o;
        o.x = interp3(a.x, b.x, c.x, w1, w2, w3);
        o.y = interp3(a.y, b.y, c.y, w1, w2, w3);
        o.z = interp3(a.z, b.z, c.z, w1, w2, w3);
        o.w = interp3(a.w, b.w, c.w, w1, w2, w3);
        out[e] = o;
    }
}

// Second generation (see group.hip): grid = (parts per cloud) x (clouds) with the XCD-aware decode, the
// (row, chunk) pair derived once per thread and advanced by constants (no integer division per element),
// U rows' worth of idx / weight / three gathered rows in flight per lane.
template <int U, bool NT>
__global__ __launch_bounds__(kThreads) void three_interpolate_rows_v4_kernel(int n, int m, int c4, int dr, int dl,
                                                                             int rows_per_part, int parts, int b,
                                                                             const float4 *__restrict__ points,
                                                                             const int *__restrict__ idx,
                                                                             const float *__restrict__ weight,
                                                                             float4 *__restrict__ out)
{
    int cloud, part;
    decode_cloud_block(blockIdx.x, parts, b, cloud, part);
    const int rb = part * rows_per_part, re = min(rb + rows_per_part, n);
    const int *__restrict__ idc = idx + (size_t)cloud * n * 3;
    const float *__restrict__ wc = weight + (size_t)cloud * n * 3;
    const float4 *__restrict__ src = points + (size_t)cloud * m * c4;
    float4 *__restrict__ dst = out + (size_t)cloud * n * c4;
    int r = rb + (int)threadIdx.x / c4, l = (int)threadIdx.x % c4;
    while (r < re) {
        int rr[U], ll[U], q0[U], q1[U], q2[U];
        float w1[U], w2[U], w3[U];
        float4 a[U], bb[U], cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rr[u] = r; ll[u] = l;
            l += dl; r += dr;
            if (l >= c4) { l -= c4; ++r; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = rr[u] < re ? rr[u] : rb;             // clamp: the loads stay in range, the store is skipped
            const int *q = idc + (unsigned)row * 3u;
            const float *w = wc + (unsigned)row * 3u;
            q0[u] = q[0]; q1[u] = q[1]; q2[u] = q[2];
            w1[u] = w[0]; w2[u] = w[1]; w3[u] = w[2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = src[(unsigned)q0[u] * (unsigned)c4 + (unsigned)ll[u]];
            bb[u] = src[(unsigned)q1[u] * (unsigned)c4 + (unsigned)ll[u]];
            cc[u] = src[(unsigned)q2[u] * (unsigned)c4 + (unsigned)ll[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rr[u] < re) {
                float4 o;
                o.x = interp3(a[u].x, bb[u].x, cc[u].x, w1[u], w2[u], w3[u]);
                o.y = interp3(a[u].y, bb[u].y, cc[u].y, w1[u], w2[u], w3[u]);
                o.z = interp3(a[u].z, bb[u].z, cc[u].z, w1[u], w2[u], w3[u]);
                o.w = interp3(a[u].w, bb[u].w, cc[u].w, w1[u], w2[u], w3[u]);
                float4 *po = dst + ((unsigned)rr[u] * (unsigned)c4 + (unsigned)ll[u]);
                if (NT) {
                    typedef float pn2_v4f __attribute__((ext_vector_type(4)));
                    pn2_v4f w = {o.x, o.y, o.z, o.w};
                    __builtin_nontemporal_store(w, reinterpret_cast<pn2_v4f *>(po));
                } else {
                    *po = o;
                }
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void three_interpolate_s_kernel(long long elems, int m, int n, int c,
                                                                       const float *__restrict__ points,
                                                                       const int *__restrict__ idx,
                                                                       const float *__restrict__ weight,
                                                                       float *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < elems; e += (long long)gridDim.x * kThreads) {
        const long long r = e / c;
        const int l = (int)(e - r * c);
        const long long i = r / n;
        const int *q = idx + r * 3;
        const float *wq = weight + r * 3;
        const float *base = points + i * m * c;
        out[e] = interp3(base[(long long)q[0] * c + l], base[(long long)q[1] * c + l], base[(long long)q[2] * c + l],
                         wq[0], wq[1], wq[2]);
    }
}

__global__ __launch_bounds__(kThreads) void three_interpolate_grad_kernel(long long elems, int m, int n, int c,
                                                                          const float *__restrict__ grad_out,
                                                                          const int *__restrict__ idx,
                                                                          const float *__restrict__ weight,
                                                                          float *__restrict__ grad_points)
{
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < elems; e += (long long)gridDim.x * kThreads) {
        const long long r = e / c;
        const int l = (int)(e - r * c);
        const long long i = r / n;
        const int *q = idx + r * 3;
        const float *wq = weight + r * 3;
        const float g = grad_out[e];
        float *base = grad_points + i * m * c;
        atomicAdd(base + (long long)q[0] * c + l, __fmul_rn(g, wq[0]));   // :146-148
        atomicAdd(base + (long long)q[1] * c + l, __fmul_rn(g, wq[1]));
        atomicAdd(base + (long long)q[2] * c + l, __fmul_rn(g, wq[2]));
    }
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- the input rows of a feature-propagation level's layer stack in ONE launch (training path) ------------------------------
// pointnet_fp_module, utils/pointnet_util.py:211-219: dist = max(dist, 1e-10); norm = sum(1/dist); weight = (1/dist)/norm;
// interpolated = three_interpolate(points2, idx, weight); new_points1 = concat([interpolated, points1]). As operators that is
// four elementwise launches for the weights, the interpolation, the concatenation and (odd widths) a zero pad; the small
// levels of the segmentation networks are made of launch latencies. One thread per (row, four output columns); the
// arithmetic is the operators' own, operation for operation (weights: IEEE divisions, (r1 + r2) + r3; rows:
// (p1 w1 + p2 w2) + p3 w3 without FMA); against torch's elementwise kernels the weights may differ in the last place.
__global__ __launch_bounds__(kThreads) void fp_interp_concat_kernel(long long chunks, int n, int m, int c2, int c1, int pitch,
                                                                    const float *__restrict__ points2,
                                                                    const float *__restrict__ points1,
                                                                    const int *__restrict__ idx, const float *__restrict__ dist,
                                                                    float *__restrict__ out, float *__restrict__ weight)
{
    const int pc4 = pitch / 4;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < chunks; e += (long long)gridDim.x * kThreads) {
        const long long r = e / pc4;                               // r = cloud * n + point
        const int col = (int)(e - r * pc4) * 4;
        const long long cloud = r / n;
        float4 o = {0.0f, 0.0f, 0.0f, 0.0f};
        if (col < c2) {
            const int *q = idx + r * 3;
            const float *dp = dist + r * 3;
            const float r1 = __fdiv_rn(1.0f, fmaxf(dp[0], 1e-10f)), r2 = __fdiv_rn(1.0f, fmaxf(dp[1], 1e-10f)),
                        r3 = __fdiv_rn(1.0f, fmaxf(dp[2], 1e-10f));                   // :212
            const float norm = __fadd_rn(__fadd_rn(r1, r2), r3);                     // :213
            const float w1 = __fdiv_rn(r1, norm), w2 = __fdiv_rn(r2, norm), w3 = __fdiv_rn(r3, norm);   // :215
            if (col == 0 && weight) { weight[r * 3 + 0] = w1; weight[r * 3 + 1] = w2; weight[r * 3 + 2] = w3; }
            const float *base = points2 + cloud * m * c2;
            const float *pa = base + (long long)q[0] * c2, *pb = base + (long long)q[1] * c2, *pc = base + (long long)q[2] * c2;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = col + i;
                v[i] = ch < c2 ? interp3(pa[ch], pb[ch], pc[ch], w1, w2, w3)          // :216
                               : (ch - c2 < c1 ? points1[r * c1 + (ch - c2)] : 0.0f);   // :219 (a chunk that straddles the seam)
            }
            o = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = col + i - c2;
                v[i] = k < c1 ? points1[r * c1 + k] : 0.0f;        // beyond c2 + c1: the zero pad the layer stack's 16-byte rows need
            }
            o = make_float4(v[0], v[1], v[2], v[3]);
        }
        *reinterpret_cast<float4 *>(out + r * pitch + col) = o;
    }
}

// backward of the concatenation: the layer stack's input gradient (rows, pitch) -> the interpolated part, contiguous (rows, c2)
// (three_interpolate's gradient scatters it onto points2), and the skip features' part (rows, c1) = grad of points1
__global__ __launch_bounds__(kThreads) void fp_split_grad_kernel(long long elems, int c2, int c1, int pitch,
                                                                 const float *__restrict__ gx, float *__restrict__ gi,
                                                                 float *__restrict__ g1)
{
    const int c = c2 + c1;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < elems; e += (long long)gridDim.x * kThreads) {
        const long long r = e / c;
        const int ch = (int)(e - r * c);
        const float v = gx[r * pitch + ch];
        if (ch < c2) gi[r * c2 + ch] = v;
        else if (g1) g1[r * c1 + (ch - c2)] = v;
    }
}

}  // namespace pn2

extern "C" int pn2_fp_interp_concat(int b, int n, int m, int c2, int c1, int pitch, const float *points2, const float *points1,
                                    const int *idx, const float *dist, float *out, float *weight, void *stream)
{
    using namespace pn2;
    if (b < 0 || n < 0 || m <= 0 || c2 <= 0 || c1 < 0 || pitch < c2 + c1 || pitch % 4) return PN2_E_SHAPE;
    const long long rows = (long long)b * n;
    if (rows == 0) return PN2_OK;
    if (!points2 || !idx || !dist || !out || (c1 > 0 && !points1)) return PN2_E_NULL;
    if (!aligned16(out)) return PN2_E_ARG;
    const long long chunks = rows * (pitch / 4);
    return launch(fp_interp_concat_kernel, dim3(grid_for(chunks)), dim3(kThreads), 0, as_stream(stream), chunks, n, m, c2, c1, pitch,
                  points2, points1, idx, dist, out, weight);
}

extern "C" int pn2_fp_interp_concat_grad(int b, int n, int m, int c2, int c1, int pitch, const float *grad_x, const int *idx,
                                         const float *weight, float *grad_points2, float *grad_points1, float *scratch,
                                         void *ws_seg, int deterministic, void *stream)
{
    using namespace pn2;
    if (b < 0 || n < 0 || m <= 0 || c2 <= 0 || c1 < 0 || pitch < c2 + c1) return PN2_E_SHAPE;
    const long long rows = (long long)b * n;
    if (rows == 0) {
        // no unknown point: the gradient of points2 is zero, and this entry point -- like pn2_three_interpolate_grad_seg --
        // owns the zero fill (callers allocate it uninitialised); nothing flows to points1 (b * n == 0 rows)
        if (b > 0 && grad_points2) {
            if (int rc = clear_async(grad_points2, sizeof(float) * (size_t)b * m * c2, as_stream(stream))) return rc;
        }
        return PN2_OK;
    }
    if (!grad_x || !idx || !weight || !scratch || !ws_seg || !grad_points2) return PN2_E_NULL;
    const long long elems = rows * (c2 + (grad_points1 ? c1 : 0));
    if (int rc = launch(fp_split_grad_kernel, dim3(grid_for(elems)), dim3(kThreads), 0, as_stream(stream), elems, c2,
                        grad_points1 ? c1 : 0, pitch, grad_x, scratch, grad_points1)) return rc;
    return pn2_three_interpolate_grad_seg(b, n, c2, m, scratch, idx, weight, grad_points2, ws_seg, deterministic, stream);
}

// variant: 0 = the library's choice, 1 = the sweep (three_nn_kernel), 2 = the cell list (PN2_E_ARG where it does not exist:
// fewer than 64 or more than 8192 known points)
static int three_nn_entry(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx, int variant, void *stream)
{
    using namespace pn2;
    if (b < 0 || n < 0 || m < 0) return PN2_E_SHAPE;
    // tuning hook of scripts/three_nn_probe.py: variant 2 | cell factor code << 4 | rows per workgroup / 128 << 8 | threads << 16 (0 = the defaults)
    const int lab_rows = variant > 2 ? ((variant >> 8) & 0xff) * 128 : 0, lab_nt = variant > 2 ? ((variant >> 16) & 0xfff) : 0, lab_f = variant > 2 ? ((variant >> 4) & 0xf) : 0;
    if (variant > 2 && (variant & 0xf) == 2 && lab_rows <= 1024 && (lab_nt == 0 || lab_nt == 256 || lab_nt == 512 || lab_nt == 1024)) variant = 2;
    if (variant < 0 || variant > 2) return PN2_E_ARG;
    if (b == 0 || n == 0) return PN2_OK;
    if (!xyz1 || !dist || !idx || (m > 0 && !xyz2)) return PN2_E_NULL;
    if (b > 65535) return PN2_E_TOO_LARGE;
    const bool cells_ok = m >= 64 && m <= kBqCellsMaxPoints && (long long)b * 4096 < INT_MAX;
    if (variant == 2 && !cells_ok) return PN2_E_ARG;
    // The cell list pays where the sweep is long: from ~48 M pair tests (measured, profiles/r06/three_nn_cells.txt: sem_seg FP4
    // 8 x 8192 x 1024 = 67 M: 26.5 -> 13.3 us; 32 x 4096 x 1024: 48.6 -> 20.5; part_seg FP3 16 x 2048 x 512 = 17 M: 9.1 -> 9.8, not taken)
    if (variant == 2 || (variant == 0 && cells_ok && m >= 256 && (long long)b * n * m >= (48ll << 20))) {
        // 256 unknown points per workgroup (two passes of 128 quads share one binning), 128 for small launches
        int rows = (long long)b * n >= 256ll * 256 ? 256 : 128;
        if (lab_rows > 0) rows = lab_rows;
        const float factor = lab_f == 1 ? 1.2f : lab_f == 2 ? 1.6f : lab_f == 3 ? 2.0f : lab_f == 4 ? 1.45f : kNnCellFactor;
        const int parts = (n + rows - 1) / rows;
        const size_t lds = three_nn_cells_lds(m, rows);
#define PN2_NN_CELLS(NT)                                                                                                          \
        {                                                                                                                         \
            auto kern = three_nn_cells_kernel<NT>;                                                                                \
            if (int rc = allow_dynamic_lds(kern, lds)) return rc;                                                                 \
            return launch(kern, dim3((unsigned)(parts * b)), dim3(NT), lds, as_stream(stream), n, m, rows, parts, b, factor, xyz1, xyz2,  \
                          dist, idx);                                                                                             \
        }
        if (lab_nt == 256) PN2_NN_CELLS(256)
        if (lab_nt == 1024) PN2_NN_CELLS(1024)
        PN2_NN_CELLS(kNnCellsThreads)
#undef PN2_NN_CELLS
    }
    return launch(three_nn_kernel, dim3((n + kNnPoints - 1) / kNnPoints, b), dim3(kNnThreads), 0, as_stream(stream), n, m, xyz1, xyz2,
                  dist, idx);
}

extern "C" int pn2_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
                            void *stream)
{
    return three_nn_entry(b, n, m, xyz1, xyz2, dist, idx, 0, stream);
}

extern "C" int pn2_three_nn_ex(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx, int variant,
                               void *stream)
{
    return three_nn_entry(b, n, m, xyz1, xyz2, dist, idx, variant, stream);
}

static int three_interpolate_entry(int b, int m, int c, int n, const float *points, const int *idx, const float *weight,
                                   float *out, int variant, void *stream)
{
    using namespace pn2;
    if (b < 0 || m <= 0 || c <= 0 || n < 0) return PN2_E_SHAPE;
    if (variant < 0 || variant > 3) return PN2_E_ARG;
    const long long rows = (long long)b * n;
    if (rows == 0) return PN2_OK;
    if (!points || !idx || !weight || !out) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    if (variant != 1 && c % 4 == 0 && aligned16(points) && aligned16(out) && (long long)m * c < (1ll << 31) &&
        (long long)n * c < (1ll << 31) && (long long)b * 4096 < INT_MAX) {
        constexpr int U = 4;                                       // every load of a thread's share in flight at once
        const int c4 = c / 4;
        const int rows_min = (kThreads * U + c4 - 1) / c4;
        int parts = (4096 + b - 1) / b;
        const int most = (n + rows_min - 1) / rows_min;
        if (parts > most) parts = most;
        if (parts < 1) parts = 1;
        const int rpp = (n + parts - 1) / parts;
        // beyond the Infinity Cache: stream the output. Below it non-temporal stores make THIS kernel faster (sem_seg FP4, 34 MB:
        // 10.7 -> 9.8 us) and the kernel that reads the output slower by more (a column sum behind it: 40.4 -> 42.0 us for the
        // pair; profiles/r06/nt_stores_lab.txt) -- variant 3 forces them for that measurement
        const bool nt = variant == 3 || (variant != 2 && (long long)b * n * c * 4 > (192ll << 20));
        auto kern = nt ? three_interpolate_rows_v4_kernel<U, true> : three_interpolate_rows_v4_kernel<U, false>;
        return launch(kern, dim3((unsigned)parts * b), dim3(kThreads), 0, st, n, m, c4,
                      kThreads / c4, kThreads % c4, rpp, parts, b, reinterpret_cast<const float4 *>(points), idx, weight,
                      reinterpret_cast<float4 *>(out));
    }
    if (c % 4 == 0 && aligned16(points) && aligned16(out)) {
        const long long chunks = rows * (c / 4);
        if (int rc = launch(three_interpolate_v4_kernel, dim3(grid_for(chunks)), dim3(kThreads), 0, st, chunks, m, n,
                           c / 4, reinterpret_cast<const float4 *>(points), idx, weight,
                           reinterpret_cast<float4 *>(out))) return rc;
    } else {
        const long long elems = rows * c;
        if (int rc = launch(three_interpolate_s_kernel, dim3(grid_for(elems)), dim3(kThreads), 0, st, elems, m, n, c,
                           points, idx, weight, out)) return rc;
    }
    return PN2_OK;
}

extern "C" int pn2_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx,
                                     const float *weight, float *out, void *stream)
{
    return three_interpolate_entry(b, m, c, n, points, idx, weight, out, 0, stream);
}

// pn2_three_interpolate with the kernel choice per call: 0 automatic, 1 flat first-generation kernels, 2 row kernel, 3 row kernel
// with non-temporal stores (where the row kernel does not apply, 2 and 3 fall back to the flat kernels).
extern "C" int pn2_three_interpolate_ex(int b, int m, int c, int n, const float *points, const int *idx,
                                        const float *weight, float *out, int variant, void *stream)
{
    return three_interpolate_entry(b, m, c, n, points, idx, weight, out, variant, stream);
}

extern "C" int pn2_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                          const float *weight, float *grad_points, void *stream)
{
    using namespace pn2;
    if (b < 0 || m <= 0 || c <= 0 || n < 0) return PN2_E_SHAPE;
    if (b == 0) return PN2_OK;
    if (!grad_points) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    if (int rc = clear_async(grad_points, sizeof(float) * (size_t)b * m * c, st)) return rc;   // tf_interpolate.cpp:258
    const long long elems = (long long)b * n * c;
    if (elems == 0) return PN2_OK;
    if (!grad_out || !idx || !weight) return PN2_E_NULL;
    if (int rc = launch(three_interpolate_grad_kernel, dim3(grid_for(elems)), dim3(kThreads), 0, st, elems, m, n, c,
                       grad_out, idx, weight, grad_points)) return rc;
    return PN2_OK;
}
