/*
 * pn2_oracle.c -- CPU oracle for the PointNet++ set-abstraction hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it. The product path (pointnet2_amd/) never links, imports or calls
 * anything in oracle/ and fails loudly when its HIP library is missing.
 *
 * It is a plain serial C restatement of the reference algorithms, one
 * function per reference launcher, each citing the reference file:line it
 * follows (paths relative to the charlesq34/pointnet2 tree).
 *
 * Parity pinning (see oracle/README.md and DESIGN.md section "Oracle"):
 *   - query_ball_point / group_point / group_point_grad / selection_sort /
 *     three_nn / three_interpolate / three_interpolate_grad are checked
 *     against the REAL reference functions compiled from the reference tree
 *     into oracle/_ref/ (oracle/Makefile target `ref`) and against golden
 *     fixtures generated from them (tests/golden/).
 *   - farthest_point_sample has no CPU source in the reference. The
 *     restatement here is pinned (a) against pn2_cpu_farthest_point_sample_literal,
 *     a thread-by-thread emulation of the CUDA kernel's 512-thread strided scan
 *     and shared-memory tree, and (b) on the GPU box against the reference .cu
 *     itself compiled for gfx950 into oracle/_ref/ (test-only cross-check).
 *
 * Arithmetic contract: IEEE fp32, no FMA contraction, left-to-right
 *   d = ((dx*dx) + (dy*dy)) + (dz*dz)
 * which is what `g++ -O2` (no -march, no -ffast-math) emits for the reference
 * expression. Build with: gcc -O2 -ffp-contract=off (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define PN2_FPS_REF_THREADS 512 /* blockDim of the reference launch, tf_sampling_g.cu:204 */

static inline float sqdist3(const float *p, const float *q)
{
    /* (x2-x1)*(x2-x1)+(y2-y1)*(y2-y1)+(z2-z1)*(z2-z1), evaluated left to right */
    const float dx = p[0] - q[0];
    const float dy = p[1] - q[1];
    const float dz = p[2] - q[2];
    const float xx = dx * dx;
    const float yy = dy * dy;
    const float zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}

/* ------------------------------------------------------------------------
 * farthest_point_sample
 * follows tf_ops/sampling/tf_sampling_g.cu:105-170 (kernel), :203-205 (launch)
 *
 * Restatement: out[0]=0; running min-distance `mind` starts at 1e38f
 * (:118); every round updates mind[k]=min(d(k,old),mind[k]) (:142-145) and
 * picks the arg-max. Tie rule: each of the 512 threads scans k=t,t+512,...
 * with strict `>` (:146) so it keeps its smallest k; the tree (:153-163)
 * replaces the left slot only on strict `<`, so the smallest thread id
 * wins. Net rule: maximal value, ties to the smallest (k % 512, k).
 * `temp` (n floats) may be NULL.
 * ---------------------------------------------------------------------- */
int pn2_cpu_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out)
{
    if (m <= 0) return 0; /* :106 */
    if (b < 0 || n <= 0) return -1;
    float *mind = temp ? temp : (float *)malloc(sizeof(float) * (size_t)n);
    if (!mind) return -2;
    for (int i = 0; i < b; ++i) {
        const float *cloud = inp + (size_t)i * n * 3;
        int *sel = out + (size_t)i * m;
        for (int k = 0; k < n; ++k) mind[k] = 1e38f;
        int old = 0;
        sel[0] = 0;
        for (int j = 1; j < m; ++j) {
            const float *q = cloud + (size_t)old * 3;
            float best = -1.0f;
            int besti = 0;
            int best_lane = 0;
            for (int k = 0; k < n; ++k) {
                const float d = sqdist3(cloud + (size_t)k * 3, q);
                const float d2 = fminf(d, mind[k]); /* CUDA min(float,float) */
                mind[k] = d2;
                const int lane = k % PN2_FPS_REF_THREADS;
                /* k ascends, so among equal (value,lane) the earlier k is kept */
                if (d2 > best || (d2 == best && lane < best_lane)) {
                    best = d2;
                    besti = k;
                    best_lane = lane;
                }
            }
            old = besti;
            sel[j] = old;
        }
    }
    if (!temp) free(mind);
    return 0;
}

/* Literal emulation of the same kernel: per-thread strided scans, then the
 * 9-level shared-memory tree with the exact index arithmetic of :153-163.
 * Slow (kept for pinning the tie rule of the function above). */
int pn2_cpu_farthest_point_sample_literal(int b, int n, int m, const float *inp, float *temp, int *out)
{
    enum { T = PN2_FPS_REF_THREADS };
    if (m <= 0) return 0;
    if (b < 0 || n <= 0) return -1;
    float *mind = temp ? temp : (float *)malloc(sizeof(float) * (size_t)n);
    if (!mind) return -2;
    float dists[T];
    int dists_i[T];
    for (int i = 0; i < b; ++i) {
        const float *cloud = inp + (size_t)i * n * 3;
        int old = 0;
        out[(size_t)i * m] = 0;
        for (int k = 0; k < n; ++k) mind[k] = 1e38f;
        for (int j = 1; j < m; ++j) {
            const float x1 = cloud[old * 3 + 0], y1 = cloud[old * 3 + 1], z1 = cloud[old * 3 + 2];
            for (int t = 0; t < T; ++t) { /* one "thread" at a time */
                int besti = 0;
                float best = -1.0f;
                for (int k = t; k < n; k += T) {
                    const float td = mind[k];
                    const float x2 = cloud[k * 3 + 0], y2 = cloud[k * 3 + 1], z2 = cloud[k * 3 + 2];
                    const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
                    const float d2 = fminf(d, td);
                    if (d2 != td) mind[k] = d2;
                    if (d2 > best) { best = d2; besti = k; }
                }
                dists[t] = best;
                dists_i[t] = besti;
            }
            for (int u = 0; (1 << u) < T; ++u) {
                for (int t = 0; t < (T >> (u + 1)); ++t) {
                    const int i1 = (t * 2) << u;
                    const int i2 = (t * 2 + 1) << u;
                    if (dists[i1] < dists[i2]) { dists[i1] = dists[i2]; dists_i[i1] = dists_i[i2]; }
                }
            }
            old = dists_i[0];
            out[(size_t)i * m + j] = old;
        }
    }
    if (!temp) free(mind);
    return 0;
}

/* ------------------------------------------------------------------------
 * gather_point / gather_point_grad
 * follows tf_ops/sampling/tf_sampling_g.cu:172-181 and :183-192
 * (the grad output is zero-filled by the caller in the reference,
 *  tf_sampling.cpp:174; the oracle accumulates serially in (i,j) order).
 * ---------------------------------------------------------------------- */
int pn2_cpu_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            const int a = idx[(size_t)i * m + j];
            const float *src = inp + ((size_t)i * n + a) * 3;
            float *dst = out + ((size_t)i * m + j) * 3;
            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
        }
    return 0;
}

int pn2_cpu_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j) {
            const int a = idx[(size_t)i * m + j];
            float *dst = inp_g + ((size_t)i * n + a) * 3;
            const float *src = out_g + ((size_t)i * m + j) * 3;
            dst[0] += src[0]; dst[1] += src[1]; dst[2] += src[2];
        }
    return 0;
}

/* ------------------------------------------------------------------------
 * query_ball_point
 * follows tf_ops/grouping/test/query_ball_point.cpp:19-47 (CPU twin of
 * tf_grouping_g.cu:3-36, which additionally writes pts_cnt at :34).
 * First `nsample` dataset indices in ascending order with
 *   max(sqrtf(d2),1e-20f) < radius;
 * on the first hit every slot is pre-filled with it (:34-37 of the .cpp).
 * Rows with no hit are left untouched by the reference (the harness memsets
 * idx to 0 at :94); the oracle writes zeros so the result is defined.
 * pts_cnt may be NULL.
 * ---------------------------------------------------------------------- */
int pn2_cpu_query_ball_point(int b, int n, int m, float radius, int nsample,
                             const float *xyz1, const float *xyz2, int *idx, int *pts_cnt)
{
    for (int i = 0; i < b; ++i) {
        const float *data = xyz1 + (size_t)i * n * 3;
        for (int j = 0; j < m; ++j) {
            const float *q = xyz2 + ((size_t)i * m + j) * 3;
            int *row = idx + ((size_t)i * m + j) * nsample;
            int cnt = 0;
            for (int k = 0; k < n && cnt < nsample; ++k) {
                /* note operand order of the reference: (x2-x1) with x2 the query */
                const float s = sqdist3(q, data + (size_t)k * 3);
                const float r = sqrtf(s);
                const float d = (r < 1e-20f) ? 1e-20f : r; /* std::max(r,1e-20f) */
                if (d < radius) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) row[l] = k;
                    row[cnt++] = k;
                }
            }
            if (cnt == 0)
                for (int l = 0; l < nsample; ++l) row[l] = 0;
            if (pts_cnt) pts_cnt[(size_t)i * m + j] = cnt;
        }
    }
    return 0;
}

/* The exact threshold used by the HIP kernel: smallest fp32 s* such that
 * NOT (max(sqrtf(s*),1e-20f) < radius). Because correctly rounded sqrtf is
 * monotone, in-ball <=> s < s*. Exposed so tests can pin the host-side
 * computation in pointnet2_amd against it. */
float pn2_cpu_ball_threshold(float radius)
{
    /* predicate p(s) = max(sqrtf(s),1e-20f) < radius is monotone non-increasing in s>=0 */
    if (!(radius > 0.0f)) return 0.0f;
    if (!(1e-20f < radius)) return 0.0f; /* even s=0 fails */
    union { float f; uint32_t u; } lo, hi, mid;
    lo.f = 0.0f;        /* p(lo) true */
    hi.u = 0x7f800000u; /* +inf: p false */
    while (hi.u - lo.u > 1u) {
        mid.u = lo.u + (hi.u - lo.u) / 2u;
        const float r = sqrtf(mid.f);
        const float d = (r < 1e-20f) ? 1e-20f : r;
        if (d < radius) lo = mid; else hi = mid;
    }
    return hi.f;
}

/* ------------------------------------------------------------------------
 * group_point / group_point_grad
 * follows tf_ops/grouping/test/query_ball_point.cpp:52-66 and :70-84
 * (== tf_grouping_g.cu:40-57, :61-78). Grad output zero-filled by caller.
 * ---------------------------------------------------------------------- */
int pn2_cpu_group_point(int b, int n, int c, int m, int nsample,
                        const float *points, const int *idx, float *out)
{
    const size_t rows = (size_t)m * nsample;
    for (int i = 0; i < b; ++i) {
        const float *src = points + (size_t)i * n * c;
        for (size_t r = 0; r < rows; ++r) {
            const int ii = idx[(size_t)i * rows + r];
            memcpy(out + ((size_t)i * rows + r) * c, src + (size_t)ii * c, sizeof(float) * (size_t)c);
        }
    }
    return 0;
}

int pn2_cpu_group_point_grad(int b, int n, int c, int m, int nsample,
                             const float *grad_out, const int *idx, float *grad_points)
{
    const size_t rows = (size_t)m * nsample;
    for (int i = 0; i < b; ++i) {
        float *dst = grad_points + (size_t)i * n * c;
        for (size_t r = 0; r < rows; ++r) {
            const int ii = idx[(size_t)i * rows + r];
            const float *g = grad_out + ((size_t)i * rows + r) * c;
            for (int l = 0; l < c; ++l) dst[(size_t)ii * c + l] += g[l];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------
 * selection_sort (select_top_k)
 * follows tf_ops/grouping/tf_grouping_g.cu:83-123 (== test/selection_sort.cpp:20-63
 * without the debug printf): copy dist -> out, iota -> outi, then k rounds of
 * "find first minimum of the tail (strict <), swap into slot s".
 * ---------------------------------------------------------------------- */
int pn2_cpu_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out)
{
    for (size_t r = 0; r < (size_t)b * m; ++r) {
        const float *src = dist + r * n;
        float *v = out + r * n;
        int *vi = outi + r * n;
        for (int s = 0; s < n; ++s) { v[s] = src[s]; vi[s] = s; }
        for (int s = 0; s < k && s < n; ++s) {
            int mn = s;
            for (int t = s + 1; t < n; ++t)
                if (v[t] < v[mn]) mn = t;
            if (mn != s) {
                const float tv = v[mn]; v[mn] = v[s]; v[s] = tv;
                const int ti = vi[mn]; vi[mn] = vi[s]; vi[s] = ti;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------
 * three_nn
 * follows tf_ops/3d_interpolation/tf_interpolate.cpp:60-103.
 * fp32 squared distance held in a double (:73), best* start at 1e40 (:67),
 * strict `<` cascade (:74-89) => order key (d, k). Outputs are SQUARED
 * distances; with m<3 the unused slots convert 1e40 -> +inf, index 0.
 * ---------------------------------------------------------------------- */
int pn2_cpu_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx)
{
    for (int i = 0; i < b; ++i) {
        const float *known = xyz2 + (size_t)i * m * 3;
        for (int j = 0; j < n; ++j) {
            const float *u = xyz1 + ((size_t)i * n + j) * 3;
            double bd[3] = {1e40, 1e40, 1e40};
            int bi[3] = {0, 0, 0};
            for (int k = 0; k < m; ++k) {
                const double d = (double)sqdist3(known + (size_t)k * 3, u);
                if (d < bd[0]) {
                    bd[2] = bd[1]; bi[2] = bi[1];
                    bd[1] = bd[0]; bi[1] = bi[0];
                    bd[0] = d; bi[0] = k;
                } else if (d < bd[1]) {
                    bd[2] = bd[1]; bi[2] = bi[1];
                    bd[1] = d; bi[1] = k;
                } else if (d < bd[2]) {
                    bd[2] = d; bi[2] = k;
                }
            }
            float *od = dist + ((size_t)i * n + j) * 3;
            int *oi = idx + ((size_t)i * n + j) * 3;
            for (int t = 0; t < 3; ++t) { od[t] = (float)bd[t]; oi[t] = bi[t]; }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------
 * three_interpolate / three_interpolate_grad
 * follows tf_ops/3d_interpolation/tf_interpolate.cpp:107-127 and :131-153.
 * out = (p1*w1 + p2*w2) + p3*w3 in fp32, no contraction. Grad output is
 * zero-filled by the caller (:258) and accumulated serially.
 * ---------------------------------------------------------------------- */
int pn2_cpu_three_interpolate(int b, int m, int c, int n, const float *points,
                              const int *idx, const float *weight, float *out)
{
    for (int i = 0; i < b; ++i) {
        const float *p = points + (size_t)i * m * c;
        for (int j = 0; j < n; ++j) {
            const size_t q = ((size_t)i * n + j) * 3;
            const float w1 = weight[q], w2 = weight[q + 1], w3 = weight[q + 2];
            const float *r1 = p + (size_t)idx[q] * c;
            const float *r2 = p + (size_t)idx[q + 1] * c;
            const float *r3 = p + (size_t)idx[q + 2] * c;
            float *o = out + ((size_t)i * n + j) * c;
            for (int l = 0; l < c; ++l) {
                const float a = r1[l] * w1;
                const float bq = r2[l] * w2;
                const float cq = r3[l] * w3;
                const float ab = a + bq;
                o[l] = ab + cq;
            }
        }
    }
    return 0;
}

int pn2_cpu_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out,
                                   const int *idx, const float *weight, float *grad_points)
{
    for (int i = 0; i < b; ++i) {
        float *gp = grad_points + (size_t)i * m * c;
        for (int j = 0; j < n; ++j) {
            const size_t q = ((size_t)i * n + j) * 3;
            const float *g = grad_out + ((size_t)i * n + j) * c;
            for (int l = 0; l < c; ++l) {
                /* three separate += in this order (matters when indices repeat) */
                gp[(size_t)idx[q] * c + l] += g[l] * weight[q];
                gp[(size_t)idx[q + 1] * c + l] += g[l] * weight[q + 1];
                gp[(size_t)idx[q + 2] * c + l] += g[l] * weight[q + 2];
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------
 * prob_sample
 * follows tf_ops/sampling/tf_sampling_g.cu:7-88 (cumsumKernel) and :90-104
 * (binarysearchKernel), launched at :198-201.
 * The cumulative sum is an fp32 tiled scan whose association order is part
 * of the result; this restates that order element by element:
 *   tile of 8192: groups of 4 summed as v1,(v1+v2),(v1+v2)+v3,(v3+v4)+(v1+v2),
 *   group totals combined by a Blelloch up-sweep/down-sweep, the exclusive
 *   group prefix added to each element, then `runningsum` added, with a
 *   compensated carry between tiles (:81-84).
 * temp: b*n floats (the cumulative sums), may be NULL.
 * ---------------------------------------------------------------------- */
static void cumsum_row(int n, const float *in, float *out)
{
    enum { TILE = 8192 };
    static _Thread_local float buf4[TILE];
    static _Thread_local float tot[TILE / 4];
    float runningsum = 0.0f, runningsum2 = 0.0f;
    for (int j = 0; j < n; j += TILE) {
        const int cnt = (n - j < TILE) ? (n - j) : TILE; /* n24_i */
        const int cnt4 = (cnt + 3) & ~3;                 /* n24 */
        const int n2 = cnt4 >> 2;
        for (int k = 0; k < cnt; k += 4) {
            if (k + 3 < cnt) {
                const float v1 = in[j + k];
                float v2 = in[j + k + 1];
                v2 += v1;
                float v3 = in[j + k + 2];
                float v4 = in[j + k + 3];
                v4 += v3;
                v3 += v2;
                v4 += v2;
                buf4[k] = v1; buf4[k + 1] = v2; buf4[k + 2] = v3; buf4[k + 3] = v4;
                tot[k >> 2] = v4;
            } else {
                float v = 0.0f;
                for (int k2 = k; k2 < cnt; ++k2) { v += in[j + k2]; buf4[k2] = v; }
                for (int k2 = cnt; k2 < cnt4; ++k2) buf4[k2] = v;
                tot[k >> 2] = v;
            }
        }
        int u = 0;
        for (; (2 << u) <= n2; ++u)
            for (int k = 0; k < (n2 >> (u + 1)); ++k) {
                const int i1 = (((k << 1) + 2) << u) - 1;
                const int i2 = (((k << 1) + 1) << u) - 1;
                tot[i1] += tot[i2];
            }
        for (--u; u >= 0; --u)
            for (int k = 0; k < ((n2 - (1 << u)) >> (u + 1)); ++k) {
                const int i1 = (((k << 1) + 3) << u) - 1;
                const int i2 = (((k << 1) + 2) << u) - 1;
                tot[i1] += tot[i2];
            }
        for (int k = 4; k < cnt4; k += 4) {
            const float p = tot[(k >> 2) - 1];
            buf4[k] += p; buf4[k + 1] += p; buf4[k + 2] += p; buf4[k + 3] += p;
        }
        for (int k = 0; k < cnt; ++k) out[j + k] = buf4[k] + runningsum;
        const float t = tot[n2 - 1] + runningsum2;
        const float r2 = runningsum + t;
        runningsum2 = t - (r2 - runningsum);
        runningsum = r2;
    }
}

int pn2_cpu_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out)
{
    if (n <= 0) return -1;
    float *cs = temp ? temp : (float *)malloc(sizeof(float) * (size_t)b * n);
    if (!cs) return -2;
    int base = 1;
    while (base < n) base <<= 1;
    for (int i = 0; i < b; ++i) {
        float *row = cs + (size_t)i * n;
        cumsum_row(n, inp_p + (size_t)i * n, row);
        for (int j = 0; j < m; ++j) {
            const float q = inp_r[(size_t)i * m + j] * row[n - 1];
            int r = n - 1;
            for (int k = base; k >= 1; k >>= 1)
                if (r >= k && row[r - k] >= q) r -= k;
            out[(size_t)i * m + j] = r;
        }
    }
    if (!temp) free(cs);
    return 0;
}

/* monotonic clock for bench.py's cpu_baseline leg (reference harness uses
 * clock_gettime(CLOCK_MONOTONIC), query_ball_point.cpp:12-16) */
double pn2_cpu_now(void)
{
    struct timespec tp;
    clock_gettime(CLOCK_MONOTONIC, &tp);
    return (double)tp.tv_sec + 1e-9 * (double)tp.tv_nsec;
}
