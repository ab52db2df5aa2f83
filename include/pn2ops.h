/*
 * pn2ops.h -- C ABI of libpn2ops.so: the PointNet++ set-abstraction /
 * feature-propagation operator kernels, hand-written for AMD MI355X (gfx950).
 *
 * Drop-in seam. The reference's TensorFlow op classes call free C++
 * "Launcher" functions that are declared in the op .cpp and defined in the
 * .cu. Every entry point below replaces exactly one
 * of them and keeps its argument list (same order, same meaning, same
 * row-major fp32/int32 layouts), plus
 *     - a trailing `void *stream` (a hipStream_t; NULL = the null stream), and
 *     - an `int` return: 0 ok, <0 invalid argument (PN2_E_*), >0 a hipError_t
 *       raised by the launch (the reference launchers return void and never
 *       check; validation lived in OpKernel::Compute as OP_REQUIRES).
 *
 * Ownership: the caller allocates every buffer (inputs, outputs, scratch);
 * the library holds no state and no memory, never synchronises, and only
 * enqueues work on `stream`. All pointers are DEVICE pointers. Entry points
 * are re-entrant and thread-safe.
 *
 * Gradient entry points ZERO their output themselves (the reference made the
 * caller do it: cudaMemset in tf_sampling.cpp:174, tf_grouping.cpp:204,
 * memset in tf_interpolate.cpp:258) -- a documented, deliberate difference:
 * the zero-fill is enqueued on the same stream right before the scatter.
 */
#ifndef PN2OPS_H
#define PN2OPS_H

#ifdef __cplusplus
extern "C" {
#endif

#define PN2_OK 0
#define PN2_E_NULL (-1)     /* a required pointer is NULL */
#define PN2_E_SHAPE (-2)    /* negative / zero extent where the reference requires positive */
#define PN2_E_ARG (-3)      /* attribute out of range (radius<=0, nsample<=0, npoint<=0, k<=0 ...) */
#define PN2_E_TOO_LARGE (-4)/* extent beyond what the kernels index with int32 */

/* library identification: returns "pn2ops <version> gfx950" */
const char *pn2_version(void);

/* ---- tf_ops/sampling ------------------------------------------------- */

/* replaces farthestpointsamplingLauncher(int b,int n,int m,const float*inp,float*temp,int*out)
 *   declared tf_ops/sampling/tf_sampling.cpp:94, defined tf_sampling_g.cu:203-205.
 * inp (b,n,3) f32 -> out (b,m) i32.  temp: scratch of pn2_fps_temp_floats(b,n)
 * floats (the reference allocates 32*n, tf_sampling.cpp:115); may be NULL when
 * pn2_fps_temp_floats(b,n)==0 (the register-resident tiers need no scratch).
 * Selection order and tie rule are the reference kernel's (ties -> smallest
 * (k mod 512, k)). m<=0 is a no-op like tf_sampling_g.cu:106. */
int pn2_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out, void *stream);
long long pn2_fps_temp_floats(int b, int n);

/* replaces gatherpointLauncher(b,n,m,inp,idx,out)  tf_sampling.cpp:125, tf_sampling_g.cu:206-208
 * inp (b,n,3), idx (b,m) -> out (b,m,3) */
int pn2_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, void *stream);

/* replaces scatteraddpointLauncher(b,n,m,out_g,idx,inp_g)  tf_sampling.cpp:150, tf_sampling_g.cu:209-211
 * out_g (b,m,3), idx (b,m) -> inp_g (b,n,3), zero-filled here then accumulated */
int pn2_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g, void *stream);

/* replaces probsampleLauncher(b,n,m,inp_p,inp_r,temp,out)  tf_sampling.cpp:65, tf_sampling_g.cu:198-201
 * inp_p (b,n) weights, inp_r (b,m) uniforms in [0,1) -> out (b,m); temp: b*n floats */
int pn2_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out, void *stream);

/* ---- tf_ops/grouping -------------------------------------------------- */

/* replaces queryBallPointLauncher(b,n,m,radius,nsample,xyz1,xyz2,idx,pts_cnt)
 *   tf_ops/grouping/tf_grouping.cpp:66, tf_grouping_g.cu:125-128 (CPU twin test/query_ball_point.cpp:19-47)
 * xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries -> idx (b,m,nsample), pts_cnt (b,m).
 * Rows with no point in the ball are written as zeros (the reference leaves them
 * uninitialised; its harness pre-zeroes, query_ball_point.cpp:94). */
int pn2_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                         int *idx, int *pts_cnt, void *stream);

/* replaces selectionSortLauncher(b,n,m,k,dist,outi,out)  tf_grouping.cpp:108, tf_grouping_g.cu:129-132
 * dist (b,m,n) -> outi (b,m,n) i32, out (b,m,n) f32; the first k of each row are the k smallest, ascending */
int pn2_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream);

/* knn_point (tf_grouping.py:48-73) in one kernel, without the (b,m,n) distance/index tensors: the distance
 * row ((dx*dx)+(dy*dy))+(dz*dz) of a query is built in LDS, the same k swap rounds as selectionSortLauncher
 * run on it, and only the first k (value, index) pairs are written -- identical to slicing the reference's
 * outputs, ties included. xyz1 (b,n,3), xyz2 (b,m,3) -> val (b,m,k) f32, idx (b,m,k) i32.
 * PN2_E_TOO_LARGE for n > 14336 or k > n (callers keep the matrix + pn2_selection_sort path). */
int pn2_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                  void *stream);

/* replaces groupPointLauncher(b,n,c,m,nsample,points,idx,out)  tf_grouping.cpp:142, tf_grouping_g.cu:133-136
 * points (b,n,c), idx (b,m,nsample) -> out (b,m,nsample,c) */
int pn2_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out,
                    void *stream);

/* replaces groupPointGradLauncher(b,n,c,m,nsample,grad_out,idx,grad_points)  tf_grouping.cpp:173, tf_grouping_g.cu:137-141
 * grad_out (b,m,nsample,c), idx -> grad_points (b,n,c), zero-filled here then accumulated */
int pn2_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                         float *grad_points, void *stream);

/* ---- tf_ops/3d_interpolation ------------------------------------------ */

/* replaces threenn_cpu(b,n,m,xyz1,xyz2,dist,idx)  tf_ops/3d_interpolation/tf_interpolate.cpp:60-103
 * xyz1 (b,n,3) unknown, xyz2 (b,m,3) known -> dist (b,n,3) SQUARED distances ascending, idx (b,n,3) */
int pn2_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx, void *stream);

/* replaces threeinterpolate_cpu(b,m,c,n,points,idx,weight,out)  tf_interpolate.cpp:107-127
 * points (b,m,c), idx/weight (b,n,3) -> out (b,n,c) */
int pn2_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx, const float *weight,
                          float *out, void *stream);

/* replaces threeinterpolate_grad_cpu(b,n,c,m,grad_out,idx,weight,grad_points)  tf_interpolate.cpp:131-153
 * grad_out (b,n,c), idx/weight (b,n,3) -> grad_points (b,m,c), zero-filled here then accumulated */
int pn2_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                               const float *weight, float *grad_points, void *stream);

/* ---- run-to-run reproducible gradients (no reference counterpart) -------------
 * Same contracts as pn2_gather_point_grad / pn2_group_point_grad / pn2_three_interpolate_grad above
 * (tf_sampling_g.cu:182-190, tf_grouping_g.cu:60-78, tf_interpolate.cpp:131-153), but the scatter-add
 * accumulates 64-bit fixed-point integers, so the result does not depend on the order in which the
 * atomics land: two calls on the same inputs return identical bits. `ws` is device scratch of
 * pn2_det_grad_ws_bytes(b, rows, c) bytes (rows = n for gather/group, m for three_interpolate; c = 3 for
 * gather); it needs no initialisation. Non-finite gradients fall back to the fp32-atomic accumulation. */
long long pn2_det_grad_ws_bytes(int b, int rows, int c);
int pn2_gather_point_grad_det(int b, int n, int m, const float *out_g, const int *idx, float *inp_g, void *ws,
                              void *stream);
int pn2_group_point_grad_det(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                             float *grad_points, void *ws, void *stream);
int pn2_three_interpolate_grad_det(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                   const float *weight, float *grad_points, void *ws, void *stream);

/* ---- the same gradients as a segmented reduction (no reference counterpart) ----------------------
 * idx is inverted first (counting sort by target row), then one lane group sums the grad_out rows of
 * every output row: no float atomics, no zero-fill, ~3x the throughput of the atomic scatter from 16
 * channels up. deterministic != 0: per-element 64-bit fixed-point sums (identical bits on every run).
 * ws: pn2_seg_grad_ws_bytes(b, rows, entries) bytes of uninitialised device scratch
 * (group_point: rows = n, entries = m * nsample; three_interpolate: rows = m, entries = 3 * n). */
long long pn2_seg_grad_ws_bytes(int b, int rows, long long entries);
/* Host logic, no device work (tests, maintainers): how the default mode of pn2_*_grad_seg lays out the long-row part of its
 * reduction for out_rows = b * rows target rows of c channels with `entries` references per cloud. long_from: rows with that
 * many references or more are summed by a whole workgroup each (twice the average row, between 32 and 64); long_blocks: the
 * number of such workgroups = the stride of the rows one of them looks at (rows w, w + long_blocks, ...): coprime with `rows`,
 * residue at its golden section, so that the low point numbers of every cloud -- where a ball query's padding piles the
 * references up, tf_grouping_g.cu:24-31 -- spread evenly over the workgroups. */
int pn2_seg_grad_plan(int rows, long long entries, int c, long long out_rows, int *long_from, int *long_blocks);
int pn2_group_point_grad_seg(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                             float *grad_points, void *ws, int deterministic, void *stream);
int pn2_three_interpolate_grad_seg(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                   const float *weight, float *grad_points, void *ws, int deterministic, void *stream);

/* ---- grouped local MLP + max-pool of a set-abstraction layer, fused, on the matrix cores ---------
 * (no reference kernel; replaces for INFERENCE the TF graph of utils/pointnet_util.py:44-50 + :117-127:
 *  group_point(xyz)-new_xyz ++ group_point(points) -> 3 x [conv2d 1x1 + batch_norm + ReLU] -> reduce_max
 *  over nsample). fp32 in, fp32 out, fp32 accuracy: every product is evaluated as six
 *  bf16 MFMA terms on three-level bf16 operands (error of an fp32 evaluation; csrc/sa_mlp.hip).
 *   xyz (b,n,3), new_xyz (b,m,3), points (b,n,cfeat) or NULL when cfeat == 0, idx (b,m,nsample) i32
 *   -> out (b,m,c3).  Input channel order is the reference's: [relative xyz (3), features (cfeat)].
 * Three kernels behind one entry: weights resident in LDS (3 + cfeat <= 32, widths within (64,96,128);
 * nsample 16 or a multiple of 32), streamed through LDS (up to 384 input channels, widths within
 * (128,128,256); nsample a multiple of 32), or the cooperative kernel for wide stacks -- (256,256,512) and the
 * group_all level's (256,512,1024), any number of input channels, any nsample (the tail of the group is
 * masked). idx == NULL together with new_xyz == NULL selects the group_all form (sample_and_group_all,
 * pointnet_util.py:59-84: m = 1, nsample = n, the group is the whole cloud, no centroid subtraction).
 * PN2_E_TOO_LARGE outside: callers keep the unfused path.
 * Weights: w_i (cin_i, cout_i) row-major = the reference's conv kernel [1,1,cin,cout] (tf_util.py:113-117)
 * with batch norm folded in by the caller; xyz_first says whether the rows of w1 are [xyz, features]
 * (pointnet_util.py:50) or [features, xyz] (:184, MSG). pn2_sa_mlp3_pack (host code) permutes them into the
 * order the kernel consumes and splits every weight into its three bf16 levels: wpacked / bpacked of the sizes
 * pn2_sa_mlp3_config reports in 4-byte words (info4 = {kind: 0 resident / 1 streamed / 2 cooperative, output
 * tiles of the three layers}; 6 bytes per padded weight), uploaded by the caller. */
int pn2_sa_mlp3_config(int cin, int c1, int c2, int c3, int nsample, int *info4, long long *w_floats,
                       long long *b_floats);
int pn2_sa_mlp3_pack(int cin, int c1, int c2, int c3, int nsample, int xyz_first, const float *w1, const float *bias1,
                     const float *w2, const float *bias2, const float *w3, const float *bias3, float *wpacked,
                     float *bpacked);
/* ws: scratch of pn2_sa_mlp3_ws_bytes(b, n, m, 3 + cfeat, c1, c2, c3, nsample) bytes (0 for the resident kernel and most
 * cooperative shapes: NULL is fine then). The streamed kernel evaluates the FEATURE part of layer 1 once per point into
 * it (W1^T [f_j, xyz_j - c] = W1f^T f_j + W1x^T (xyz_j - c): the first term does not depend on the centroid, and a point
 * belongs to nsample * m / n groups) and starts every sample from its point's row. Stacks whose last layer is wider than
 * 512 (the group_all level's 256-512-1024) keep the second layer's output there for the GEMM that runs the last layer. */
long long pn2_sa_mlp3_ws_bytes(int b, int n, int m, int cin, int c1, int c2, int c3, int nsample);
int pn2_sa_mlp3_maxpool(int b, int n, int m, int nsample, int cfeat, const float *xyz, const float *new_xyz,
                        const float *points, const int *idx, int c1, int c2, int c3, const float *wpacked,
                        const float *bpacked, float *out, void *ws, void *stream);
/* The same with the organisation of the resident kernel chosen by the caller: variant 0 = by the size rule, 1 = one 32-sample
 * item per wave (two waves per SIMD), 2 / 3 = two items per wave, half a layer apart (csrc/sa_mlp.hip: sa_mlp3_pair_kernel), one /
 * two waves per SIMD; nsample 32 or 16, at most 16 input channels and widths up to (64, 64, 128), else ignored. Results are
 * bit-identical. */
int pn2_sa_mlp3_maxpool_ex(int b, int n, int m, int nsample, int cfeat, const float *xyz, const float *new_xyz,
                           const float *points, const int *idx, int c1, int c2, int c3, const float *wpacked,
                           const float *bpacked, float *out, void *ws, int variant, void *stream);
/* The reference's other pooling modes behind the same stack (utils/pointnet_util.py:128-140): pooling 0 max
 * (= pn2_sa_mlp3_maxpool), 1 avg (tf.reduce_mean over the group), 2 weighted_avg (weights exp(-5 |grouped_xyz|) normalised
 * over the group, :132-138), 3 max_and_avg (out (b, m, 2 c3) = concat([avg, max]), :139-142). Same operands, same packed
 * weights, same scratch (pn2_sa_mlp3_ws_bytes). Modes 1-3 are served for the stacks the resident and the streamed kernel
 * cover (pn2_sa_mlp3_config kind 0 / 1: widths up to (128, 128, 256), nsample 16 or a multiple of 32);
 * pn2_sa_mlp3_pool_supported says 1 / 0, the cooperative kernel's shapes return PN2_E_TOO_LARGE and the caller evaluates
 * the level layer by layer. idx / new_xyz NULL (group_all) only with pooling 0. */
int pn2_sa_mlp3_pool_supported(int cin, int c1, int c2, int c3, int nsample, int pooling);
int pn2_sa_mlp3_pool(int b, int n, int m, int nsample, int cfeat, const float *xyz, const float *new_xyz,
                     const float *points, const int *idx, int c1, int c2, int c3, const float *wpacked,
                     const float *bpacked, int pooling, float *out, void *ws, void *stream);

/* ---- a feature-propagation layer behind three_nn, fused, on the matrix cores --------------------------
 * (no reference kernel; replaces for INFERENCE the TF graph of utils/pointnet_util.py:212-226: the
 *  inverse-distance weights, three_interpolate, concat([interpolated, points1]) and 2-3 x [conv2d 1x1 +
 *  batch_norm + ReLU]). fp32 in, fp32 out, fp32 accuracy (the arithmetic of pn2_sa_mlp3_maxpool).
 *   points2 (b,m,c2) known features, points1 (b,n,c1) skip features or NULL when c1 == 0,
 *   idx (b,n,3) i32 and dist (b,n,3) squared distances as pn2_three_nn writes them -> out (b,n,widths[nlayers-1]).
 * nlayers 2 or 3, widths <= 256 (padded to the instantiated tile shapes; PN2_E_TOO_LARGE outside: callers keep
 * the unfused path). w[i] (cin_i, cout_i) row-major with batch norm folded in by the caller, rows of w[0] in
 * the reference's concat order [interpolated, points1]; pn2_fp_mlp_pack (host code) permutes them into the
 * streams the kernels consume (sizes from pn2_fp_mlp_config; tiles4 = skip-link / layer tile counts).
 * The first layer is linear and so is the interpolation: W1^T [interp(points2), points1] = interp(points2 . W1a) +
 * W1b^T points1. Q = points2 . W1a is evaluated once per KNOWN point into `ws` (pn2_fp_mlp_ws_bytes bytes, caller's),
 * rows of Q are interpolated instead of rows of points2, and only the skip-link part of layer 1 is matrix work per
 * unknown point. */
int pn2_fp_mlp_config(int c2, int c1, int nlayers, const int *widths, int kind, int *tiles4, long long *w_floats,
                      long long *b_floats);
int pn2_fp_mlp_pack(int c2, int c1, int nlayers, const int *widths, int kind, const float *const *w,
                    const float *const *bias, float *wpacked, float *bpacked);
long long pn2_fp_mlp_ws_bytes(int b, int m, int c2, int c1, int nlayers, const int *widths, int kind);
int pn2_fp_mlp(int b, int n, int m, int c2, int c1, const float *points2, const float *points1, const int *idx,
               const float *dist, int nlayers, const int *widths, int kind, const float *wpacked, const float *bpacked,
               float *out, void *ws, void *stream);
/* kind selects the kernel (and with it the packed layout): 0 = one wave per 32 points, weights streamed
 * through LDS (many points: sem_seg FP4, 65536 points, 88 TFLOP/s); 1 = four waves share 32 points and split
 * each layer's output tiles (few points, wide layers: a 512-point level is otherwise 16 serial MFMA chains).
 * Same results either way; pack with the kind you launch with. */

/* ---- fused entry points (no reference counterpart) ---- */

/* farthest_point_sample + gather_point(inp, out) in one launch: what
 * pointnet_util.py:40 computes with two ops. out (b,m) i32 as pn2_farthest_point_sample,
 * out_xyz (b,m,3) f32 = inp[out] (bit-exact copies). temp as pn2_farthest_point_sample. */
int pn2_farthest_point_sample_gather(int b, int n, int m, const float *inp, float *temp, int *out, float *out_xyz,
                                     void *stream);

/* farthest_point_sample (+ gather_point when out_xyz is not NULL) for input the caller BELIEVES to be in farthest-point
 * order already -- the second and later levels of every network sample from the previous level's samples
 * (models/pointnet2_sem_seg.py:28-31, pointnet2_cls_ssg.py:27-29), whose first m points are, up to exact ties under the
 * renumbered tie rule (tf_sampling_g.cu:146,153-163) and clouds that ran out of distinct points, selected as 0, 1, ..., m-1.
 * The belief is CHECKED in parallel (n independent prefix-minimum rows, no dependent rounds) and the chain runs only for
 * the clouds where it fails: the result is pn2_farthest_point_sample's for any input. ws: pn2_fps_ordered_ws_bytes(b)
 * bytes, zeroed ONCE by the caller (every call leaves it zeroed). Outside 2 <= m <= min(n, 1024), n <= 2048 the call is
 * the plain operator (PN2_E_TOO_LARGE beyond 16384 points, where that one needs its temp buffer). */
long long pn2_fps_ordered_ws_bytes(int b);
int pn2_farthest_point_sample_ordered(int b, int n, int m, const float *inp, int *out, float *out_xyz, void *ws, void *stream);
/* test hook: the check alone. flags (b ints, zeroed by the caller) comes back 1 exactly for the clouds whose sampling is not
 * 0 .. m-1. PN2_E_ARG outside the short cut's envelope. */
int pn2_fps_ordered_check(int b, int n, int m, const float *inp, int *flags, void *stream);

/* query_ball_point + group_point(xyz1, idx) - centroid in one pass over the
 * LDS-resident cloud: what pointnet_util.py:44-46 computes with three ops.
 * grouped_xyz (b,m,nsample,3) = xyz1[idx] - xyz2[:, :, None] when subtract_centroid!=0,
 * else the plain group. idx / pts_cnt as pn2_query_ball_point (either may be NULL). */
int pn2_query_ball_group_xyz(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                             int subtract_centroid, int *idx, int *pts_cnt, float *grouped_xyz, void *stream);

/* pn2_query_ball_group_xyz with the kernel choice as PER-CALL arguments (no process state): kernel 0 =
 * automatic, 1 = sweep kernel, 2 = cell-list kernel whenever it fits, 3 = cell-list kernel with 512-thread
 * workgroups; cells_qpb = queries per workgroup of the cell-list kernel (0 = automatic). grouped_xyz may be
 * NULL (plain query_ball_point). Used by the parity tests to force every kernel at every shape and by
 * scripts/bq_probe.py; results are identical whatever the choice. */
int pn2_query_ball_group_xyz_ex(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                                int subtract_centroid, int *idx, int *pts_cnt, float *grouped_xyz, int kernel,
                                int cells_qpb, void *stream);

/* Multi-radius form for multi-scale grouping (pointnet_sa_module_msg, utils/pointnet_util.py:175-186: one
 * query_ball_point + group_point + centroid subtraction PER RADIUS over the same xyz / new_xyz): the cloud is
 * staged and binned ONCE per workgroup (cells sized for the smallest radius) and queried once per radius.
 * radii / nsamples: HOST arrays of nscales (<= 4) entries; idx / pts_cnt / grouped_xyz: HOST arrays of nscales
 * DEVICE pointers, scale i shaped (b,m,nsamples[i]) / (b,m) / (b,m,nsamples[i],3); the arrays themselves or
 * single entries of pts_cnt and of one of idx / grouped_xyz may be NULL. Every output is bit-identical to
 * pn2_query_ball_group_xyz called per radius. PN2_E_TOO_LARGE -- before anything is launched -- for n > 8192 and
 * whenever no LDS geometry fits the combination (e.g. n = 8192 with nsample >= 127 on a later radius, n = 8000 with
 * nsample 256): call the single-radius operator per radius then, as the Python operator does. */
int pn2_query_ball_group_xyz_msg(int b, int n, int m, int nscales, const float *radii, const int *nsamples,
                                 const float *xyz1, const float *xyz2, int subtract_centroid, int *const *idx,
                                 int *const *pts_cnt, float *const *grouped_xyz, void *stream);

/* pn2_three_nn with the kernel chosen by the caller (results never depend on it; tests force each, scripts time them):
 * 0 = the library's choice, 1 = the sweep of every known point (round 1), 2 = the cell list (round 6: the known points binned
 * once per workgroup, 3 x 3 x 3 cells visited per unknown point, an exactness test on the third-best distance and a sweep for
 * the points that fail it; PN2_E_ARG below 64 or above 8192 known points). */
int pn2_three_nn_ex(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx, int variant, void *stream);
/* pn2_group_point / pn2_three_interpolate with the kernel choice per call (parity tests force every kernel,
 * scripts/bw_probe.py times them). group: 0 automatic, 1 flat first-generation kernels, 2 row kernels,
 * 3 row kernels with non-temporal stores; three_interpolate: 0 automatic, 1 flat, 2 row kernel, 3 row kernel with
 * non-temporal stores. */
int pn2_group_point_ex(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out,
                       int variant, void *stream);
int pn2_three_interpolate_ex(int b, int m, int c, int n, const float *points, const int *idx, const float *weight,
                             float *out, int variant, void *stream);

/* farthest_point_sample (register-resident tier, n <= 16384) with an explicit workgroup geometry: T threads
 * in {256, 512, 1024}, P points per thread (a power of two, T * P >= 512 * ceil(n / 512)). Every geometry
 * returns the same indices; tests cover all of them, scripts/ time them. Stateless. */
int pn2_farthest_point_sample_ex(int T, int P, int b, int n, int m, const float *inp, int *out, void *stream);

/* farthest_point_sample [+ gather_point when out_xyz != NULL] with the TIER chosen by the caller. Every tier returns the
 * reference's indices (tf_sampling_g.cu:105-170); tests force each one, scripts time them.
 *   PN2_FPS_AUTO    what pn2_farthest_point_sample does: the batched tier at 513..8192 rank slots (= 512 * ceil(n / 512)) with
 *                   npoint >= 256, the pruned tier at 4097..8192 with 128 <= npoint < 256, the full tier otherwise;
 *   PN2_FPS_FULL    every point's running distance is updated against every new sample (csrc/fps_body.h);
 *   PN2_FPS_PRUNED  points dealt to the threads by a kd-tree built in LDS; per round only the groups whose bounding box
 *                   lies within the current farthest-point distance of the new sample are updated -- exactly the points
 *                   the reference's min() can change (csrc/fps_pruned_body.h). PN2_E_ARG outside 2049..8192 rank slots;
 *   PN2_FPS_BATCH   (round 6) the pruned tier's groups, several samples per arg-max exchange: a candidate list per batch, a wave of
 *                   its own for the arg-max chain, eight updater waves behind it (csrc/fps_batch_body.h). PN2_E_ARG outside
 *                   513..8192 rank slots. */
#define PN2_FPS_AUTO 0
#define PN2_FPS_FULL 1
#define PN2_FPS_PRUNED 2
#define PN2_FPS_BATCH 3
int pn2_farthest_point_sample_variant(int variant, int b, int n, int m, const float *inp, float *temp, int *out,
                                      float *out_xyz, void *stream);

/* The whole xyz half of sample_and_group (pointnet_util.py:40-46) in ONE launch, with the ball
 * queries overlapped under the farthest-point-sampling chain: producer workgroups (one per cloud)
 * publish each sample as they select it, consumer workgroups on the other CUs run query j as soon as
 * sample j exists. Outputs are bit-identical to the separate operators:
 *   fps_idx (b,m) i32, new_xyz (b,m,3) f32, idx (b,m,nsample) i32, pts_cnt (b,m) i32,
 *   grouped_xyz (b,m,nsample,3) f32 (minus the centroid when subtract_centroid != 0).
 * ws: device scratch of pn2_sample_and_group_ws_bytes(b,m) bytes (zeroed here on `stream`): the sample
 * granules and a status word.
 * Returns PN2_E_TOO_LARGE for shapes outside the overlapped launch's envelope (b > 256, n > 8192,
 * n < 64, nsample > 256) and when the device cannot hold all b producer workgroups at once
 * (occupancy query at launch): use pn2_farthest_point_sample_gather + pn2_query_ball_group_xyz then.
 * Clouds too large for a cell list beside their sorted copy in LDS (n > ~7000) are served by exactly those two launches from
 * inside this call (their consumers would have to sweep the whole cloud per query and no longer hide under the chain:
 * 478 us against 459 at b = 8, 8192 -> 1024); same outputs, ws untouched.
 * Inside a captured HIP graph (hipStreamIsCapturing on `stream`) this call enqueues exactly those two launches as well: a
 * replayed overlapped launch would carry the same tag as the replay before it and depend on the clear alone, and round 5's
 * soak of a serving loop saw such launches accept granules that were not theirs (profiles/r06/stale_granules.md). The
 * overlapped launch INSIDE a graph is pn2_sample_and_group_xyz_gen(generation = PN2_GENERATION_DEVICE). */
int pn2_sample_and_group_xyz(int b, int n, int m, float radius, int nsample, const float *xyz, void *ws,
                             int *fps_idx, float *new_xyz, int *idx, int *pts_cnt, float *grouped_xyz,
                             int subtract_centroid, void *stream);
long long pn2_sample_and_group_ws_bytes(int b, int m);
/* The same launch without the per-call clear of ws: the caller manages generations. Zero ws ONCE when it is
 * allocated, then pass a generation that no earlier launch on this workspace used (1, 2, 3, ... 0xfffffffe: what an
 * earlier generation left behind can never be mistaken for a published sample). One ws per stream;
 * generation 0 is PN2_E_ARG; re-zero ws before wrapping around. Saves a memset launch (~5 us) per call.
 * generation = PN2_GENERATION_DEVICE (round 6): the launch numbers ITSELF -- every workgroup of a cloud arrives at a counter
 * word of ws with one returning atomic add, which gives the cloud's producer and consumers the same launch ordinal and
 * consecutive launches different tags, with no clear and no number from the host. This is the form for captured graphs
 * (frozen arguments): zero ws ONCE when it is allocated (outside the graph), give every captured call site a workspace of
 * its own, never use a workspace in two forms or from two launches at the same time. Costs the chain one atomic round trip
 * (~2 us) per launch. */
#define PN2_GENERATION_DEVICE 0xffffffffu
int pn2_sample_and_group_xyz_gen(int b, int n, int m, float radius, int nsample, const float *xyz, void *ws,
                                 unsigned generation, int *fps_idx, float *new_xyz, int *idx, int *pts_cnt,
                                 float *grouped_xyz, int subtract_centroid, void *stream);
/* Byte offset inside ws of the launch status word (u32): 0 ok, 1 = a consumer stopped waiting for its producer
 * after ~10 s (cannot happen while the device makes progress; the launch's outputs are then incomplete). The
 * library never synchronises, so a caller that wants to assert forward progress reads the word after
 * synchronising the stream. */
long long pn2_sample_and_group_status_offset(int b, int m);
/* pn2_sample_and_group_xyz[_gen] with the organisation of the launch chosen by the caller (outputs never depend on it):
 * fps_variant = FPS tier of the producer workgroups (PN2_FPS_AUTO / PN2_FPS_FULL / PN2_FPS_PRUNED / PN2_FPS_BATCH as in
 * pn2_farthest_point_sample_variant); consumers = persistent consumer workgroups per cloud (0 = the library's choice, which
 * includes the two launches for clouds beyond ~7000 points; > 0 = always the overlapped launch; each
 * stages its cloud once and walks the 64-query ranges c, c + consumers, ... in publish order; the grid is
 * b * (1 + consumers) workgroups, consumers <= 16). generation 0 = clear `ws` first (eager; the two launches inside a capture),
 * PN2_GENERATION_DEVICE as in pn2_sample_and_group_xyz_gen. */
int pn2_sample_and_group_xyz_ex(int b, int n, int m, float radius, int nsample, const float *xyz, void *ws, unsigned generation,
                                int fps_variant, int consumers, int *fps_idx, float *new_xyz, int *idx, int *pts_cnt,
                                float *grouped_xyz, int subtract_centroid, void *stream);


/* ---- one call per level (inference) ---------------------------------------------------------------------
 * pn2_sa_level = pointnet_sa_module (utils/pointnet_util.py:87-154) for max pooling and three layers: the overlapped
 * sample-and-group launch (or, outside its envelope, pn2_farthest_point_sample_gather + pn2_query_ball_group_xyz) followed
 * by pn2_sa_mlp3_maxpool, enqueued by ONE call. ws_sample: pn2_sample_and_group_ws_bytes(b, m) bytes, handled as in
 * pn2_sample_and_group_xyz_gen (generation > 0: zeroed once by the caller, a fresh generation per call, or
 * PN2_GENERATION_DEVICE for captured graphs) or cleared here (generation = 0; the two launches inside a capture); NULL: never the overlapped launch, always the two-launch path; fps_temp: pn2_fps_temp_floats(b, n) floats or NULL when that is 0; ws_mlp: pn2_sa_mlp3_ws_bytes(...)
 * bytes or NULL when that is 0; wpacked / bpacked from pn2_sa_mlp3_pack. Outputs as the two operators' (all required).
 * pn2_fp_level = pointnet_fp_module (:199-229): pn2_three_nn followed by pn2_fp_mlp (dist / idx are outputs too). */
int pn2_sa_level(int b, int n, int m, float radius, int nsample, int cfeat, const float *xyz, const float *points,
                 void *ws_sample, unsigned generation, float *fps_temp, int c1, int c2, int c3, const float *wpacked,
                 const float *bpacked, int *fps_idx, float *new_xyz, int *idx, int *pts_cnt, float *grouped_xyz, float *out,
                 void *ws_mlp, void *stream);
int pn2_fp_level(int b, int n, int m, int c2, int c1, const float *xyz1, const float *xyz2, const float *points2,
                 const float *points1, int nlayers, const int *widths, int kind, const float *wpacked, const float *bpacked,
                 float *dist, int *idx, float *out, void *ws, void *stream);
/* pn2_sa_level for a level whose xyz is believed to be in farthest-point order (the previous level's new_xyz):
 * pn2_farthest_point_sample_ordered (ws_ordered: its workspace) + pn2_query_ball_group_xyz + pn2_sa_mlp3_maxpool. Same
 * outputs as pn2_sa_level for any input. */
int pn2_sa_level_ordered(int b, int n, int m, float radius, int nsample, int cfeat, const float *xyz, const float *points,
                         void *ws_ordered, int c1, int c2, int c3, const float *wpacked, const float *bpacked, int *fps_idx,
                         float *new_xyz, int *idx, int *pts_cnt, float *grouped_xyz, float *out, void *ws_mlp, void *stream);

/* ---- training mode of the shared MLPs (second half) ---------------------------
 *
 * The reference builds every SA / FP level with is_training = True (train.py:188): each of the level's
 * tf_util.conv2d 1x1 layers is followed by batch normalisation over ALL rows of the level
 * (tf_util.py:512-531: batch moments, eps 1e-3, moving averages updated with bn_decay, train.py:96-104) and a
 * ReLU; SA levels end in reduce_max over nsample (utils/pointnet_util.py:113-127), FP levels do not (:222-226).
 * pn2_mlp_train_forward / _backward run such a stack in training mode on the matrix cores:
 *   forward, per layer l:   z_l = h_{l-1} W_l + b_l         one MFMA GEMM launch; its prologue forms
 *                                                           h_{l-1} = relu(a_{l-1} z_{l-1} + c_{l-1}) from the stored
 *                                                           pre-norm tensor (layer 1: gathers the grouped rows from
 *                                                           xyz / points / idx -- the (b,m,nsample,C) tensor never
 *                                                           exists); its epilogue accumulates sum z, sum z^2 per
 *                                                           channel (and, last SA layer, the per-group max / min of z
 *                                                           with their sample numbers: BN + ReLU are monotone per
 *                                                           channel, so the pool commutes with them)
 *                           (a_l, c_l) from the batch moments, running statistics updated in place
 *   backward, per layer:    dz_l = s_l dy_l - c0_l - c1_l z_l   (batch-norm backward written per channel)
 *                           dW_l = h_{l-1}^T dz_l           MFMA, contraction over the rows
 *                           dy_{l-1} = (dz_l W_l^T) . [h_{l-1} > 0]    MFMA GEMM; epilogue accumulates the two
 *                                                           per-channel sums batch-norm backward needs one layer down
 * fp32 in, fp32 out; products on the bf16 matrix pipe as six bf16 terms (see pn2_sa_mlp3_maxpool).
 * Everything is enqueued on `stream` from ONE call per direction; no host synchronisation; all buffers the
 * caller's. rows must be a multiple of 32; widths multiples of 4 (layer 1's input width is free when grouped).
 */
typedef struct pn2_group_src {     /* the rows of a grouped tensor (pointnet_util.py:44-50, :59-84): row = (cloud*m + j)*nsample + k */
    int b, n, m, nsample;
    int cfeat;                     /* channels of `points` (0: xyz only) */
    int xyz_first;                 /* channel order [xyz, features] (:50, single-scale) or [features, xyz] (:184, MSG) */
    const float *xyz;              /* (b,n,3) */
    const float *new_xyz;          /* (b,m,3), subtracted from the grouped xyz; NULL: no centroid (group_all) */
    const float *points;           /* (b,n,cfeat) or NULL */
    const int *idx;                /* (b,m,nsample); NULL: sample k of the group is point k (group_all: m = 1, nsample = n) */
} pn2_group_src;

typedef struct pn2_bn_layer {
    int cin, cout;
    const float *weight;           /* W[k][n] = weight[k*w_stride_k + n*w_stride_n]; a torch conv kernel (cout,cin,1,1): 1, cin */
    long long w_stride_k, w_stride_n;
    const float *bias;             /* (cout) or NULL */
    const float *gamma, *beta;     /* batch-norm scale / offset (cout) */
    float *running_mean, *running_var;   /* (cout), updated in place with `momentum`; NULL: not tracked */
    float momentum, eps;           /* torch convention: new = (1-momentum)*old + momentum*batch; momentum = 1 - bn_decay */
    float *z;                      /* (rows, cout) pre-norm tensor h W (WITHOUT the bias, which batch norm cancels), written by forward, read by backward */
    float *save;                   /* (4, cout): batch mean of z (= the layer's batch mean - bias), 1/sqrt(var+eps), a = gamma*invstd, c = beta - a*mean */
    float *grad_weight;            /* backward: same strides as weight */
    float *grad_gamma, *grad_beta; /* backward: (cout) */
    int grad_accumulate;           /* backward: 0 = the three gradients are written, 1 = ADDED to what the buffers hold (one fp32
                                      add per element, as a framework's own accumulation into .grad would do) */
    int running_var_biased;        /* which batch variance enters running_var: 0 = the UNBIASED one, var * N / (N - 1)
                                      (torch.nn.BatchNorm; also tf.contrib.layers.batch_norm, the reference's live path
                                      tf_util.py:526-531, wherever it takes the fused kernel); 1 = the biased one, var
                                      (tf.nn.moments: contrib's non-fused path, the default of the TF 1.2 the reference
                                      was tested on). The struct gained this trailing field in library version 0.2.0:
                                      callers MUST zero-initialise pn2_bn_layer (memset / = {0}) before filling it. */
} pn2_bn_layer;

/* pool_rows: 0 = no pooling, out is (rows, cout_L) = relu(bn(z_L)); else the group size (nsample: 16 or a multiple of
 * 32), out (rows/pool_rows, cout_L) = max over the group; argsel (same shape, i32) receives the sample number the
 * gradient flows to and zsel the selected pre-norm value (both needed by backward).
 * group != NULL: layer 1 reads the grouped rows; else x is the (rows, cin_1) input. */
/* Organisation of the passes (csrc/train_mlp.hip) is chosen by size rules -- results never depend on it. The *_ex entry
 * points take the rules' overrides as a per-call argument (A/B timing, and the tests that force every variant); the
 * library reads no environment variable and keeps no mode. A zeroed struct (or opts == NULL) = every rule automatic. */
enum { PN2_OPT_AUTO = 0, PN2_OPT_OFF = 1, PN2_OPT_ON = 2 };
typedef struct pn2_train_opts {
    int top_stored;                /* keep the pooled top layer's pre-norm tensor z_L (AUTO: kept below 32 MB) */
    int top_sparse;                /* routed part of that layer's weight gradient on the vector units (AUTO: from 2^24 row x inputs) */
    int l1_per_point;              /* layer 1 of a grouped level with features once per point (OFF: never) */
    int l1_coords;                 /* layer 1 of a feature-less level on the vector units (OFF: never) */
    int force_stream;              /* ON: weights streamed through LDS even where they would stay resident */
    int max_ns;                    /* cap on 32-column output tiles per wave: 0 (= 4), 1, 2, 4 */
    int nt;                        /* non-temporal stores: AUTO by size (outputs >= 128 MB), OFF never, ON always */
    int fuse_wgrad;                /* a layer's data gradient inside its weight-gradient pass (one pass over the layer's activations
                                      instead of two; AUTO: from 0.5 M rows where the layer's tiles fit one slab; ON: wherever
                                      they fit, the z-free pooled top layer included), OFF: never */
    int wgrad_two_per_cu;          /* two weight-gradient workgroups per CU where registers and LDS allow (AUTO / ON), OFF: one */
    int side_stream;               /* backward: the weight-gradient launches on a helper stream beside the data-gradient chain
                                      (fork / join by events on the caller's stream): ON only -- measured slower on all but the
                                      group_all level, so AUTO = OFF */
    int pair_launch;               /* backward: a layer's data-gradient GEMM and its weight-gradient pass as two ranges of workgroups
                                      of ONE launch (independent passes over the same tensors; AUTO: below 0.5 M rows, where each is a
                                      latency-bound launch over part of the chip; ON: wherever the shape pair has a kernel), OFF: never */
    int fold_finalize;             /* the per-channel finalisation of a pass's sums (batch moments -> coefficients, running statistics;
                                      backward: grad_gamma, grad_beta, dz coefficients) by the LAST workgroup of the pass that produced
                                      them (ticket; write-through partial rows) instead of a launch of its own: ON only -- measured
                                      1-8 us SLOWER per pass than the 5 us launch it saves (the XCDs' L2s are not coherent: the hand-off
                                      is four trips to memory), so AUTO = OFF. The sums are added in a fixed order either way; the two
                                      orders differ, so results agree to the fp64 rounding of the sums, not bit for bit */
} pn2_train_opts;
long long pn2_mlp_train_ws_bytes(long long rows, int nlayers, const int *widths /* cin_1, cout_1 .. cout_L */,
                                 int pool_rows, int backward,
                                 const int *group_dims /* grouped input: {b, n, m, nsample, cfeat, idx != NULL}; else NULL */);
/* 1: layer 1 of this grouped level is evaluated once per POINT -- z_1 = (points W1f)[idx] + (xyz - c) W1x + b, the
 * feature term being a GEMM over the b n points instead of the b m nsample rows (csrc/train_mlp.hip, tl_l1_forward_kernel)
 * -- and backward then produces the gradient of `points` itself (grad_points) instead of the per-row grad_feat_rows. */
int pn2_mlp_train_layer1_per_point(int nlayers, const int *widths, const int *group_dims);
/* 1: layers[L-1].z (the top layer's pre-norm tensor) is written by forward and read by backward; 0: it is neither --
 * pooled stacks of >= 2 layers on large levels run the passes that would read z_L on the layer's INPUT instead
 * (z_L = h W + b; csrc/train_mlp.hip, tl_top_mats_kernel), and layers[L-1].z may be NULL. */
int pn2_mlp_train_top_stored(long long rows, int nlayers, const int *widths, int pool_rows);
int pn2_mlp_train_forward(long long rows, int nlayers, const pn2_bn_layer *layers, const pn2_group_src *group,
                          const float *x, int pool_rows, float *out, int *argsel, float *zsel, void *ws, void *stream);
/* grad_out: shape of out. grad_x: (rows, cin_1) or NULL (plain input). Grouped input: grad_feat_rows (rows, cfeat) -- the
 * gradient of the grouped FEATURE rows, to be scattered with pn2_group_point_grad_seg -- or, when
 * pn2_mlp_train_layer1_per_point() says so, grad_points (b, n, cfeat), the gradient of `points` itself (the scatter runs
 * inside, `reproducible` selects its sorted-segment mode); NULL: not wanted. The bias gradient of a layer under batch
 * normalisation is identically zero and is not produced. */
int pn2_mlp_train_backward(long long rows, int nlayers, const pn2_bn_layer *layers, const pn2_group_src *group,
                           const float *x, int pool_rows, const float *out, const int *argsel, const float *zsel,
                           const float *grad_out, float *grad_x, float *grad_feat_rows, float *grad_points, int reproducible,
                           void *ws, void *stream);
/* the same five entry points with the organisation overrides (the plain ones pass NULL) */
long long pn2_mlp_train_ws_bytes_ex(long long rows, int nlayers, const int *widths, int pool_rows, int backward,
                                    const int *group_dims, const pn2_train_opts *opts);
int pn2_mlp_train_layer1_per_point_ex(int nlayers, const int *widths, const int *group_dims, const pn2_train_opts *opts);
int pn2_mlp_train_top_stored_ex(long long rows, int nlayers, const int *widths, int pool_rows, const pn2_train_opts *opts);
int pn2_mlp_train_forward_ex(long long rows, int nlayers, const pn2_bn_layer *layers, const pn2_group_src *group,
                             const float *x, int pool_rows, float *out, int *argsel, float *zsel, void *ws,
                             const pn2_train_opts *opts, void *stream);
int pn2_mlp_train_backward_ex(long long rows, int nlayers, const pn2_bn_layer *layers, const pn2_group_src *group,
                              const float *x, int pool_rows, const float *out, const int *argsel, const float *zsel,
                              const float *grad_out, float *grad_x, float *grad_feat_rows, float *grad_points,
                              int reproducible, void *ws, const pn2_train_opts *opts, void *stream);

/* The input rows of a feature-propagation level's layer stack in ONE launch (pointnet_fp_module, utils/pointnet_util.py:211-219):
 * inverse-distance weights from three_nn's `dist`, three_interpolate of points2 (b,m,c2), concatenation with the skip features
 * points1 (b,n,c1; NULL with c1 = 0), zero columns up to `pitch` (a multiple of 4 >= c2 + c1: the training entry points read
 * rows 16 bytes at a time) -> out (b,n,pitch); weight (b,n,3) receives the weights (needed by the gradient) unless NULL.
 * The operators' formulas in IEEE fp32 (agreement with the composition of torch elementwise kernels + pn2_three_interpolate:
 * 2e-6 of the tensor's scale, tests/test_fp_train_gpu.py). The gradient call splits the stack's input
 * gradient grad_x (b,n,pitch) into grad_points1 (b,n,c1; may be NULL) and the interpolated part (scratch, (b,n,c2) floats),
 * which pn2_three_interpolate_grad_seg scatters onto grad_points2 (b,m,c2) (ws_seg: pn2_seg_grad_ws_bytes(b, m, 3 n)). */
int pn2_fp_interp_concat(int b, int n, int m, int c2, int c1, int pitch, const float *points2, const float *points1,
                         const int *idx, const float *dist, float *out, float *weight, void *stream);
int pn2_fp_interp_concat_grad(int b, int n, int m, int c2, int c1, int pitch, const float *grad_x, const int *idx,
                              const float *weight, float *grad_points2, float *grad_points1, float *scratch, void *ws_seg,
                              int deterministic, void *stream);

/* diagnostics: byte offsets inside the BACKWARD workspace of the two dy buffers ((rows, max width) each; after a
 * backward of L layers they hold dy_{L-1}, dy_{L-2}, ... alternately, starting with gb when pooled) and of the
 * per-layer (2, cout) fp64 sums / (3, cout) fp32 coefficients */
int pn2_mlp_train_ws_layout(long long rows, int nlayers, const int *widths, int pool_rows, long long *ga, long long *gb,
                            long long *stats, long long *coef);

/* ---- host helpers ------------------------------------------------------- */

/* The exact fp32 threshold s* with  max(sqrtf(s),1e-20f) < radius  <=>  s < s*
 * (host computation, monotonicity of correctly-rounded sqrtf). Exposed for tests. */
float pn2_ball_threshold(float radius);

#ifdef __cplusplus
}
#endif
#endif /* PN2OPS_H */
