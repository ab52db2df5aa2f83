// levels.hip -- ONE C entry point per set-abstraction level and per feature-propagation level of the inference path
// (missing 3): an eager SA level was two operator calls (sample-and-group, fused MLP + max-pool) and an
// FP level two (three_nn, fused interpolation + MLP), 7-10 us of host time each -- what the small levels of the
// segmentation networks cost. These entry points only compose the existing launches on the caller's stream; every
// buffer is the caller's.
// Reference: utils/pointnet_util.py:87-154 (pointnet_sa_module = sample_and_group :22-56 + conv stack + reduce_max),
// :199-229 (pointnet_fp_module = three_nn + three_interpolate + concat + conv stack).
#include "pn2_device.h"

extern "C" int pn2_sa_level(int b, int n, int m, float radius, int nsample, int cfeat, const float *xyz, const float *points,
                            void *ws_sample, unsigned generation, float *fps_temp, int c1, int c2, int c3, const float *wpacked,
                            const float *bpacked, int *fps_idx, float *new_xyz, int *idx, int *pts_cnt, float *grouped_xyz,
                            float *out, void *ws_mlp, void *stream)
{
    // FPS + gather + ball query + grouping of xyz: the overlapped launch, or -- outside its envelope / on a device that
    // cannot hold every producer at once -- the two-launch path (bit-identical results)
    // ws_sample == NULL: the caller does not want the overlapped launch (tf_grouping.set_overlapped_launch(False), the switch
    // OverlappedLaunchError points at) -- straight to the two-launch path
    int rc = !ws_sample ? PN2_E_TOO_LARGE
             : generation ? pn2_sample_and_group_xyz_gen(b, n, m, radius, nsample, xyz, ws_sample, generation, fps_idx, new_xyz, idx,
                                                         pts_cnt, grouped_xyz, 1, stream)
                          : pn2_sample_and_group_xyz(b, n, m, radius, nsample, xyz, ws_sample, fps_idx, new_xyz, idx, pts_cnt,
                                                     grouped_xyz, 1, stream);
    if (rc == PN2_E_TOO_LARGE) {
        rc = pn2_farthest_point_sample_gather(b, n, m, xyz, fps_temp, fps_idx, new_xyz, stream);
        if (rc) return rc;
        rc = pn2_query_ball_group_xyz(b, n, m, radius, nsample, xyz, new_xyz, 1, idx, pts_cnt, grouped_xyz, stream);
    }
    if (rc) return rc;
    // [grouped xyz - centroid, grouped features] -> three folded layers -> max over nsample, nothing materialised
    return pn2_sa_mlp3_maxpool(b, n, m, nsample, cfeat, xyz, new_xyz, points, idx, c1, c2, c3, wpacked, bpacked, out, ws_mlp, stream);
}

extern "C" int pn2_sa_level_ordered(int b, int n, int m, float radius, int nsample, int cfeat, const float *xyz, const float *points,
                                    void *ws_ordered, int c1, int c2, int c3, const float *wpacked, const float *bpacked,
                                    int *fps_idx, float *new_xyz, int *idx, int *pts_cnt, float *grouped_xyz, float *out,
                                    void *ws_mlp, void *stream)
{
    // xyz is (believed to be) the previous level's samples in the order they were picked (pointnet2_sem_seg.py:28-31): the
    // dependent rounds are replaced by a parallel check, clouds that fail it run the chain (fps.hip)
    if (int rc = pn2_farthest_point_sample_ordered(b, n, m, xyz, fps_idx, new_xyz, ws_ordered, stream)) return rc;
    if (int rc = pn2_query_ball_group_xyz(b, n, m, radius, nsample, xyz, new_xyz, 1, idx, pts_cnt, grouped_xyz, stream)) return rc;
    return pn2_sa_mlp3_maxpool(b, n, m, nsample, cfeat, xyz, new_xyz, points, idx, c1, c2, c3, wpacked, bpacked, out, ws_mlp, stream);
}

extern "C" int pn2_fp_level(int b, int n, int m, int c2, int c1, const float *xyz1, const float *xyz2, const float *points2,
                            const float *points1, int nlayers, const int *widths, int kind, const float *wpacked,
                            const float *bpacked, float *dist, int *idx, float *out, void *ws, void *stream)
{
    if (int rc = pn2_three_nn(b, n, m, xyz1, xyz2, dist, idx, stream)) return rc;
    return pn2_fp_mlp(b, n, m, c2, c1, points2, points1, idx, dist, nlayers, widths, kind, wpacked, bpacked, out, ws, stream);
}
