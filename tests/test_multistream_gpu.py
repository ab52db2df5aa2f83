"""Multi-stream use of the library as a TESTED contract.

include/pn2ops.h promises re-entrancy: the library holds no mode and no memory, every entry point only enqueues on the
caller's stream (the reference's launchers are stateless too, tf_grouping_g.cu:125-141; TF runs op instances concurrently
on inter-op threads, SURVEY.md 8b "Threading / streams"). Round 4 saw ONE unexplained mismatch in an experiment that ran
the geometry kernels on a side stream (profiles/r04/geometry_prefetch_experiment.txt) and nothing in tests/ ran product
kernels on two streams at once. These tests do, and every output must be BIT-identical to the single-stream result (and
the geometry to the CPU oracle):

  * four streams, each with its own shapes, all in flight at once: the overlapped launch (csrc/sa_fused.hip) at two
    different (b, m) -- incl. b not a multiple of 8, where a cloud's consumers sit on other XCDs than its producer --,
    the two-launch path, the fused MLP + max-pool, three_nn + the fused feature-propagation kernel, group_point /
    three_interpolate; fresh output tensors every iteration (the caching allocator recycles them across iterations);
  * the Python layer's granule-workspace cache pushed past its 64 entries while launches are in flight
    (tf_grouping._GRANULES, one workspace per (device, stream, b * m));
  * the producer / consumer pattern of the removed experiment: geometry on a side stream, the MLP kernel on the main
    stream behind an event, tensors handed across with record_stream.
Mismatches are counted ON THE DEVICE (no host synchronisation inside the loops: the point is that the streams overlap)."""
import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu

ITERS = 200


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _layers(rng, dims):
    return [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
             (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(len(dims) - 1)]


class _Worker:
    """The jobs of ONE stream: inputs, single-stream reference outputs, a device-side mismatch counter per job."""

    def __init__(self, cuda, s):
        import pointnet2_amd as P
        from pointnet2_amd import sa_mlp
        self.P, self.sa_mlp = P, sa_mlp
        rng = np.random.default_rng(100 + s)
        # (b, n, m, radius, nsample): every stream its own shapes; b = 3, 5 are not multiples of 8 (consumers of a cloud on
        # other XCDs than its producer), 32 x 1024 -> 512 runs the four-wave producers, n = 8192 the sweep consumers
        self.shape_a = [(8, 2048, 256, 0.2, 32), (3, 1024, 128, 0.25, 16), (16, 1024, 512, 0.2, 32), (5, 4096, 256, 0.15, 64)][s]
        self.shape_b = [(4, 4096, 512, 0.2, 32), (32, 1024, 512, 0.2, 32), (2, 8192, 128, 0.1, 32), (8, 512, 128, 0.4, 64)][s]
        gen = [S.sphere_clouds, S.uniform_clouds, S.duplicated_clouds, S.sphere_clouds][s]
        self.xyz_a_np = gen(self.shape_a[0], self.shape_a[1], 10 + s)
        self.xyz_b_np = S.sphere_clouds(self.shape_b[0], self.shape_b[1], 20 + s)
        self.xyz_a, self.xyz_b = _dev(self.xyz_a_np, cuda), _dev(self.xyz_b_np, cuda)
        b, n, m, _, ns = self.shape_a
        cf = [0, 6, 64, 3][s]
        self.feats = _dev(rng.standard_normal((b, n, cf)).astype(np.float32), cuda) if cf else None
        widths = [(64, 64, 128), (32, 32, 64), (64, 64, 128), (64, 96, 128)][s]
        self.packed = sa_mlp.PackedMLP3(_layers(rng, (3 + cf,) + widths), cuda, ns)
        # feature propagation: unknown points = cloud a, known points = its first m_known points
        self.mk = [128, 1, 64, 16][s]
        self.c2, self.c1 = [(128, 0), (64, 32), (256, 64), (32, 6)][s]
        fpw = [[128, 128, 128], [64, 64], [256, 128], [32, 48]][s]
        self.known = self.xyz_a[:, :self.mk].contiguous()
        self.p2 = _dev(rng.standard_normal((b, self.mk, self.c2)).astype(np.float32), cuda)
        self.p1 = _dev(rng.standard_normal((b, n, self.c1)).astype(np.float32), cuda) if self.c1 else None
        kind = sa_mlp.fp_kind(b * n, self.c2, self.c1, fpw)
        assert kind is not None
        self.fp_packed = sa_mlp.PackedFPMLP(_layers(rng, [self.c2 + self.c1] + fpw), self.c2, self.c1, cuda, kind)
        self.gfeat = _dev(rng.standard_normal((b, n, 16)).astype(np.float32), cuda)
        self.jobs = [self.job_overlap_a, self.job_overlap_b, self.job_two_launch, self.job_mlp, self.job_fp, self.job_rows]
        self.ref = None
        self.bad = None

    # every job returns a tuple of freshly allocated output tensors
    def job_overlap_a(self):
        b, n, m, r, ns = self.shape_a
        return self.P.sample_and_group_xyz(m, r, ns, self.xyz_a, True)

    def job_overlap_b(self):
        b, n, m, r, ns = self.shape_b
        return self.P.sample_and_group_xyz(m, r, ns, self.xyz_b, True)

    def job_two_launch(self):
        b, n, m, r, ns = self.shape_a
        fps, new_xyz = self.P.farthest_point_sample_gather(m, self.xyz_a)
        idx, cnt, grouped = self.P.query_ball_group_xyz(r, ns, self.xyz_a, new_xyz, True)
        return fps, new_xyz, idx, cnt, grouped

    def job_mlp(self):
        _, new_xyz, idx, _, _ = self.ref[0]
        return (self.sa_mlp.sa_mlp_maxpool(self.xyz_a, new_xyz, self.feats, idx, self.packed),)

    def job_fp(self):
        dist, idx = self.P.three_nn(self.xyz_a, self.known)
        return dist, idx, self.sa_mlp.fp_mlp(self.p2, self.p1, idx, dist, self.fp_packed)

    def job_rows(self):
        _, _, idx, _, _ = self.ref[0]
        dist, nn = self.ref[4][0], self.ref[4][1]
        w = torch.full_like(dist, 1.0 / 3.0)
        return self.P.group_point(self.gfeat, idx), self.P.three_interpolate(self.p2, nn, w)

    def make_reference(self, cuda):
        self.ref = []
        for job in self.jobs:
            self.ref.append(tuple(t.clone() for t in job()))
        torch.cuda.synchronize()
        self.bad = torch.zeros((len(self.jobs),), dtype=torch.int64, device=cuda)

    def run_and_compare(self, j):
        outs = self.jobs[j]()
        n_bad = None
        for o, r in zip(outs, self.ref[j]):
            d = (o != r).sum()
            n_bad = d if n_bad is None else n_bad + d
        self.bad[j] += n_bad


def _check_geometry_against_oracle(oracle, w):
    """The single-stream references themselves: FPS / gather / ball query / group against the CPU oracle."""
    for ref, xyz, shape in ((w.ref[0], w.xyz_a_np, w.shape_a), (w.ref[1], w.xyz_b_np, w.shape_b), (w.ref[2], w.xyz_a_np, w.shape_a)):
        b, n, m, r, ns = shape
        fps = oracle.farthest_point_sample(m, xyz)
        new_xyz = oracle.gather_point(xyz, fps)
        idx, cnt = oracle.query_ball_point(r, ns, xyz, new_xyz)
        gx = oracle.group_point(xyz, idx) - new_xyz[:, :, None, :]
        got = [t.cpu().numpy() for t in ref]
        assert np.array_equal(got[0], fps) and np.array_equal(got[1], new_xyz)
        assert np.array_equal(got[2], idx) and np.array_equal(got[3], cnt) and np.array_equal(got[4], gx)


def test_four_streams_every_kernel_family_bit_identical(cuda, oracle):
    from pointnet2_amd import tf_grouping as G
    workers = [_Worker(cuda, s) for s in range(4)]
    for w in workers:
        w.make_reference(cuda)
    for w in workers:
        _check_geometry_against_oracle(oracle, w)
    streams = [torch.cuda.Stream(device=cuda) for _ in workers]
    torch.cuda.synchronize()
    for it in range(ITERS):
        for j in range(len(workers[0].jobs)):
            # job j of every stream is enqueued back to back, with a different rotation every iteration, so the kernels
            # that share the device change from iteration to iteration
            for k in range(len(workers)):
                s = (k + it) % len(workers)
                with torch.cuda.stream(streams[s]):
                    workers[s].run_and_compare((j + s) % len(workers[s].jobs))
    torch.cuda.synchronize()
    G.check_overlapped_launches(cuda)                               # no consumer ever gave up on its producer
    names = ["overlapped launch A", "overlapped launch B", "two-launch path", "sa_mlp_maxpool", "three_nn + fp_mlp", "group_point + three_interpolate"]
    report = {"stream %d: %s" % (s, names[j]): int(w.bad[j]) for s, w in enumerate(workers) for j in range(len(names)) if int(w.bad[j])}
    assert not report, "elements that differ from the single-stream result over %d iterations: %r" % (ITERS, report)


def test_granule_cache_overflow_while_launches_are_in_flight(cuda, oracle):
    """More than 64 (device, stream, b * m) workspaces: tf_grouping drops the WHOLE cache while earlier launches on the other
    streams still use their buffers (stream-ordered reuse by the caching allocator makes that safe; this test is the proof).
    Sample j of an FPS chain and its ball query do not depend on npoint, so one npoint = 96 reference serves every m."""
    import pointnet2_amd as P
    from pointnet2_amd import tf_grouping as G
    b, n, r, ns = 6, 1024, 0.25, 16
    ms = list(range(60, 96, 2))                                     # 18 sizes x 4 streams = 72 keys
    xyz_np = [S.sphere_clouds(b, n, 40 + s) for s in range(4)]
    xyz = [_dev(a, cuda) for a in xyz_np]
    ref = []
    for s in range(4):
        out = P.sample_and_group_xyz(96, r, ns, xyz[s], True)
        fps = oracle.farthest_point_sample(96, xyz_np[s])
        idx, cnt = oracle.query_ball_point(r, ns, xyz_np[s], oracle.gather_point(xyz_np[s], fps))
        assert np.array_equal(out[0].cpu().numpy(), fps) and np.array_equal(out[2].cpu().numpy(), idx)
        assert np.array_equal(out[3].cpu().numpy(), cnt)
        ref.append(tuple(t.clone() for t in out))
    streams = [torch.cuda.Stream(device=cuda) for _ in range(4)]
    bad = torch.zeros((4,), dtype=torch.int64, device=cuda)
    torch.cuda.synchronize()
    G._GRANULES.clear()
    dropped = 0
    for rep in range(3):
        for m in ms:
            for s in range(4):
                with torch.cuda.stream(streams[s]):
                    before = len(G._GRANULES)
                    out = P.sample_and_group_xyz(m, r, ns, xyz[s], True)
                    dropped += 1 if len(G._GRANULES) < before else 0
                    d = None
                    for o, rf in zip(out, ref[s]):
                        e = (o != rf[:, :m]).sum()
                        d = e if d is None else d + e
                    bad[s] += d
    torch.cuda.synchronize()
    G.check_overlapped_launches(cuda)
    assert dropped >= 1, "the cache never overflowed: the test does not test what it says"
    assert bad.tolist() == [0, 0, 0, 0]


def test_geometry_on_a_side_stream_consumed_on_the_main_stream(cuda):
    """The removed experiment's pattern, done by the book: the level's geometry (overlapped launch, then a second level in
    the two-launch path on ITS output) on a side stream, an event, the fused MLP kernels on the main stream; every tensor
    that crosses streams is record_stream'ed for the allocator. 100 iterations against the single-stream results."""
    import pointnet2_amd as P
    from pointnet2_amd import sa_mlp
    rng = np.random.default_rng(7)
    b, n = 16, 2048
    xyz = _dev(S.sphere_clouds(b, n, 3), cuda)
    pk1 = sa_mlp.PackedMLP3(_layers(rng, (3, 64, 64, 128)), cuda, 32)
    pk2 = sa_mlp.PackedMLP3(_layers(rng, (3 + 128, 128, 128, 256)), cuda, 64)

    def geometry():
        _, nx1, idx1, _, _ = P.sample_and_group_xyz(512, 0.2, 32, xyz, True)
        _, nx2 = P.farthest_point_sample_gather(128, nx1)
        idx2, _, _ = P.query_ball_group_xyz(0.4, 64, nx1, nx2, True)
        return nx1, idx1, nx2, idx2

    def mlps(nx1, idx1, nx2, idx2):
        f1 = sa_mlp.sa_mlp_maxpool(xyz, nx1, None, idx1, pk1)
        return f1, sa_mlp.sa_mlp_maxpool(nx1, nx2, f1, idx2, pk2)

    g_ref = geometry()
    f_ref = mlps(*g_ref)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=cuda)
    main = torch.cuda.current_stream(cuda)
    bad = torch.zeros((), dtype=torch.int64, device=cuda)
    for it in range(100):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            g = geometry()
            ev = torch.cuda.Event()
            ev.record(side)
        main.wait_event(ev)
        for t in g:
            t.record_stream(main)
        f = mlps(*g)
        for o, rf in zip(g + f, g_ref + f_ref):
            bad += (o != rf).sum()
        del g, f
    torch.cuda.synchronize()
    assert int(bad) == 0


def test_overlapped_launches_on_four_streams_with_inputs_that_change_every_iteration(cuda):
    """the four-stream test above re-runs ONE input per stream, where a granule taken from the launch
    before is invisible by construction. Here every stream rotates three clouds, so launch i's consumers would show launch
    i - 1's samples at once: 4 streams x 300 overlapped launches (host generations on the stream's workspace), each compared on
    the device with the single-stream reference of the cloud it was given."""
    import pointnet2_amd as P
    shapes = [(8, 2048, 256, 0.2, 32), (3, 1024, 128, 0.25, 16), (16, 1024, 512, 0.2, 32), (5, 4096, 256, 0.15, 64)]
    gens = [S.sphere_clouds, S.uniform_clouds, S.sphere_clouds, S.uniform_clouds]
    clouds, refs = [], []
    for s, (b, n, m, r, ns) in enumerate(shapes):
        cs = [_dev(gens[s](b, n, 40 + 3 * s + k), cuda) for k in range(3)]
        clouds.append(cs)
        refs.append([tuple(t.clone() for t in P.sample_and_group_xyz(m, r, ns, c, True)) for c in cs])
        assert not torch.equal(refs[s][0][0], refs[s][1][0])             # the clouds really differ in their samples
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=cuda, priority=-(s & 1)) for s in range(4)]
    bad = torch.zeros((4,), dtype=torch.int64, device=cuda)
    for it in range(300):
        for s, (b, n, m, r, ns) in enumerate(shapes):
            k = (it + s) % 3
            with torch.cuda.stream(streams[s]):
                out = P.sample_and_group_xyz(m, r, ns, clouds[s][k], True)
                d = None
                for o, rf in zip(out, refs[s][k]):
                    e = (o != rf).sum()
                    d = e if d is None else d + e
                bad[s] += d
    torch.cuda.synchronize()
    P.tf_grouping.check_overlapped_launches(cuda)
    assert bad.tolist() == [0, 0, 0, 0]
