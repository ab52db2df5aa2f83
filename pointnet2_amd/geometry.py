"""The geometry of a network's levels AHEAD of its layer stacks, on a HIP stream of its own.

Every level of the reference networks samples, groups and interpolates from coordinates only
(models/pointnet2_sem_seg.py:28-37, pointnet2_cls_ssg.py:27-29, pointnet2_part_seg.py:24-33): l{k}_xyz depends on
l{k-1}_xyz, never on a feature, and the INDEX outputs carry no gradient (farthest_point_sample, query_ball_point and
three_nn are registered NotDifferentiable / index outputs: tf_sampling.py:57, tf_grouping.py:29, tf_interpolate.py:20).
The one gradient that does pass through a level's geometry is GatherPoint's (tf_sampling.py:43-47: d new_xyz / d xyz): a
geometry is computed from xyz.detach(), so a module whose xyz requires a gradient rebuilds new_xyz from the geometry's
sample indices with the differentiable gather_point (SAGeometry.new_xyz_for) -- same values, the centroid term kept.
The farthest-point chains are also the one part of a forward that cannot use the machine: b workgroups on 256 CUs for
hundreds of microseconds (132 us of cls_ssg's 486, 435 us of sem_seg's 897), while the layer stacks that follow fill every CU.

So the two halves run on two streams. GeometryAhead.submit(xyz) enqueues, on its own stream, every SA level's
sample_and_group_xyz (or the multi-radius launches of an MSG level, or FPS + kNN) and every FP level's three_nn, one event per
level; the modules take `geometry=` and run only their layer stack after waiting for their level's event. Within one batch
this overlaps level k's stack with the geometry of levels k+1..; submitting batch i+1 before running batch i hides the whole
geometry, level-1 chain included, under batch i's stacks (what a serving loop or a training loop with a prefetching loader
does) -- scripts/model_forward_bench.py reports both.

The results are the results of the plain forward, bit for bit (tests/test_geometry_ahead_gpu.py): the same kernels compute
them, only their stream differs (inside captured graphs the overlapped launch numbers itself: tf_grouping._capture_workspace). Multi-stream use of the library is a tested contract since round 5
(tests/test_multistream_gpu.py; the v_pk_add_f32 hazard beside MFMA kernels is documented in csrc/pn2_device.h).
"""
import torch

from .tf_interpolate import three_nn


class _Ready:
    """Tensors produced on another stream + the event that says they are complete."""

    __slots__ = ("event", "stream", "_tensors")

    def _init(self, tensors):
        self._tensors = [t for t in tensors if isinstance(t, torch.Tensor)]
        self.event, self.stream = None, None

    def mark(self, stream):
        """Called by the producer, on its stream, after the launches were enqueued."""
        self.stream = stream
        self.event = stream.record_event()
        return self

    def wait(self):
        """Make the CURRENT stream wait for the level's launches. No Tensor.record_stream here (see GeometryAhead.submit): the
        tensors belong to the producer stream's pool, and the producer's next allocations are ordered behind this stream's
        work by the wait at the top of every submit()."""
        if self.event is not None:
            cur = torch.cuda.current_stream(self._tensors[0].device)
            if cur != self.stream:
                cur.wait_event(self.event)
        return self


class SAGeometry(_Ready):
    """One set-abstraction level: new_xyz (b, m, 3) and idx (b, m, nsample) i32 -- a list of idx, one per radius, for an
    MSG level (pointnet_util.py:156-197). fps_idx (b, m) i32: the samples' indices, kept so that a consumer whose xyz needs a
    gradient can rebuild new_xyz = gather_point(xyz, fps_idx) differentiably (the reference registers a gradient for
    GatherPoint, tf_sampling.py:43-47); None for a geometry that was copied without it."""

    __slots__ = ("new_xyz", "idx", "fps_idx")

    def __init__(self, new_xyz, idx, fps_idx=None):
        self.new_xyz, self.idx, self.fps_idx = new_xyz, idx, fps_idx
        self._init([new_xyz] + (list(idx) if isinstance(idx, (list, tuple)) else [idx]) + ([fps_idx] if fps_idx is not None else []))

    def new_xyz_for(self, xyz):
        """new_xyz as the consumer must use it: the precomputed tensor, or -- when xyz needs a gradient -- the same values
        gathered from xyz by the differentiable operator, so that the centroid term of d / d xyz is not lost (the geometry
        itself is computed from xyz.detach())."""
        if not (torch.is_grad_enabled() and xyz.requires_grad):
            return self.new_xyz
        if self.fps_idx is None:
            raise ValueError("this geometry carries no sample indices, and xyz requires a gradient: the centroids cannot be "
                             "rebuilt differentiably -- compute the geometry with GeometryAhead / module.geometry(), or detach xyz")
        from .tf_sampling import gather_point
        return gather_point(xyz, self.fps_idx)


class FPGeometry(_Ready):
    """One feature-propagation level: three_nn's dist / idx (b, n, 3) (pointnet_util.py:211)."""

    __slots__ = ("dist", "idx")

    def __init__(self, dist, idx):
        self.dist, self.idx = dist, idx
        self._init([dist, idx])


class NetworkGeometry:
    """sa[k]: SAGeometry of the k-th SA module (None for a group_all level, which has no geometry);
    fp[k]: FPGeometry of the k-th (xyz1 level, xyz2 level) pair given to GeometryAhead."""

    __slots__ = ("sa", "fp", "_src")

    def __init__(self, sa, fp):
        self.sa, self.fp = sa, fp
        self._src = None                 # the coordinates the launches read (kept alive with the result, see GeometryAhead.submit)

    def tensors(self):
        """Every tensor of every level, in a fixed order."""
        return [t for g in list(self.sa) + list(self.fp) if g is not None for t in g._tensors]

    def static_copy(self):
        """The same geometry in tensors of its own, without events."""
        sa = [None if g is None else SAGeometry(g.new_xyz.clone(), [i.clone() for i in g.idx] if isinstance(g.idx, (list, tuple))
                                                else g.idx.clone(), None if g.fps_idx is None else g.fps_idx.clone()) for g in self.sa]
        fp = [None if g is None else FPGeometry(g.dist.clone(), g.idx.clone()) for g in self.fp]
        return NetworkGeometry(sa, fp)

    def copy_(self, other):
        """Refill this (static) geometry from another one of the same shapes, on the current stream."""
        for dst, src in zip(self.tensors(), other.tensors()):
            dst.copy_(src)
        return self


class GeometryAhead:
    """sa_modules: the network's SA modules in order (PointnetSAModule / PointnetSAModuleMSG; level 0 is the input cloud,
    level k the output of the k-th module). fp_pairs: (i, j) per FP module = three_nn(level i's xyz, level j's xyz), e.g.
    sem_seg's fp1..fp4 (pointnet2_sem_seg.py:34-37) are [(3, 4), (2, 3), (1, 2), (0, 1)].

        ahead = GeometryAhead([net.sa1, net.sa2, net.sa3], [(2, 3), (1, 2), (0, 1)])
        g = ahead.submit(xyz)               # returns at once; the launches go to ahead.stream
        out = net(xyz, geometry=g)          # stacks on the current stream, each waiting for its level

    high_priority: ask for a high-priority queue for the geometry stream (the chains are short kernels whose latency is the
    point; the layer stacks would otherwise queue in front of them).
    """

    def __init__(self, sa_modules, fp_pairs=(), device=None, high_priority=True):
        self.sa_modules = list(sa_modules)
        self.fp_pairs = [tuple(p) for p in fp_pairs]
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.device = dev
        self.stream = torch.cuda.Stream(device=dev, priority=-1 if high_priority else 0)
        self._in_flight = []             # (event after a submit's last launch, its input): inputs stay alive while launches read them

    def compute(self, xyz, mark=None):
        """Every level's geometry on the CURRENT stream (no stream switch, no events unless `mark` is a stream to record them
        on): what submit() enqueues on self.stream; also what a caller captures into a HIP graph of its own."""
        xyz = xyz.detach()
        sa, fp, level_xyz = [], [], [xyz]
        with torch.no_grad():
            for mod in self.sa_modules:
                src = level_xyz[-1]
                g = None if src is None else mod.geometry(src)
                sa.append(g if g is None or mark is None else g.mark(mark))
                level_xyz.append(None if g is None else g.new_xyz)
            for i, j in self.fp_pairs:
                if level_xyz[i] is None or level_xyz[j] is None:      # the known points of a group_all level: one point at the origin
                    fp.append(None)
                    continue
                dist, idx = three_nn(level_xyz[i], level_xyz[j])
                g = FPGeometry(dist, idx)
                fp.append(g if mark is None else g.mark(mark))
        return NetworkGeometry(sa, fp)

    def submit(self, xyz):
        """xyz (b, n, 3) f32 on the device, produced on the current stream -> NetworkGeometry (launches enqueued on
        self.stream, one event per level, nothing waited for). Consume the result on the stream that is current HERE.

        Lifetimes without Tensor.record_stream: the input is kept referenced until its launches are done, and the results --
        blocks of self.stream's pool -- can only be reused by a later submit, whose launches wait (first line below) for
        everything the consumer stream held when it was called."""
        if not (isinstance(xyz, torch.Tensor) and xyz.is_cuda and xyz.device == self.device and xyz.dim() == 3 and xyz.shape[2] == 3):
            raise ValueError("GeometryAhead.submit expects (batch_size, num_points, 3) coordinates on %s" % (self.device,))
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_event(cur.record_event())
        self._in_flight = [(ev, t) for ev, t in self._in_flight if not ev.query()]
        with torch.cuda.stream(self.stream):
            g = self.compute(xyz, mark=self.stream)
            g._src = xyz
            self._in_flight.append((self.stream.record_event(), xyz))
        return g


class PipelinedInference:
    """A serving loop with the geometry one batch ahead and no host time in the loop: per input slot (two of them) one HIP
    graph of the network's geometry, replayed on the geometry stream, and one of its layer stacks reading that graph's output
    tensors, replayed on the caller's stream; two events per batch order them. While the stacks of batch i fill the CUs, the
    farthest-point chains of batch i + 1 run beside them (eval forwards per batch, same box: cls_ssg 0.53 -> 0.43 ms,
    cls_msg 1.42 -> 1.23, part_seg 0.68 -> 0.53, sem_seg 1.05 -> 0.68, 0.58 with geometry_streams=2;
    profiles/r05/model_forward.txt). Every output is the plain forward's bit for bit: 600-batch soaks with three inputs in
    rotation per network and setting in scripts/model_forward_bench.py, a 1,200-batch soak in the test suite.

    The soak is what found round 5's one real defect (profiles/r05/geometry_ahead.txt): inside a CAPTURED graph the
    overlapped launch -- workspace cleared by a memset node, constant tag -- accepted sample granules that were not the
    replay's own from some replay on (duplicated rows in new_xyz). Round 6 found the cause (profiles/r06/stale_granules.md: a
    replayed memset NODE of this runtime fills with stale launch arguments instead of zeros) and a form that depends on no
    clear: inside graphs the launch numbers itself (tf_grouping.sample_and_group_xyz, PN2_GENERATION_DEVICE), on a
    workspace per captured call site that the eager warm-up below stocks.

        ahead = GeometryAhead([net.sa1, net.sa2, net.sa3], [(2, 3), (1, 2), (0, 1)])   # the network's SA modules / FP pairs
        pipe = PipelinedInference(net, ahead, example_batch)              # captures; shapes are fixed from here on
        for x, ready in loader:                                           # x filled by the loader's stream, `ready` its event
            y = pipe.push(x, ready)                                       # returns at once; y is valid in stream order
            consume(y)                                                    # ... on the current stream

    model(x, geometry) is the network (modules called with `geometry=`), in eval mode under no_grad; coords(x) -> the (b, n, 3)
    coordinates the geometry is computed from (default: x itself). The output is a static buffer of its slot: the push
    `slots` (2, or 2 * geometry_streams) calls after the one that returned it overwrites it -- consume or clone it before.

    What was measured on the way (profiles/r05/geometry_ahead.txt): ONE graph with a forked branch does not overlap anything,
    hipGraphLaunch ran the two branches one after the other (0.72 ms on cls_ssg against 0.49 plain). The stacks on a stream
    created for them land, by the round-robin of streams over hardware queues, in the geometry stream's queue every other time
    (0.46 / 0.84-0.95 ms in alternation, whatever GPU_MAX_HW_QUEUES says); the caller's stream beside a HIGH-PRIORITY geometry
    stream (its own queue class) is 0.45-0.46 every time, hence this organisation. Three conservative choices date from the
    hunt for the defect above and were kept although none of them turned out to be its cause: a slot's graphs are always
    replayed on the same streams, the host never runs more than `slots` batches ahead, and no Tensor.record_stream is used.
    """

    def __init__(self, model, ahead, example, coords=None, no_grad=True, geometry_streams=1):
        """geometry_streams: how many batches' geometry may run at once, each on a high-priority stream of its own. 1 (two input
        slots) hides the geometry of batch i + 1 under the stacks of batch i; 2 (four slots: a slot's graphs are always replayed
        on the same stream: profiles/r05/geometry_ahead.txt) pays when the geometry stream is the bottleneck (sem_seg: 435 us of
        one-CU-per-cloud chains per batch on 8 of 256 CUs)."""
        self.ahead = ahead
        coords = coords or (lambda x: x)
        dev = example.device
        capture_stream = torch.cuda.Stream(device=dev)
        G = max(1, int(geometry_streams))
        S = 2 * G if G > 1 else 2                                       # a slot's graphs are always replayed on the same streams
        self._geo_streams = [ahead.stream] + [torch.cuda.Stream(device=dev, priority=-1) for _ in range(G - 1)]
        self._in = [example.clone() for _ in range(S)]
        self._geo_graphs, self._stack_graphs, self._sets, self._outs = [], [], [], []
        cur = torch.cuda.current_stream(dev)
        # no_grad=False: `model` is a whole training step (forward, loss, backward, optimiser step on static gradient
        # buffers) -- scripts/train_step_bench.py --graph; what must hold for such a capture is torch's (torch.cuda.graphs)
        with (torch.no_grad() if no_grad else torch.enable_grad()):
            for k in range(S):
                ahead.stream.wait_stream(cur)
                with torch.cuda.stream(ahead.stream):                   # warm-up on the capture streams (lazy initialisations)
                    g = ahead.compute(coords(self._in[k]))
                capture_stream.wait_stream(ahead.stream)
                with torch.cuda.stream(capture_stream):
                    model(self._in[k], g)
                torch.cuda.synchronize(dev)
                gg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gg, stream=ahead.stream):
                    self._sets.append(ahead.compute(coords(self._in[k])))
                sg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(sg, stream=capture_stream):
                    self._outs.append(model(self._in[k], self._sets[k]))
                self._geo_graphs.append(gg)
                self._stack_graphs.append(sg)
        self._held = [None] * S                                         # the caller's batch of each slot, until the slot's next use
        self._geo_done = [torch.cuda.Event() for _ in range(S)]
        self._stack_done = [torch.cuda.Event() for _ in range(S)]
        for e in self._stack_done:
            e.record(cur)
        self._i = 0

    def push(self, x, ready=None):
        """Enqueue one batch (same shape and dtype as the example) -> its output, valid in the current stream's order.
        ready: an event recorded after x's producer (a loader stream); False = x is complete already; None = x was produced on
        the current stream (an event is recorded there now -- which also orders this batch's geometry behind everything the
        current stream holds, the previous batch's stacks included: correct, but nothing overlaps)."""
        ref = self._in[0]
        if not (isinstance(x, torch.Tensor) and x.shape == ref.shape and x.dtype == ref.dtype and x.device == ref.device):
            raise ValueError("PipelinedInference.push: the batch must be a %s tensor of shape %s on %s like the example (the graphs "
                             "were captured for that)" % (ref.dtype, tuple(ref.shape), ref.device))
        k = self._i % len(self._in)
        a = self._geo_streams[k % len(self._geo_streams)]
        self._i += 1
        cur = torch.cuda.current_stream(x.device)
        # bounded run-ahead: the host waits HERE for the slot's previous batch (work that is `slots` batches old) instead of in a
        # full hardware queue a little later; after it nothing reads the batch the slot held, which is released below
        self._stack_done[k].synchronize()
        if ready is not False:
            a.wait_event(cur.record_event() if ready is None else ready)
        a.wait_event(self._stack_done[k])                               # the stacks that read this slot len(slots) pushes ago
        self._held[k] = x                                               # (no Tensor.record_stream: see GeometryAhead.submit)
        with torch.cuda.stream(a):
            self._in[k].copy_(x, non_blocking=True)
            self._geo_graphs[k].replay()
            self._geo_done[k].record(a)
        cur.wait_event(self._geo_done[k])
        self._stack_graphs[k].replay()
        self._stack_done[k].record(cur)
        return self._outs[k]
