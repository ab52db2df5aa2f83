"""The three ways the overlapped sample-and-group launch (csrc/sa_fused.hip) tells ITS granules from an older launch's, each
driven through the C ABI with inputs that CHANGE from launch to launch -- a granule accepted from the wrong launch is invisible
when every launch sees the same cloud:

* generation 0 -- pn2_sample_and_group_xyz, pn2_sa_level(generation = 0): workspace cleared on the stream, constant tag. Eager:
  600 launches, three inputs in rotation, every output compared on the device. Inside a capture the same call must enqueue the
  two launches (hipStreamIsCapturing guard) and leave the workspace untouched;
* PN2_GENERATION_DEVICE -- the launch numbers itself with an arrival counter per cloud: eager with a changing number of
  consumer workgroups per launch, and inside captured graphs replayed 600 times, two graphs at once on two streams;
* the Python operator inside torch.cuda.graph: takes a stocked workspace (device form) after an eager warm-up, the two
  launches without one.
What every form must produce is what the separate operators produce (reference utils/pointnet_util.py:40-46), bit for bit.
"""
import ctypes

import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEVICE_GEN = 0xFFFFFFFF
B, N, M, NS, R = 8, 2048, 256, 32, 0.2


def _inputs(cuda, count=3, b=B, n=N):
    return [torch.from_numpy(S.sphere_clouds(b, n, 11 + 7 * k)).to(cuda) for k in range(count)]


def _reference(xyz, m=M, ns=NS, r=R):
    """fps_idx, new_xyz, idx, cnt, grouped by the separate operators (themselves checked against the oracle in test_parity_gpu)."""
    import pointnet2_amd as P
    fps = P.farthest_point_sample(m, xyz)
    new_xyz = P.gather_point(xyz, fps)
    idx, cnt = P.query_ball_point(r, ns, xyz, new_xyz)
    grouped = P.group_point(xyz, idx) - new_xyz.unsqueeze(2)
    return fps, new_xyz, idx, cnt, grouped


class _Outs:
    def __init__(self, cuda, b=B, m=M, ns=NS):
        self.t = (torch.empty((b, m), dtype=torch.int32, device=cuda), torch.empty((b, m, 3), device=cuda),
                  torch.empty((b, m, ns), dtype=torch.int32, device=cuda), torch.empty((b, m), dtype=torch.int32, device=cuda),
                  torch.empty((b, m, ns, 3), device=cuda))

    def ptrs(self):
        return [t.data_ptr() for t in self.t]

    def mismatches(self, ref):
        d = None
        for o, r in zip(self.t, ref):
            e = (o != r).sum()
            d = e if d is None else d + e
        return d

    def poison(self):
        for t in self.t:
            t.fill_(-7)


def _ws(lib, cuda, b=B, m=M):
    return torch.zeros((lib.pn2_sample_and_group_ws_bytes(b, m),), dtype=torch.uint8, device=cuda)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_plain_entry_eager_with_rotating_inputs(cuda):
    """pn2_sample_and_group_xyz (clear + constant tag), eager, 600 launches, the input changing every launch."""
    from pointnet2_amd import _C
    lib = _C.lib()
    xs = _inputs(cuda)
    refs = [_reference(x) for x in xs]
    outs, ws = _Outs(cuda), _ws(lib, cuda)
    ws.fill_(0x5A)                                                       # the entry clears what it needs itself
    bad = torch.zeros((), dtype=torch.int64, device=cuda)
    for it in range(600):
        k = it % 3
        rc = lib.pn2_sample_and_group_xyz(B, N, M, R, NS, xs[k].data_ptr(), ws.data_ptr(), *outs.ptrs(), 1, _stream())
        assert rc == 0
        bad += outs.mismatches(refs[k])
    torch.cuda.synchronize()
    assert int(bad) == 0
    off = lib.pn2_sample_and_group_status_offset(B, M)
    assert int(ws[off:off + 4].view(torch.int32)) == 0


def test_sa_level_generation_zero_eager_with_rotating_inputs(cuda):
    """pn2_sa_level(generation = 0): the level entry's cleared form, same soak; the pooled features against the level built
    from the separate operators' geometry (pn2_sa_mlp3_maxpool on the reference idx)."""
    import numpy as np
    from pointnet2_amd import _C, sa_mlp
    lib = _C.lib()
    rng = np.random.default_rng(1)
    dims = (3, 32, 32, 64)
    packed = sa_mlp.PackedMLP3([((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
                                 (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)], cuda, nsample=NS)
    xs = _inputs(cuda)
    refs = [_reference(x) for x in xs]
    pooled = [sa_mlp.sa_mlp_maxpool(x, r[1], None, r[2], packed) for x, r in zip(xs, refs)]
    outs, ws = _Outs(cuda), _ws(lib, cuda)
    out = torch.empty((B, M, 64), device=cuda)
    nbytes = lib.pn2_sa_mlp3_ws_bytes(B, N, M, packed.cin, 32, 32, 64, NS)
    wsm = torch.empty((max(1, (nbytes + 3) // 4),), device=cuda)
    bad = torch.zeros((), dtype=torch.int64, device=cuda)
    for it in range(600):
        k = it % 3
        rc = lib.pn2_sa_level(B, N, M, R, NS, 0, xs[k].data_ptr(), None, ws.data_ptr(), 0, None, 32, 32, 64, packed.wp.data_ptr(),
                              packed.bp.data_ptr(), *outs.ptrs(), out.data_ptr(), wsm.data_ptr(), _stream())
        assert rc == 0
        bad += outs.mismatches(refs[k]) + (out != pooled[k]).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0


def test_device_numbered_launches_eager_with_changing_consumer_counts(cuda):
    """PN2_GENERATION_DEVICE: no clear, no number from the host. 600 launches on ONE workspace, rotating inputs, the number of
    consumer workgroups per cloud changing from launch to launch (the arrival count must not depend on it); afterwards every
    cloud's counter word reads (ordinal = launches, arrivals = 0)."""
    from pointnet2_amd import _C
    lib = _C.lib()
    xs = _inputs(cuda)
    refs = [_reference(x) for x in xs]
    outs, ws = _Outs(cuda), _ws(lib, cuda)
    bad = torch.zeros((), dtype=torch.int64, device=cuda)
    launches = 600
    for it in range(launches):
        k = it % 3
        if it % 50 == 0:
            outs.poison()
        rc = lib.pn2_sample_and_group_xyz_ex(B, N, M, R, NS, xs[k].data_ptr(), ws.data_ptr(), DEVICE_GEN, 0, (0, 1, 2, 4)[it % 4],
                                             *outs.ptrs(), 1, _stream())
        assert rc == 0
        bad += outs.mismatches(refs[k])
    torch.cuda.synchronize()
    assert int(bad) == 0
    off = lib.pn2_sample_and_group_status_offset(B, M)
    assert int(ws[off:off + 4].view(torch.int32)) == 0
    words = ws[off + 16:off + 16 + 8 * B].view(torch.int64).tolist()
    assert words == [launches << 16] * B, [hex(w) for w in words]


def _capture(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    return g


def test_device_numbered_launch_inside_graphs_on_two_streams(cuda):
    """Two graphs, each ONE device-numbered overlapped launch on a workspace of its own and a static input, replayed 600 times
    on two streams at once while the inputs rotate (copied into the static input on the graph's stream in front of every
    replay). Round 5's serving loop failed exactly here with the cleared form."""
    from pointnet2_amd import _C
    lib = _C.lib()
    xs = _inputs(cuda)
    refs = [_reference(x) for x in xs]
    streams = [torch.cuda.Stream(device=cuda), torch.cuda.Stream(device=cuda, priority=-1)]
    slots = []
    for s in streams:
        outs, ws, xin = _Outs(cuda), _ws(lib, cuda), xs[0].clone()
        torch.cuda.synchronize()

        def launch(outs=outs, ws=ws, xin=xin):
            rc = lib.pn2_sample_and_group_xyz_gen(B, N, M, R, NS, xin.data_ptr(), ws.data_ptr(), DEVICE_GEN, *outs.ptrs(), 1, _stream())
            assert rc == 0
        slots.append((s, outs, ws, xin, _capture(launch, s)))
    bad = [torch.zeros((), dtype=torch.int64, device=cuda) for _ in slots]
    torch.cuda.synchronize()
    for it in range(600):
        for j, (s, outs, ws, xin, g) in enumerate(slots):
            k = (it + j) % 3
            with torch.cuda.stream(s):
                xin.copy_(xs[k])
                g.replay()
                bad[j] += outs.mismatches(refs[k])
    torch.cuda.synchronize()
    assert [int(x) for x in bad] == [0, 0]
    off = lib.pn2_sample_and_group_status_offset(B, M)
    for _s, _o, ws, _x, _g in slots:
        assert ws[off + 16:off + 16 + 8 * B].view(torch.int64).tolist() == [600 << 16] * B


def test_plain_entry_inside_a_capture_takes_the_two_launches(cuda):
    """generation 0 inside a capture: the library must not enqueue memset + constant-tag launch (the form that failed); it
    enqueues FPS + ball query instead and never touches the workspace."""
    from pointnet2_amd import _C
    lib = _C.lib()
    xs = _inputs(cuda)
    refs = [_reference(x) for x in xs]
    outs, ws, xin = _Outs(cuda), _ws(lib, cuda), xs[0].clone()
    ws.fill_(0xA5)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(device=cuda)

    def launch():
        assert lib.pn2_sample_and_group_xyz(B, N, M, R, NS, xin.data_ptr(), ws.data_ptr(), *outs.ptrs(), 1, _stream()) == 0
        assert lib.pn2_sample_and_group_xyz_ex(B, N, M, R, NS, xin.data_ptr(), ws.data_ptr(), 0, 0, 0, *outs.ptrs(), 1, _stream()) == 0
    g = _capture(launch, s)
    bad = torch.zeros((), dtype=torch.int64, device=cuda)
    for it in range(60):
        k = it % 3
        with torch.cuda.stream(s):
            xin.copy_(xs[k])
            g.replay()
            bad += outs.mismatches(refs[k])
    torch.cuda.synchronize()
    assert int(bad) == 0
    assert bool((ws == 0xA5).all()), "the captured generation-0 call wrote to its workspace: it launched the overlapped kernel"


def test_python_operator_inside_a_graph(cuda):
    """sample_and_group_xyz under torch.cuda.graph: with a stocked workspace (any eager call of the shape stocks them) the
    captured level is the device-numbered overlapped launch on a workspace of its own; with an empty stock it is the two
    launches. Both replayed with rotating inputs against the separate operators."""
    import pointnet2_amd as P
    import pointnet2_amd.tf_grouping as G
    xs = _inputs(cuda)
    refs = [_reference(x) for x in xs]
    xin = xs[0].clone()
    side = torch.cuda.Stream(device=cuda)
    for stocked in (True, False):
        G._SPARES.clear()
        if stocked:
            P.sample_and_group_xyz(M, R, NS, xin)                        # eager warm-up: stocks workspaces for (B, M)
            assert len(G._SPARES[(cuda.index, B, M)]) == G._SPARE_STOCK
        torch.cuda.synchronize()
        taken = len(G._CAPTURED)
        box = []
        g = _capture(lambda: box.append(P.sample_and_group_xyz(M, R, NS, xin)), side)
        assert len(G._CAPTURED) == taken + (1 if stocked else 0)
        out = box[0]
        bad = torch.zeros((), dtype=torch.int64, device=cuda)
        for it in range(300):
            k = it % 3
            with torch.cuda.stream(side):
                xin.copy_(xs[k])
                g.replay()
                for o, r in zip(out, refs[k]):
                    bad += (o != r).sum()
        torch.cuda.synchronize()
        assert int(bad) == 0
        G.check_overlapped_launches(cuda)
        if stocked:
            ent = G._CAPTURED[-1][1]
            words = ent[0][ent[5] + 16:ent[5] + 16 + 8 * B].view(torch.int64).tolist()
            assert words == [300 << 16] * B                               # the captured launch really was the overlapped one
