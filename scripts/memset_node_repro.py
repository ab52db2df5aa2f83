"""The same question as scripts/memset_node_repro.hip inside a PyTorch process: a hipMemsetAsync captured into a HIP graph
(torch.cuda.graph), replayed between EAGER torch kernels -- does the memset node still clear its buffer? The graph is
[hipMemsetAsync(buf, 0, bytes); out.copy_(buf)]; before every replay the buffer is poisoned with what a published granule looks
like (tag 1) and a few eager elementwise kernels of other sizes run (the serving loop's `output != expected` compares). After
every replay `out` must be zero. Several graphs, two streams, like pointnet2_amd.geometry.PipelinedInference.
    python scripts/memset_node_repro.py [replays] [graphs]
Findings: profiles/r06/stale_granules.md."""
import ctypes
import sys

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")
replays = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ngraphs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nbytes = 8 * 16 * 512 + 16
streams = [torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev, priority=-1)]
conv = torch.nn.Conv2d(16, 32, 1).to(dev)                      # MIOpen work first, like the failing process (profiles/r05/geometry_ahead.txt)
conv(torch.randn(8, 16, 64, 64, device=dev))
graphs = []
for k in range(ngraphs):
    s = streams[0]
    buf = torch.empty((nbytes // 8,), dtype=torch.int64, device=dev)
    out = torch.empty_like(buf)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        out.copy_(buf)
    graphs.append((g, buf, out, streams[k % 2]))
    conv(torch.randn(8, 16, 64, 64, device=dev))               # eager work between the captures
a = torch.randn(32, 40, device=dev)
b = a.clone()
bad = torch.zeros((replays,), dtype=torch.int64, device=dev)
first = torch.zeros((replays, 4), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for it in range(replays):
    g, buf, out, s = graphs[it % ngraphs]
    d = (a != b).sum()                                         # eager elementwise kernels on the caller's stream (1280 elements)
    with torch.cuda.stream(s):
        buf.fill_((1 << 32) | (it & 0xffff))                   # what the previous replay's granules look like: tag 1
        g.replay()
        nz = out != 0
        bad[it] = nz.sum()
        first[it] = out[-4:]
torch.cuda.synchronize()
nb = (bad != 0).nonzero().flatten().tolist()
print("memset node inside replayed torch graphs: %d of %d replays left non-zero words behind; first %s" % (len(nb), replays, nb[:8]))
for it in nb[:6]:
    print("   replay %d: %d of %d words non-zero; last four words %s" % (it, int(bad[it]), nbytes // 8, [hex(v & (2 ** 64 - 1)) for v in first[it].tolist()]))
print("DEFECT REPRODUCED" if nb else "clean in this run")
