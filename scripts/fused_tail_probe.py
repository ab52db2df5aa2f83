"""Development aid: where does the overlapped launch's time go? Needs a library built with -DPN2_FUSED_LAB_TIMES
(producer 0 writes its chain duration, in 10 ns ticks of s_memrealtime, behind the status word of ws)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pointnet2_amd import synthetic as S
dev = torch.device("cuda:0")
st = bench.Stage(dev, S.sphere_clouds(32, 4096, 1000))
for _ in range(5):
    st.overlap_()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
chains = []
for s, e in ev:
    s.record(); st.overlap_(); e.record()
    torch.cuda.synchronize()
    off = st.lib.pn2_sample_and_group_status_offset(32, 1024)
    chains.append(int(st.ws[off + 4:off + 8].view(torch.int32).item()) * 0.01)
tot = [s.elapsed_time(e) * 1e3 for s, e in ev]
print("launch (HIP events) %.1f us median; producer-0 chain incl. staging %.1f us median; difference %.1f us"
      % (np.median(tot), np.median(chains), np.median(tot) - np.median(chains)))
