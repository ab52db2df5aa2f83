"""Development aid: where does the overlapped launch's time go? Needs a library built with -DPN2_FUSED_LAB_TIMES
(producer 0 writes its chain duration, in 10 ns ticks of s_memrealtime, behind the status word of ws)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pointnet2_amd import synthetic as S
dev = torch.device("cuda:0")
st = bench.Stage(dev, S.sphere_clouds(32, 4096, 1000))
st.ws = torch.zeros((st.ws.numel() + 4096,), dtype=torch.uint8, device=dev)      # room for the per-producer stamps
for _ in range(5):
    st.overlap_()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
chains, tails = [], []
for s, e in ev:
    s.record(); st.overlap_(); e.record()
    torch.cuda.synchronize()
    off = st.lib.pn2_sample_and_group_status_offset(32, 1024)
    words = st.ws[off:off + 16].view(torch.int32).cpu().numpy().astype("int64") & 0xffffffff
    chains.append(int(words[1]) * 0.01)
    tails.append(((int(words[3]) - int(words[2])) & 0xffffffff) * 0.01)
    st.ws[off + 12:off + 16].zero_()
    stamps = (st.ws[off + 32:off + 32 + 8 * 32].view(torch.int32).cpu().numpy().astype("int64") & 0xffffffff).reshape(32, 2)
    last_stamps = stamps
tot = [s.elapsed_time(e) * 1e3 for s, e in ev]
print("last workgroup ends %.2f us (median; min %.2f, max %.2f) after producer 0's chain" % (np.median(tails), min(tails), max(tails)))
t0s, t1s = last_stamps[:, 0], last_stamps[:, 1]
base = t0s.min()
print("producers: start spread %.2f us, chain durations min %.1f / median %.1f / max %.1f us, end spread %.2f us"
      % ((t0s.max() - base) * 0.01, ((t1s - t0s) * 0.01).min(), np.median((t1s - t0s) * 0.01), ((t1s - t0s) * 0.01).max(),
         (t1s.max() - t1s.min()) * 0.01))
print("launch (HIP events) %.1f us median; producer-0 chain incl. staging %.1f us median; difference %.1f us"
      % (np.median(tot), np.median(chains), np.median(tot) - np.median(chains)))
