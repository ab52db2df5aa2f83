"""Development aid (library built with -DPN2_FUSED_LAB_TIMES, see fused_tail_probe.py): are the slow chains of the overlapped
launch the same CLOUDS launch after launch (the data's lists) or different ones (timing)? Prints every cloud's chain duration
over 12 launches and the rank correlation between launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pointnet2_amd import synthetic as S
dev = torch.device("cuda:0")
st = bench.Stage(dev, S.sphere_clouds(32, 4096, 1000))
st.ws = torch.zeros((st.ws.numel() + 4096,), dtype=torch.uint8, device=dev)
for _ in range(5):
    st.overlap_()
torch.cuda.synchronize()
off = st.lib.pn2_sample_and_group_status_offset(32, 1024)
dur = []
for _ in range(12):
    st.overlap_()
    torch.cuda.synchronize()
    stamps = (st.ws[off + 32:off + 32 + 8 * 32].view(torch.int32).cpu().numpy().astype("int64") & 0xffffffff).reshape(32, 2)
    dur.append((stamps[:, 1] - stamps[:, 0]) * 0.01)
dur = np.array(dur)
print("per cloud, mean over 12 launches (us):", np.round(dur.mean(0), 1))
print("per cloud, std over launches (us):    ", np.round(dur.std(0), 2))
print("spread between clouds (max - min of the means): %.1f us; mean within-cloud std %.2f us" % (dur.mean(0).max() - dur.mean(0).min(), dur.std(0).mean()))
r = np.corrcoef(dur)
print("correlation of the clouds' durations between launches: min %.3f mean %.3f" % (r[np.triu_indices(12, 1)].min(), r[np.triu_indices(12, 1)].mean()))
print("slowest cloud per launch:", dur.argmax(1), " max per launch:", np.round(dur.max(1), 1))
