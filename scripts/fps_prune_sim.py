"""Numpy simulation of the pruned FPS tier's box test (csrc/fps_pruned_body.h) on the bench clouds: how many groups does a round
touch, in total and in the busiest wave, by round index -- for the shipped 4 x 4 x 2 axis leaves and wave map, a kd-tree with 32 / 64
leaves, 64 axis leaves. Results: profiles/r06/fps_round6.txt. Uses the oracle for the sample order
(test infrastructure; this is a measurement aid, not product code)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from pointnet2_amd import synthetic as S
import oracle as O
F = np.float32

def leaves_axis(x, K=(4,4,2)):
    n = x.shape[0]
    ext = x.max(0) - x.min(0)
    a0, a1, a2 = np.argsort(-ext, kind="stable")
    ids = np.arange(n); out = []
    for i0, p0 in enumerate(np.array_split(ids[np.argsort(x[:, a0], kind="stable")], K[0])):
        for i1, p1 in enumerate(np.array_split(p0[np.argsort(x[p0, a1], kind="stable")], K[1])):
            for i2, p2 in enumerate(np.array_split(p1[np.argsort(x[p1, a2], kind="stable")], K[2])):
                out.append(((i0, i1, i2), p2))
    return out

def leaves_kd(x, depth):
    # balanced kd tree, widest axis per node
    def rec(ids, d):
        if d == 0: return [ids]
        ext = x[ids].max(0) - x[ids].min(0)
        a = int(np.argmax(ext))
        o = ids[np.argsort(x[ids, a], kind="stable")]
        h = len(o)//2
        return rec(o[:h], d-1) + rec(o[h:], d-1)
    return [((i,), g) for i, g in enumerate(rec(np.arange(x.shape[0]), depth))]

def chain(x, m):
    fps = O.farthest_point_sample(m, x[None])[0]
    n = x.shape[0]
    mind = np.full(n, 1e38, F)
    vs = np.zeros(m, F)
    for j in range(1, m):
        s = x[fps[j-1]]
        d = ((x[:,0]-s[0])**2 + (x[:,1]-s[1])**2 + (x[:,2]-s[2])**2).astype(F)
        mind = np.minimum(mind, d)
        vs[j] = mind[fps[j]]
    return fps, vs

def touched(x, fps, vs, groups):
    lo = np.stack([x[g].min(0) for _, g in groups]); hi = np.stack([x[g].max(0) for _, g in groups])
    m = len(fps)
    T = np.zeros((m, len(groups)), bool)
    vprev = F(1e38)
    for j in range(1, m):
        s = x[fps[j-1]]
        a = s[None,:] - np.clip(s[None,:], lo, hi)
        bd = (a*a).sum(1)
        thr = vprev * 1.00001 + 1e-30
        T[j] = ~(bd >= thr)
        vprev = vs[j]
    return T

def report(name, T, wave_of, W=4):
    m, G = T.shape
    per_wave = np.zeros((m, W), int)
    for g in range(G):
        per_wave[:, wave_of[g]] += T[:, g]
    busy = per_wave.max(1)
    rng = [(1,17),(17,33),(33,65),(65,129),(129,257),(257,513),(513,1024)]
    s = "%-28s total/round %.2f busiest %.3f |" % (name, T[1:].sum(1).mean(), busy[1:].mean())
    for a,b in rng:
        s += " [%d,%d) %.2f/%.2f" % (a,b, T[a:b].sum(1).mean(), busy[a:b].mean())
    print(s)
    return busy

if __name__ == "__main__":
    n, m = 4096, 1024
    for kind, mk in (("sphere", S.sphere_clouds), ("uniform", S.uniform_clouds)):
        cl = mk(2, n, 0)
        for b in range(2):
            x = cl[b].astype(F)
            fps, vs = chain(x, m)
            g = leaves_axis(x)
            T = touched(x, fps, vs, g)
            # current map: leaf id = 8a + r (a=i0, r = i1*2+i2); wave (r+a)%4
            wave_cur = [((i1*2+i2) + i0) % 4 for (i0,i1,i2), _ in g]
            wave_mod = [k % 4 for k in range(32)]
            busy = report("%s%d axis442 cur-map" % (kind,b), T, wave_cur)
            report("%s%d axis442 id%%4" % (kind,b), T, wave_mod)
            gk = leaves_kd(x, 5)
            Tk = touched(x, fps, vs, gk)
            report("%s%d kd32 id%%4" % (kind,b), Tk, [k % 4 for k in range(32)])
            gk6 = leaves_kd(x, 6)
            Tk6 = touched(x, fps, vs, gk6)
            report("%s%d kd64 id%%4" % (kind,b), Tk6, [k % 4 for k in range(64)])
            g64 = leaves_axis(x, (4,4,4))
            T64 = touched(x, fps, vs, g64)
            report("%s%d axis444 (r+a)%%4" % (kind,b), T64, [((i1*4+i2) + i0) % 4 for (i0,i1,i2), _ in g64])
            # histogram of busiest
            print("   busiest-wave histogram (cur):", np.bincount(busy[1:], minlength=9).tolist())

def sub_or(x, fps, vs):
    g64 = leaves_axis(x, (4,4,4))
    T64 = touched(x, fps, vs, g64)
    # group = pair of sub-leaves along the last axis: (i0,i1,i2//2)
    T32 = np.zeros((T64.shape[0], 32), bool)
    keys = []
    for k, ((i0,i1,i2), _) in enumerate(g64):
        gid = (i0*4 + i1)*2 + i2//2
        T32[:, gid] |= T64[:, k]
    wave = [0]*32
    for i0 in range(4):
        for i1 in range(4):
            for h in range(2):
                gid = (i0*4+i1)*2+h
                wave[gid] = ((i1*2+h) + i0) % 4
    return T32, wave
