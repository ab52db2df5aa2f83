"""Numpy simulation of BATCHED farthest point sampling (round 6, csrc/fps_batch_body.h): how many samples can one arg-max
exchange yield?

A batch starts from the running distances after j samples. Every lane (unit = the 16 rank slots of one thread) whose best running
distance is >= theta = f * (value of the last sample) is a CANDIDATE (at most CAP per wave; a wave with more raises its own
threshold by bisection). All points that are not a candidate's best point have a value < B = max(theta_w, second-best values of
the candidate lanes). The candidates are then picked greedily: the best candidate IS the next sample as long as its value is
>= B (everything outside the candidate list is below B and running distances only fall); the pick lowers the other candidates'
values and the loop repeats. The batch ends when the best candidate falls below B.

The simulation reports samples per batch by round range and checks that the picks are the oracle's sequence.
Uses the oracle for the reference order (measurement aid, not product code)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from pointnet2_amd import synthetic as S
import oracle as O
from fps_prune_sim import leaves_axis
F = np.float32


def lane_layout(x, W=4):
    """point -> (wave, lane) as the pruned tier deals them: leaf (a, r) -> wave (r + a) % 4, position p -> lane p % 64"""
    n = x.shape[0]
    wave = np.zeros(n, int); lane = np.zeros(n, int)
    for (i0, i1, i2), ids in leaves_axis(x):
        a = i0; r = i1 * 2 + i2
        wave[ids] = (r + a) % W
        lane[ids] = np.arange(len(ids)) % 64
    return wave * 64 + lane


def d2(x, s):
    return ((x[:, 0] - s[0]) ** 2 + (x[:, 1] - s[1]) ** 2 + (x[:, 2] - s[2]) ** 2).astype(F)


def simulate(x, m, f0=0.9, cap=8, adapt=True, verbose=False, maxb=64):
    n = x.shape[0]
    fps = O.farthest_point_sample(m, x[None])[0]
    unit = lane_layout(x)
    nunits = 256
    # tie rank of the reference: smaller (k % 512, k) wins
    rank = (np.arange(n) % 512) * ((n + 511) // 512) + np.arange(n) // 512
    td = np.full(n, 1e38, F)
    td = np.minimum(td, d2(x, x[0]))
    picks = [0]
    batches = []
    vlast = F(1e38)
    f = f0
    members = [np.where(unit == u)[0] for u in range(nunits)]
    while len(picks) < m:
        theta = F(f) * vlast if vlast < 1e37 else F(0)
        # per unit: best (by value, then rank) and second-best value
        best = np.zeros(nunits, int); bv = np.zeros(nunits, F); sv = np.zeros(nunits, F)
        for u in range(nunits):
            ids = members[u]
            o = np.lexsort((rank[ids], -td[ids]))
            best[u] = ids[o[0]]; bv[u] = td[ids[o[0]]]; sv[u] = td[ids[o[1]]]
        cand = []
        Btheta = F(0); Bsec = F(-1)
        overflow = 0
        for w in range(4):
            us = np.arange(w * 64, w * 64 + 64)
            th = theta
            sel = us[bv[us] >= th]
            if len(sel) == 0:             # the wave lowers its threshold until it has a candidate
                th = bv[us].max()
                sel = us[bv[us] >= th]
            if len(sel) > cap:            # keep the cap best; the threshold becomes the value of the first dropped (strict)
                overflow += 1
                o = np.argsort(-bv[sel], kind="stable")
                dropped = sel[o[cap:]]
                sel = sel[o[:cap]]
                Bsec = max(Bsec, bv[dropped].max())
            Btheta = max(Btheta, th)
            cand += list(sel)
            if len(sel):
                Bsec = max(Bsec, sv[sel].max())
        cand = np.array(cand, int)
        cp = best[cand]; cv = td[cp].copy()
        a = 0
        while len(picks) < m and a < maxb:
            o = np.lexsort((rank[cp], -cv))
            c = o[0]
            ok = a == 0 or (cv[c] >= Btheta and cv[c] > Bsec)
            if not ok:
                break
            p = cp[c]
            assert p == fps[len(picks)], (len(picks), p, fps[len(picks)])
            picks.append(p); vlast = cv[c]
            dd = d2(x, x[p])
            td = np.minimum(td, dd)
            cv = np.minimum(cv, dd[cp])
            a += 1
        batches.append((len(picks) - a, a, len(cand), overflow))
        if adapt:
            if len(cand) > 3 * cap: f = min(0.98, f ** 0.7)
            elif len(cand) < 2 * cap: f = max(0.3, f ** 1.3)
    return np.array(batches)


if __name__ == "__main__":
    n, m = 4096, 1024
    kinds = (("sphere", S.sphere_clouds), ("uniform", S.uniform_clouds))
    for cap in (4, 8, 16):
        for kind, mk in kinds:
            x = mk(1, n, 0)[0].astype(F)
            for f0, adapt in ((0.9, True), (0.8, False), (0.9, False)):
                b = simulate(x, m, f0=f0, cap=cap, adapt=adapt)
                s = "%-8s cap %2d f0 %.2f adapt %d: batches %4d, mean A %.2f, overflow batches %d |" % (
                    kind, cap, f0, adapt, len(b), b[:, 1].mean(), (b[:, 3] > 0).sum())
                for lo, hi in ((1, 65), (65, 257), (257, 513), (513, 1024)):
                    q = b[(b[:, 0] >= lo) & (b[:, 0] < hi)]
                    s += " [%d,%d) nb %d A %.2f c %.1f" % (lo, hi, len(q), q[:, 1].mean() if len(q) else 0, q[:, 2].mean() if len(q) else 0)
                print(s, flush=True)
