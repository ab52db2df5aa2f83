"""CPU model of the batched FPS tier's acceptance rule (csrc/fps_batch_body.h): one candidate LIST per batch
determines several samples. Restated in numpy with the kernel's arithmetic -- values compared as fp32 BIT PATTERNS, threshold
theta = fl32(v_last * (1 - g)), at most 8 candidate lanes per updater wave (bisection on the bits, exact best lane when that fails or
nobody reaches theta), the bound = max(wave thresholds, second-best value of a candidate lane + 1 ulp), greedy picks by
(value, tie rank) while the pick's value bits are >= the bound, the list-size feedback on g, the first samples taken one per
exchange, the tail once a value is 0 -- and run as whole chains against the oracle's sequential sampling. The assertion inside
the pick loop is the claim itself: every accepted pick IS the arg-max over the WHOLE cloud at that step. No GPU (the device
tier is tested against the same oracle in tests/test_parity_gpu.py)."""
import numpy as np
import pytest

from pointnet2_amd import synthetic as S

F = np.float32
REF_THREADS = 512
WAVES, CAP, LIST_HI, LIST_LO, G0, EARLY = 8, 8, 36, 20, F(0.10), 48   # fps_batch_body.h: updater waves, 64 / waves, PN2_BT_LIST_HI / _LO, PN2_BT_G0, PN2_BT_EARLY


def _bits(v):
    return np.asarray(v, dtype=F).view(np.uint32).astype(np.int64)


def _sqdist(p, s):
    dx, dy, dz = (p[:, 0] - s[0]).astype(F), (p[:, 1] - s[1]).astype(F), (p[:, 2] - s[2]).astype(F)
    return ((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F) + (dz * dz).astype(F)


def _lanes(x, slots):
    """point -> (wave, lane): 32 leaves of equal size along the axes sorted by extent, leaf (a, r) -> wave (r + a) % 8, position
    p of the leaf -> lane p % 64 (fps_pruned_prologue; any dealing gives the same samples, this one gives the kernel's lists)"""
    n = x.shape[0]
    ext = x.max(axis=0) - x.min(axis=0)
    a0, a1, a2 = np.argsort(-ext, kind="stable")
    ids = np.arange(n)
    unit = np.zeros(n, dtype=np.int64)
    for a, p0 in enumerate(np.array_split(ids[np.argsort(x[:, a0], kind="stable")], 4)):
        for i1, p1 in enumerate(np.array_split(p0[np.argsort(x[p0, a1], kind="stable")], 4)):
            for i2, p2 in enumerate(np.array_split(p1[np.argsort(x[p1, a2], kind="stable")], 2)):
                r = i1 * 2 + i2
                unit[p2] = ((r + a) % WAVES) * 64 + np.arange(len(p2)) % 64
    return unit


def _batched_fps(x, m, early=EARLY, runs=None):
    """runs: None, or a function batch number -> how many samples to take one per exchange after that batch (the kernel decides
    that by its clock -- SLOW BATCHES in fps_batch_body.h --, so any schedule must give the oracle's samples)"""
    x = x.astype(F)
    n = x.shape[0]
    q = (n + REF_THREADS - 1) // REF_THREADS
    rank = (np.arange(n) % REF_THREADS) * q + np.arange(n) // REF_THREADS             # smaller wins a tie (tf_sampling_g.cu:146,153-163)
    unit = _lanes(x, 16)
    members = [np.nonzero(unit == u)[0] for u in range(64 * WAVES)]
    td = np.minimum(np.full(n, 1e38, dtype=F), _sqdist(x, x[0]))
    out = [0]

    def argmax_all():
        best = td.max()
        c = np.nonzero(td == best)[0]
        return int(c[np.argmin(rank[c])]), best

    vlast = F(1e38)
    while len(out) < min(early, m):                                                       # EARLY: one sample per exchange
        p, vlast = argmax_all()
        out.append(p)
        td = np.minimum(td, _sqdist(x, x[p]))
    g = G0
    theta_b = int(_bits(F(vlast * F(F(1.0) - g)))) if len(out) > 1 else int(_bits(F(1e38)))
    vlast_b = int(_bits(vlast))
    batches = []
    while len(out) < m:
        # COLLECT: per lane the best point (value, then rank) and the second-best value
        lane_best = np.full(64 * WAVES, -1, dtype=np.int64)
        vb = np.zeros(64 * WAVES, dtype=np.int64)
        sb = np.zeros(64 * WAVES, dtype=np.int64)
        for u in range(64 * WAVES):
            ids = members[u]
            if len(ids) == 0:
                continue                                                                  # a lane of padding slots: value 0, never a candidate that matters
            o = np.lexsort((rank[ids], -td[ids].astype(np.float64)))
            lane_best[u] = ids[o[0]]
            vb[u] = _bits(td[ids[o[0]]])
            sb[u] = _bits(td[ids[o[1]]]) if len(ids) > 1 else 0
        cand, bound, total = [], 0, 0
        for w in range(WAVES):
            us = np.arange(w * 64, w * 64 + 64)
            us = us[lane_best[us] >= 0]
            thb = theta_b
            sel = us[vb[us] >= thb]
            exact = False
            if len(sel) > CAP:
                lob, hib = theta_b, vlast_b + 1
                for _ in range(16):
                    mid = lob + ((hib - lob) >> 1)
                    if mid == lob:
                        break
                    s2 = us[vb[us] >= mid]
                    if len(s2) > CAP:
                        lob = mid
                    elif len(s2) == 0:
                        hib = mid
                    else:
                        sel, thb = s2, mid
                        break
                exact = len(sel) > CAP
            elif len(sel) == 0:
                exact = True
            if exact:
                had = len(sel)
                o = np.lexsort((rank[lane_best[us]], -vb[us]))
                sel = us[o[:1]]
                if had != 0:
                    thb = int(vb[sel[0]]) + 1
            bound = max(bound, thb, int((sb[sel] + 1).max()))
            cand += list(sel)
            total += len(sel)
        cand = np.array(cand, dtype=np.int64)
        cp = lane_best[cand]
        cv = td[cp].copy()
        alive = np.ones(len(cp), dtype=bool)
        a = 0
        filled = False
        while len(out) < m and a < 64:
            live = np.nonzero(alive)[0]
            if len(live) == 0:
                break
            o = np.lexsort((rank[cp[live]], -cv[live].astype(np.float64)))
            c = live[o[0]]
            bh = int(_bits(cv[c]))
            if a > 0 and bh < bound:
                break
            p = int(cp[c])
            # THE CLAIM: the accepted pick is the arg-max over the whole cloud
            pa, va = argmax_all()
            assert p == pa and cv[c] == va, "sample %d: the list's pick %d is not the cloud's arg-max %d" % (len(out), p, pa)
            out.append(p)
            a += 1
            vlast_b = bh
            alive[c] = False
            d = _sqdist(x, x[p])
            td = np.minimum(td, d)
            cv = np.minimum(cv, d[cp])
            if bh == 0:
                filled = True
                break
        batches.append(a)
        if filled:
            while len(out) < m:
                out.append(out[-1])                                                       # every running distance is 0: point 0 again and again
            break
        if total > LIST_HI:
            g = max(F(g * F(0.8)), F(1.0 / 128.0))
        elif total < LIST_LO:
            g = min(F(g * F(1.25)), F(0.5))
        theta_b = int(_bits(F(_bits_to_float(vlast_b) * F(F(1.0) - g))))
        single = min(runs(len(batches)) if runs else 0, m - len(out))
        if single:                                                                       # one per exchange, then lists again from theta = (1 - G0) v_last
            for _ in range(single):
                p, vlast = argmax_all()
                out.append(p)
                td = np.minimum(td, _sqdist(x, x[p]))
            vlast_b = int(_bits(vlast))
            theta_b = int(_bits(F(vlast * F(F(1.0) - G0))))
    return np.array(out, dtype=np.int32), batches


def _bits_to_float(b):
    return np.array([b], dtype=np.uint32).view(F)[0]


CASES = [("sphere", S.sphere_clouds, 4096, 400), ("uniform", S.uniform_clouds, 4096, 300), ("duplicated", S.duplicated_clouds, 2500, 300),
         ("lattice", S.lattice_clouds, 3000, 200), ("dropout", S.dropout_clouds, 2100, 400), ("identical", S.identical_clouds, 2100, 100)]


@pytest.mark.parametrize("name,make,n,m", CASES, ids=[c[0] for c in CASES])
def test_a_list_determines_several_samples_and_they_are_the_oracles(oracle, name, make, n, m):
    clouds = make(1, n, 17)
    want = oracle.farthest_point_sample(m, clouds)
    got, batches = _batched_fps(clouds[0], m)
    assert np.array_equal(got, want[0]), "%s: first mismatch at %s" % (name, np.argwhere(got != want[0])[:3].ravel())
    if name in ("sphere", "uniform"):
        assert np.mean(batches) > 4.0, "a batch should yield several samples on a smooth cloud: %.2f" % np.mean(batches)


def test_batches_without_the_early_phase_are_exact_too(oracle):
    clouds = S.sphere_clouds(1, 4096, 23)
    want = oracle.farthest_point_sample(200, clouds)
    got, _ = _batched_fps(clouds[0], 200, early=1)
    assert np.array_equal(got, want[0])


@pytest.mark.parametrize("name,make,n,m", CASES[:4], ids=[c[0] for c in CASES[:4]])
def test_runs_of_single_rounds_between_batches_are_exact(oracle, name, make, n, m):
    """the kernel leaves batches for R samples when its clock says they do not pay, R = 16, 32, ...: whatever the schedule, the
    samples are the oracle's (the hand-over is the threshold: theta restarts from the last single sample's value)"""
    clouds = make(1, n, 29)
    want = oracle.farthest_point_sample(m, clouds)
    for runs in (lambda k: 16 if k % 3 == 2 else 0, lambda k: 5 * k % 23, lambda k: 64 if k == 2 else 0):
        got, _ = _batched_fps(clouds[0], m, runs=runs)
        assert np.array_equal(got, want[0]), "%s: first mismatch at %s" % (name, np.argwhere(got != want[0])[:3].ravel())
