"""The algorithm behind pn2_knn_point (csrc/topk.hip), as a numpy model on the CPU: replaying the
reference's k swap rounds (tf_grouping_g.cu:83-123) on the candidate set {value <= tau}, tau >= v_k, with
slot tracking, gives exactly the first k outputs of the literal selection sort -- for ANY such tau
(including the loose bound from the per-lane minima) and on arrays full of exact ties. The HIP kernel is
checked against the oracle on the GPU (tests/test_parity_gpu.py); this pins the reasoning itself."""
import numpy as np


def literal_rounds(v, k):
    v = v.copy()
    idx = np.arange(len(v))
    for s in range(k):
        mn = s
        for t in range(s + 1, len(v)):
            if v[t] < v[mn]:
                mn = t
        if mn != s:
            v[mn], v[s] = v[s], v[mn]
            idx[mn], idx[s] = idx[s], idx[mn]
    return v[:k].copy(), idx[:k].copy()


def replay_on_candidates(v, k, tau):
    cand = [i for i in range(len(v)) if v[i] <= tau]
    pos = {c: c for c in cand}                       # current slot of every live candidate
    out_v, out_i = [], []
    for s in range(k):
        win = min(pos, key=lambda c: (v[c], pos[c]))
        q = pos.pop(win)
        out_v.append(v[win])
        out_i.append(win)
        if q != s:                                   # the swap: whoever sits at slot s moves to slot q
            for c in pos:
                if pos[c] == s:
                    pos[c] = q
    return np.array(out_v), np.array(out_i)


def lane_minima_bound(v, k):
    """k-th smallest of the 64 strided per-lane minima (what the kernel uses for k <= 40)."""
    mins = sorted(v[l::64].min() for l in range(min(64, len(v))))
    return mins[k - 1]


def test_replay_equals_literal_selection_sort_for_any_valid_tau():
    rng = np.random.default_rng(0)
    for trial in range(1500):
        n = int(rng.integers(2, 90))
        k = int(rng.integers(1, n + 1))
        v = rng.integers(0, int(rng.integers(1, 8)), size=n).astype(np.float64)      # heavy ties
        vk = np.sort(v)[k - 1]
        want = literal_rounds(v, k)
        for tau in (vk, vk + 0.5, vk + 3.0, v.max()):
            got = replay_on_candidates(v, k, tau)
            assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]), (trial, tau)


def test_lane_minima_bound_is_a_valid_threshold():
    rng = np.random.default_rng(1)
    for trial in range(300):
        n = int(rng.integers(64, 700))
        k = int(rng.integers(1, 41))
        v = np.round(rng.random(n) * rng.choice([4, 50, 1000])) / 7.0
        tau = lane_minima_bound(v, k)
        assert tau >= np.sort(v)[k - 1]
        want = literal_rounds(v, k)
        got = replay_on_candidates(v, k, tau)
        assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]), trial
