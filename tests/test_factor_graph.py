"""Index parity of the host factor-graph mirror against the oracle restatement of the reference's
Python loops (oracle/graph_oracle.py) on random graphs and distance matrices."""
import numpy as np
import pytest

from oracle import graph_oracle as go


def _random_graph(rng, kf_idx, n_act, n_ina):
    from nerfslam.factor_graph import FactorGraph
    g = FactorGraph(max_factors=48)
    pairs = set()
    while len(pairs) < n_act + n_ina:
        i, j = (int(v) for v in rng.integers(0, kf_idx + 1, 2))
        if i != j:
            pairs.add((i, j))
    pairs = list(pairs)
    rng.shuffle(pairs)
    a, b = pairs[:n_act], pairs[n_act:]
    g.ii, g.jj = np.array([p[0] for p in a], np.int64), np.array([p[1] for p in a], np.int64)
    g.age = rng.integers(0, 30, len(a)).astype(np.int64)
    g.ii_inactive = np.array([p[0] for p in b], np.int64)
    g.jj_inactive = np.array([p[1] for p in b], np.int64)
    return g, a, b


@pytest.mark.parametrize("seed", range(12))
def test_proximity_edges_match_reference_loops(seed):
    rng = np.random.default_rng(seed)
    kf_idx = int(rng.integers(6, 30))
    g, act, ina = _random_graph(rng, kf_idx, int(rng.integers(0, 40)), int(rng.integers(0, 30)))
    kf0 = max(kf_idx - 4, 0) if seed % 2 else 0
    kf1 = max(kf_idx + 1 - 25, 0) if seed % 2 else 0
    rad, nms = (2, 1) if seed % 2 else (2, 2)
    t = kf_idx + 1
    d = rng.uniform(0, 40, (t - kf0) * (t - kf1)).astype(np.float32)
    d[rng.uniform(size=d.shape) < 0.1] = 1000.0
    thresh = 16.0
    ref = go.proximity_factors(d, act + ina, kf_idx, kf0, kf1, rad, nms, thresh, g.max_factors)
    got = g.proximity_edges(d, kf_idx, kf0, kf1, rad, nms, thresh)
    assert got == ref


@pytest.mark.parametrize("seed", range(8))
def test_add_remove_keyframe_sequences(seed):
    from nerfslam.factor_graph import FactorGraph
    rng = np.random.default_rng(100 + seed)
    g = FactorGraph(max_factors=20)
    # oracle state
    ii, jj, age, ii_in, jj_in = [np.zeros(0, np.int64) for _ in range(5)]
    for step in range(12):
        n = int(rng.integers(1, 9))
        ci, cj = rng.integers(0, 10, n), rng.integers(0, 10, n)
        keep = ci != cj
        ci, cj = ci[keep].astype(np.int64), cj[keep].astype(np.int64)
        # de-duplicate inside the batch (the reference never passes duplicates inside one call)
        _, first = np.unique(ci * 100 + cj, return_index=True)
        ci, cj = ci[np.sort(first)], cj[np.sort(first)]
        # ---- oracle ----
        k = go.filter_repeated_edges(ci, cj, list(zip(ii.tolist(), jj.tolist())), list(zip(ii_in.tolist(), jj_in.tolist())))
        ni, nj = ci[k], cj[k]
        if len(ni):
            m = go.add_factors_removal_mask(age, len(ni), 20)
            if m is not None:
                ii_in, jj_in = np.concatenate([ii_in, ii[m]]), np.concatenate([jj_in, jj[m]])
                ii, jj, age = ii[~m], jj[~m], age[~m]
            ii, jj, age = np.concatenate([ii, ni]), np.concatenate([jj, nj]), np.concatenate([age, np.zeros_like(ni)])
        # ---- product ----
        g.add(ci, cj, remove=True)
        np.testing.assert_array_equal(g.ii, ii)
        np.testing.assert_array_equal(g.jj, jj)
        np.testing.assert_array_equal(g.ii_inactive, ii_in)
        np.testing.assert_array_equal(g.jj_inactive, jj_in)
        age = age + 1
        g.age = g.age + 1
        if step % 4 == 3:  # age-out, then drop a keyframe
            old = age > 3
            ii_in, jj_in = np.concatenate([ii_in, ii[old]]), np.concatenate([jj_in, jj[old]])
            ii, jj, age = ii[~old], jj[~old], age[~old]
            g.remove(g.age > 3, store=True)
            kf = int(rng.integers(0, 10))
            ka, ii, jj = go.rm_keyframe_edges(ii, jj, kf)
            age = age[ka]
            _, ii_in, jj_in = go.rm_keyframe_edges(ii_in, jj_in, kf)
            g.remove_keyframe(kf)
            np.testing.assert_array_equal(g.ii, ii)
            np.testing.assert_array_equal(g.jj, jj)
            np.testing.assert_array_equal(g.age, age)
            np.testing.assert_array_equal(g.ii_inactive, ii_in)


def test_neighborhood_and_ba_edges():
    from nerfslam.factor_graph import FactorGraph
    for (a, b, r, st) in ((0, 8, 3, False), (2, 6, 2, True)):
        i, j = FactorGraph.neighborhood_edges(a, b, r, st)
        ri, rj = go.neighborhood_factors(a, b, r, st)
        np.testing.assert_array_equal(i, ri)
        np.testing.assert_array_equal(j, rj)
    g = FactorGraph()
    g.ii, g.jj = np.array([5, 6, 7]), np.array([6, 7, 5])
    g.ii_inactive, g.jj_inactive = np.array([1, 2, 3, 4]), np.array([4, 5, 1, 6])
    ii, jj, m = g.ba_edges(kf0=5)
    np.testing.assert_array_equal(m, [False, True, False, True])
    np.testing.assert_array_equal(ii, [2, 4, 5, 6, 7])


def test_reset_keeps_the_version_monotonic():
    from nerfslam.factor_graph import FactorGraph
    g = FactorGraph(max_factors=10)
    g.add([1, 2], [2, 1])
    v = g.version
    g.reset(max_factors=99)
    assert g.version > v and len(g.ii) == 0 and len(g.ii_inactive) == 0 and g.max_factors == 99
    g.add([1], [2])
    assert g.version > v + 1


def test_eviction_is_sort_independent_when_ages_are_unique():
    """visual_frontend.py:826-828: the edges that leave in add_factors are a positional mask through argsort(age).  With
    unique ages the permutation is unique, so any sorting routine (CPU torch as in the golden, a stable numpy sort, the
    device sort of the live product) removes the SAME edges; with tied ages the routine's tie order decides (documented in
    nerfslam/factor_graph.py:add): here the tie case is shown to be routine dependent, the unique case not."""
    import torch
    from nerfslam.factor_graph import FactorGraph
    rng = np.random.default_rng(5)
    for trial in range(20):
        n_old, n_new, max_factors = 40, 12, 44
        ii = rng.integers(0, 30, n_old); jj = ii + 1 + rng.integers(0, 5, n_old)
        age = rng.permutation(n_old * 3)[:n_old]                         # unique
        masks = []
        for sorter in ("torch", "numpy_stable", "reversed_ties"):
            if sorter == "torch":
                fn = None
            elif sorter == "numpy_stable":
                fn = lambda a: np.argsort(np.asarray(a), kind="stable")
            else:
                fn = lambda a: np.lexsort((-np.arange(len(a)), np.asarray(a)))      # ties (there are none) broken the other way
            masks.append(go.add_factors_removal_mask(age.tolist(), n_new, max_factors, argsort=fn))
        assert all(np.array_equal(masks[0], m) for m in masks[1:])
        g = FactorGraph(max_factors=max_factors)
        g.ii, g.jj, g.age = ii.astype(np.int64), jj.astype(np.int64), age.astype(np.int64)
        new_i = np.arange(100, 100 + n_new); new_j = new_i + 1
        _, _, removed = g.add(new_i, new_j, remove=True)
        assert np.array_equal(removed, masks[0])
    # tied ages: the positional mask depends on the tie order of the routine
    age = np.array([6, 6, 6, 6, 2, 2, 2, 2, 0, 0])
    a = go.add_factors_removal_mask(age.tolist(), 4, 10, argsort=lambda x: np.argsort(np.asarray(x), kind="stable"))
    b = go.add_factors_removal_mask(age.tolist(), 4, 10, argsort=lambda x: np.lexsort((-np.arange(len(x)), np.asarray(x))))
    assert a.sum() == b.sum() == 4 and not np.array_equal(a, b)
