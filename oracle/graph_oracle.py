"""CPU ORACLE of the factor-graph bookkeeping (test infrastructure, NOT product code).

Plain-Python restatement of the host loops that decide the edge lists `ii / jj / age` in
/root/reference/slam/visual_frontends/visual_frontend.py (lines cited per function).  "Identical
factor-graph indices" is a graded criterion of the north star; the product implementation
(nerf-slam_amd/nerfslam/factor_graph.py) is written differently (vectorised numpy on a host mirror of the
edge lists) and is tested against this file on random inputs.
"""
import numpy as np


def neighborhood_factors(kf0, kf1, radius, stereo=False):
    """visual_frontend.py:690-708 -> (ii, jj) in meshgrid (row-major, ii slow) order."""
    c = 1 if stereo else 0
    ii, jj = [], []
    for i in range(kf0, kf1 + 1):
        for j in range(kf0, kf1 + 1):
            if c < abs(i - j) <= radius:
                ii.append(i)
                jj.append(j)
    return np.array(ii, np.int64), np.array(jj, np.int64)


def proximity_factors(d, existing, kf_idx, kf0, kf1, rad, nms, thresh, max_factors, stereo=False):
    """visual_frontend.py:712-775.  `d` = bidirectional frame distances over the candidate grid
    ii in [kf0, t) x jj in [kf1, t), t = kf_idx + 1, flattened row-major (ii slow); it is modified
    the way the reference modifies it.  `existing` = list of (i, j) over active + bad + inactive edges.
    Returns the ordered edge list `es` handed to add_factors."""
    t = kf_idx + 1
    d = np.array(d, np.float32).copy()
    nj = t - kf1
    ii = np.repeat(np.arange(kf0, t), nj)
    jj = np.tile(np.arange(kf1, t), t - kf0)
    d[(ii - rad) < jj] = np.inf                                   # :724
    d[d > 100] = np.inf                                           # :725
    for (i, j) in existing:                                       # :729-737
        for di in range(-nms, nms + 1):
            for dj in range(-nms, nms + 1):
                if abs(di) + abs(dj) <= max(min(abs(i - j) - 2, nms), 0):
                    i1, j1 = i + di, j + dj
                    if (kf0 <= i1 < t) and (kf1 <= j1 < t):
                        d[(i1 - kf0) * nj + (j1 - kf1)] = np.inf
    es = []
    for i in range(kf0, t):                                       # :740-748
        if stereo:
            es.append((i, i))
            d[(i - kf0) * nj + (i - kf1)] = np.inf
        for j in range(max(i - rad - 1, 0), i):
            es.append((i, j))
            es.append((j, i))
            d[(i - kf0) * nj + (j - kf1)] = np.inf
    ix = np.argsort(d, kind="stable")                             # :750 (torch.argsort is not stable; ties are inf only)
    for k in ix:                                                  # :751-772
        if d[k] > thresh:
            continue
        if len(es) > max_factors:
            break
        i, j = int(ii[k]), int(jj[k])
        es.append((i, j))
        es.append((j, i))
        for di in range(-nms, nms + 1):
            for dj in range(-nms, nms + 1):
                if abs(di) + abs(dj) <= max(min(abs(i - j) - 2, nms), 0):
                    i1, j1 = i + di, j + dj
                    if (kf0 <= i1 < t) and (kf1 <= j1 < t):
                        d[(i1 - kf0) * nj + (j1 - kf1)] = np.inf
    return es


def filter_repeated_edges(ii, jj, active, inactive):
    """visual_frontend.py:896-907 -> keep mask."""
    eset = set(active) | set(inactive)
    return np.array([(int(i), int(j)) not in eset for i, j in zip(ii, jj)], bool)


def add_factors_removal_mask(age, n_new, max_factors):
    """visual_frontend.py:821-828: when old + new > max_factors (and volumes exist and remove=True) the
    mask passed to rm_factors is `ix >= max_factors - new` with ix = arange(n)[argsort(age)] -- a
    positional mask through the age permutation.  Returns the boolean mask over the OLD edges or None."""
    n_old = len(age)
    if max_factors > 0 and n_old + n_new > max_factors:
        ix = np.arange(n_old)[np.argsort(np.asarray(age), kind="stable")]
        return ix >= (max_factors - n_new)
    return None


def rm_keyframe_edges(ii, jj, k):
    """visual_frontend.py:552-574 for one edge list: edges touching k are dropped, indices >= k shift down.
    Returns (keep_mask, new_ii, new_jj).  (The reference decrements before masking; the mask is taken on
    the original indices.)"""
    ii, jj = np.asarray(ii, np.int64), np.asarray(jj, np.int64)
    drop = (ii == k) | (jj == k)
    ni, nj = ii.copy(), jj.copy()
    ni[ni >= k] -= 1
    nj[nj >= k] -= 1
    return ~drop, ni[~drop], nj[~drop]
