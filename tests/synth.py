"""Seeded synthetic inputs for the tracking hot path; numpy only."""
import numpy as np


def quat_exp(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.array([0, 0, 0, 1.0])
    return np.concatenate([np.sin(th / 2) * w / th, [np.cos(th / 2)]])


def make_graph(P, M, rng, kf0=0, extra_fixed=0):
    """Edges over frames [kf0-extra_fixed, kf0+P): radius-3 chain in both directions, then random
    pairs (no duplicates, no self loops) up to M edges."""
    lo, hi = kf0 - extra_fixed, kf0 + P
    assert lo >= 0, "frames below 0"
    es = []
    for i in range(lo, hi):
        for j in range(max(lo, i - 3), min(hi, i + 4)):
            if i != j:
                es.append((i, j))
    seen = set(es)
    M = min(M, (hi - lo) * (hi - lo - 1))
    while len(es) < M:
        i, j = rng.integers(lo, hi, 2)
        if i != j and (i, j) not in seen:
            seen.add((i, j))
            es.append((int(i), int(j)))
    es = es[:M] if len(es) > M else es
    perm = rng.permutation(len(es))
    ii = np.array([es[k][0] for k in perm], np.int64)
    jj = np.array([es[k][1] for k in perm], np.int64)
    return ii, jj


def make_problem(ht=12, wd=16, P=5, M=24, seed=0, kf0=0, extra_fixed=0, nbuf=None, sensed_frac=0.0,
                 trans_sigma=0.05, rot_sigma=0.02, noise=0.5):
    rng = np.random.default_rng(seed)
    nbuf = nbuf or (kf0 + P + 1)
    HW = ht * wd
    poses = np.zeros((nbuf, 7), np.float32)
    poses[:, 6] = 1
    for k in range(nbuf):
        poses[k, :3] = rng.normal(0, trans_sigma, 3)
        poses[k, 3:] = quat_exp(rng.normal(0, rot_sigma, 3))
    disps = rng.uniform(0.2, 2.0, (nbuf, ht, wd)).astype(np.float32)
    disps_sens = np.zeros_like(disps)
    if sensed_frac > 0:
        m = rng.uniform(size=disps.shape) < sensed_frac
        disps_sens[m] = (disps[m] * rng.uniform(0.9, 1.1, m.sum())).astype(np.float32)
    W = wd * 8.0
    intr = np.array([0.5 * W, 0.5 * W, (W - 1) / 2, (ht * 8.0 - 1) / 2], np.float32) / 8.0
    extr = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
    ii, jj = make_graph(P, M, rng, kf0, extra_fixed)
    M = ii.shape[0]
    # targets = reprojection + noise (computed with the oracle-independent float64 formulas)
    targets = np.zeros((M, 2, ht, wd), np.float32)
    gy, gx = np.meshgrid(np.arange(ht), np.arange(wd), indexing="ij")
    for e in range(M):
        pi, pj = poses[ii[e]].astype(np.float64), poses[jj[e]].astype(np.float64)
        X = np.stack([(gx - intr[2]) / intr[0], (gy - intr[3]) / intr[1], np.ones_like(gx, float)], -1)
        d = disps[ii[e]].astype(np.float64)
        Ri, Rj = quat_to_R(pi[3:]), quat_to_R(pj[3:])
        # world point (scaled by inverse depth): Xw = Ri^T (X - d ti)
        Xw = (X - d[..., None] * pi[:3]) @ Ri
        Xj = Xw @ Rj.T + d[..., None] * pj[:3]
        z = np.maximum(Xj[..., 2], 0.25)
        targets[e, 0] = intr[0] * Xj[..., 0] / z + intr[2] + rng.normal(0, noise, (ht, wd))
        targets[e, 1] = intr[1] * Xj[..., 1] / z + intr[3] + rng.normal(0, noise, (ht, wd))
    weights = rng.uniform(0, 1, (M, 2, ht, wd)).astype(np.float32)
    kf1 = kf0 + P
    K = np.unique(np.concatenate([np.arange(kf0, kf1), ii])).shape[0]
    eta = (0.2 * rng.uniform(1e-4, 2e-2, (K, ht, wd)) + 1e-7).astype(np.float32)
    return dict(poses=poses, disps=disps, disps_sens=disps_sens, intr=intr, extr=extr, ii=ii, jj=jj,
                targets=targets, weights=weights, eta=eta, kf0=kf0, kf1=kf1, ht=ht, wd=wd, HW=HW, K=K)


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def lookup_inputs(E=3, ht=12, wd=16, seed=0, dtype=np.float16, spread=8.0, oob_frac=0.05, levels=4):
    rng = np.random.default_rng(seed)
    pyr = [rng.standard_normal((E, ht, wd, ht >> l, wd >> l)).astype(dtype) for l in range(levels)]
    gy, gx = np.meshgrid(np.arange(ht), np.arange(wd), indexing="ij")
    coords = np.stack([gx, gy], -1)[None].repeat(E, 0).astype(np.float32)
    coords += rng.uniform(-spread, spread, coords.shape).astype(np.float32)
    oob = rng.uniform(size=(E, ht, wd)) < oob_frac
    coords[oob] += rng.choice([-1.0, 1.0], size=(int(oob.sum()), 2)).astype(np.float32) * (wd + ht)
    return pyr, coords
