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


def add_factors_removal_mask(age, n_new, max_factors, argsort=None):
    """visual_frontend.py:821-828: when old + new > max_factors (and volumes exist and remove=True) the
    mask passed to rm_factors is `ix >= max_factors - new` with ix = arange(n)[argsort(age)] -- a
    positional mask through the age permutation.  Returns the boolean mask over the OLD edges or None.

    The permutation is `torch.argsort(self.age)` in the reference (:826), an UNSTABLE sort applied to ages that tie
    (all edges of one add_factors call share an age): the tie order of that routine decides which edges are dropped.
    Replaying the reference's own methods (tests/golden/factor_graph_sequences.json.gz) showed a stable argsort does
    NOT reproduce it, so the default here is the same torch call (CPU torch, what generated the golden)."""
    n_old = len(age)
    if max_factors > 0 and n_old + n_new > max_factors:
        if argsort is None:
            import torch
            argsort = lambda a: torch.argsort(torch.as_tensor(np.asarray(a, np.int64))).numpy()
        ix = np.arange(n_old)[argsort(age)]
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


class GraphReplay:
    """Plain-list restatement of the CALL ORDER of the reference's keyframe loop, composed from the functions above:
    `forward` (:322-336, :363), `__initialize` (:641-688), `__update` (:577-638), `add_factors` (:806-833, :856-862),
    `rm_factors` (:868-892), `rm_keyframe` (:530-574).  Distances are REPLAYED from a log (tests/golden/
    factor_graph_sequences.json.gz, produced by the reference's own methods: tools/gen_golden_graph.py), every edge's
    payload is a serial number handed out in add order (what the golden's mock `reproject` does)."""

    def __init__(self, seq):
        self.seq = seq
        self.max_factors, self.max_age, self.stereo = seq["max_factors"], seq["max_age"], seq["stereo"]
        self.ii, self.jj, self.age, self.payload = [], [], [], []
        self.ii_in, self.jj_in, self.payload_in = [], [], []
        self.kf_idx, self.serial, self.calls, self.have_volumes = 0, 0, iter(seq["distance_calls"]), False
        self.slots = [0] * (seq["buffer"] + 1)
        self.initialized = False
        self.events = []

    # -- primitives ---------------------------------------------------------------------------------
    def distance(self, ii, jj):
        c = next(self.calls)
        assert c["ii"] == [int(v) for v in ii] and c["jj"] == [int(v) for v in jj], "candidate grid differs from the reference's"
        return np.asarray(c["d"], np.float32)

    def rm_factors(self, mask, store):
        keep = [not m for m in mask]
        if store:
            for k, m in enumerate(mask):
                if m:
                    self.ii_in.append(self.ii[k]); self.jj_in.append(self.jj[k]); self.payload_in.append(self.payload[k])
        f = lambda a: [v for v, k in zip(a, keep) if k]
        self.ii, self.jj, self.age, self.payload = f(self.ii), f(self.jj), f(self.age), f(self.payload)

    def add_factors(self, es, remove):
        ii, jj = [e[0] for e in es], [e[1] for e in es]
        keep = filter_repeated_edges(ii, jj, list(zip(self.ii, self.jj)), list(zip(self.ii_in, self.jj_in)))
        ii, jj = [v for v, k in zip(ii, keep) if k], [v for v, k in zip(jj, keep) if k]
        if not ii:
            return
        if self.have_volumes and remove:
            mask = add_factors_removal_mask(self.age, len(ii), self.max_factors)
            if mask is not None:
                self.rm_factors(list(mask), store=True)
        self.ii += ii; self.jj += jj; self.age += [0] * len(ii)
        self.payload += list(range(self.serial, self.serial + len(ii)))
        self.serial += len(ii)
        self.have_volumes = True

    def proximity(self, kf0, kf1, rad, nms, thresh, remove):
        t = self.kf_idx + 1
        gi = np.repeat(np.arange(kf0, t), t - kf1); gj = np.tile(np.arange(kf1, t), t - kf0)
        d = self.distance(gi, gj)
        existing = list(zip(self.ii, self.jj)) + list(zip(self.ii_in, self.jj_in))      # ii_bad is never filled (:233-234)
        es = proximity_factors(d, existing, self.kf_idx, kf0, kf1, rad, nms, thresh, self.max_factors, self.stereo)
        self.add_factors(es, remove)

    def updates(self, n):
        self.age = [a + n for a in self.age]

    # -- the keyframe loop --------------------------------------------------------------------------
    def snapshot(self, what, accepted):
        self.events.append(dict(what=what, kf_idx=self.kf_idx, accepted=accepted, ii=list(self.ii), jj=list(self.jj),
                                age=list(self.age), ii_inactive=list(self.ii_in), jj_inactive=list(self.jj_in),
                                payload=list(self.payload), payload_inactive=list(self.payload_in),
                                slot_frame_ids=self.slots[:self.kf_idx + 1]))

    def initialize(self):
        ni, nj = neighborhood_factors(0, self.kf_idx, 3, self.stereo)
        self.add_factors(list(zip(ni.tolist(), nj.tolist())), False)
        self.updates(8)
        self.proximity(0, 0, 2, 2, 16.0, False)
        self.updates(8)
        self.initialized = True
        self.rm_factors([i < 8 - 4 for i in self.ii], store=True)                      # :687

    def update(self):
        k = self.kf_idx
        if self.have_volumes:
            self.rm_factors([a > self.max_age for a in self.age], store=True)
        self.proximity(k - 4, max(k + 1 - 25, 0), 2, 1, 16.0, True)
        self.updates(4)
        if float(self.distance([k - 2], [k - 1])[0]) < 4.0:
            return False
        self.updates(2)
        return True

    def rm_keyframe(self, k):
        self.slots[k] = self.slots[k + 1]
        keep, ni, nj = rm_keyframe_edges(self.ii_in, self.jj_in, k)
        self.ii_in, self.jj_in = ni.tolist(), nj.tolist()
        self.payload_in = [p for p, m in zip(self.payload_in, keep) if m]
        keep, ni, nj = rm_keyframe_edges(self.ii, self.jj, k)
        self.ii, self.jj = ni.tolist(), nj.tolist()
        self.age = [a for a, m in zip(self.age, keep) if m]
        self.payload = [p for p, m in zip(self.payload, keep) if m]

    def run(self):
        frame, self.kf_idx = 0, 1
        while self.kf_idx < self.seq["buffer"] - 1:
            frame += 1
            self.slots[self.kf_idx] = frame
            if not self.initialized:
                if self.kf_idx >= 8:
                    self.initialize()
                    self.snapshot("initialize", True)
            else:
                if not self.update():
                    self.snapshot("update", False)
                    self.rm_keyframe(self.kf_idx - 1)
                    self.snapshot("rm_keyframe", False)
                    continue
                self.snapshot("update", True)
            self.kf_idx += 1
        return self.events
