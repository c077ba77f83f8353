"""Index-for-index parity of the factor-graph bookkeeping with THE REFERENCE'S OWN METHODS.

tests/golden/factor_graph_sequences.json.gz was produced by tools/gen_golden_graph.py, which imports
/root/reference/slam/visual_frontends/visual_frontend.py and runs its own `__initialize`, `__update`, `rm_keyframe`,
`add_proximity_factors`, `add_neighborhood_factors`, `add_factors`, `rm_factors`, `__filter_repeated_edges` (with seeded
frame distances, logged).  Here the same distance log is replayed through

  * oracle/graph_oracle.py (the CPU restatement)                            -> pins the oracle
  * the PRODUCT: nerfslam.slam.TrackingSLAM._initialize/_track/rm_keyframe driving nerfslam.frontend.TrackingFrontend
    (add_proximity_factors / add_factors / rm_factors / payload moves) and nerfslam.factor_graph.FactorGraph
    - on the host (device work replaced by id-carrying stand-ins)            [CPU suite]
    - on the GPU with the real correlation pool / keyframe buffers           [-m gpu]

and every event (one per keyframe candidate) must reproduce ii / jj / age / ii_inactive / jj_inactive, the payload
permutation of both lists and the keyframe-slot contents exactly.
"""
import gzip
import json
import os
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SEQS = json.load(gzip.open(os.path.join(HERE, "golden", "factor_graph_sequences.json.gz"), "rt"))["sequences"]
KEYS = ("what", "kf_idx", "accepted", "ii", "jj", "age", "ii_inactive", "jj_inactive", "payload", "payload_inactive")


def _compare(got, ref, label):
    assert len(got) == len(ref), f"{label}: {len(got)} events, the reference has {len(ref)}"
    for n, (g, r) in enumerate(zip(got, ref)):
        for k in KEYS:
            assert g[k] == r[k], f"{label}: event {n} ({r['what']}, kf_idx {r['kf_idx']}): `{k}` differs from the reference"
        m = r["kf_idx"] + 1
        assert list(g["slot_frame_ids"][:m]) == list(r["slot_frame_ids"][:m]), f"{label}: event {n}: keyframe slots differ"


@pytest.mark.parametrize("seq", SEQS, ids=lambda s: f"seed{s['seed']}")
def test_oracle_replay_matches_reference_methods(seq):
    from oracle.graph_oracle import GraphReplay
    _compare(GraphReplay(seq).run(), seq["events"], "oracle")


# ---------------------------------------------------------------------------------------------------------------------
class _FakePool:
    """stands in for nerfslam.corr.CorrPool on the host: slot bookkeeping only"""

    def __init__(self, capacity):
        self.capacity = capacity

    def grow(self, n):
        self.capacity = n

    def build(self, *a, **k):
        pass


def _product_replay(seq, device):
    """drive the PRODUCT's keyframe loop with the golden's distance log; device = 'cpu' replaces the HIP-backed pieces by
    stand-ins, a cuda device keeps the real correlation pool, keyframe buffers and payload tensors"""
    from nerfslam.frontend import TrackingFrontend
    from nerfslam.slam import TrackingSLAM
    dev = torch.device(device)
    host = dev.type == "cpu"
    calls = iter(seq["distance_calls"])
    serial = [0]

    class ReplayFrontend(TrackingFrontend):
        def distance(self, ii, jj, bidirectional=True):
            c = next(calls)
            assert c["ii"] == np.asarray(ii).reshape(-1).tolist() and c["jj"] == np.asarray(jj).reshape(-1).tolist(), \
                "candidate grid differs from the reference's"
            return torch.tensor(c["d"], dtype=torch.float32, device=dev)

        def update(self, itrs=2):
            self.graph.age += 1                      # the graph-visible effect of update() (visual_frontend.py:465)

        def reproject(self, ii, jj):                 # id-carrying targets, as in the golden generator
            n = ii.shape[0]
            ids = torch.arange(serial[0], serial[0] + n, dtype=torch.float32, device=dev)
            serial[0] += n
            return ids.view(n, 1, 1, 1).expand(n, self.ht, self.wd, 2).contiguous()

        if host:
            def _take_slots(self, n):
                if self.corr is None:
                    self.corr = _FakePool(max(self.graph.max_factors + 16, 2 * n))
                    self._free_slots = list(range(self.corr.capacity - 1, -1, -1))
                if len(self._free_slots) < n:
                    old = self.corr.capacity
                    self.corr.grow(max(2 * old, old + n))
                    self._free_slots = list(range(self.corr.capacity - 1, old - 1, -1)) + self._free_slots
                return np.asarray([self._free_slots.pop() for _ in range(n)], np.int32)

    args = types.SimpleNamespace(buffer=seq["buffer"], networks=types.SimpleNamespace(), slam=True, global_ba=False)
    slam = TrackingSLAM("replay", args, dev)
    fe = slam.fe = ReplayFrontend(seq["buffer"], 128, 128, [100.0, 100.0, 64.0, 64.0], device=dev,
                                  max_factors=seq["max_factors"])
    fe.max_age = seq["max_age"]
    # ages tie and the reference's torch.argsort (:826) is unstable: the golden holds CPU torch's tie order, so the age
    # permutation is sorted on the CPU here (the product's default on a GPU is the device sort, as in the reference)
    fe.graph.sort_device = "cpu"
    fe.graph.stereo = seq["stereo"]
    events = []

    def snapshot(what, accepted):
        g = fe.graph
        assert fe.target.shape[0] == g.ii.shape[0] == fe.slots.shape[0] and fe.target_inactive.shape[0] == g.ii_inactive.shape[0]
        assert len(set(fe.slots.tolist())) == fe.slots.shape[0], "two live edges share a correlation-pool slot"
        events.append(dict(what=what, kf_idx=fe.kf_idx, accepted=accepted, ii=g.ii.tolist(), jj=g.jj.tolist(), age=g.age.tolist(),
                           ii_inactive=g.ii_inactive.tolist(), jj_inactive=g.jj_inactive.tolist(),
                           payload=fe.target[:, 0, 0, 0].long().tolist(),
                           payload_inactive=fe.target_inactive[:, 0, 0, 0].long().tolist(),
                           slot_frame_ids=fe.images[:fe.kf_idx + 1, 0, 0, 0].long().tolist()))

    frame, fe.kf_idx = 0, 1
    while fe.kf_idx < seq["buffer"] - 1:            # the keyframe branch of TrackingSLAM._frontend (visual_frontend.py:322-363)
        frame += 1
        fe.images[fe.kf_idx] = frame
        if not slam.is_initialized:
            if fe.kf_idx >= slam.keyframe_warmup:
                slam._initialize()
                snapshot("initialize", True)
        elif not slam._track():
            snapshot("update", False)
            slam.rm_keyframe(fe.kf_idx - 1)
            snapshot("rm_keyframe", False)
            continue
        else:
            snapshot("update", True)
        fe.kf_idx += 1
    return events


@pytest.mark.parametrize("seq", SEQS, ids=lambda s: f"seed{s['seed']}")
def test_product_keyframe_loop_matches_reference_methods_host(seq):
    _compare(_product_replay(seq, "cpu"), seq["events"], "product (host)")


@pytest.mark.gpu
@pytest.mark.parametrize("seq", SEQS[:3] + SEQS[4:5], ids=lambda s: f"seed{s['seed']}")
def test_product_keyframe_loop_matches_reference_methods_gpu(seq):
    _compare(_product_replay(seq, "cuda:0"), seq["events"], "product (gpu)")
