#!/usr/bin/env python3
"""Generate tests/golden/factor_graph_sequences.json.gz by RUNNING THE REFERENCE'S OWN factor-graph methods.

`north_star` grades "identical factor-graph indices".  This script imports
/root/reference/slam/visual_frontends/visual_frontend.py (read-only) behind import-time stand-ins for its
un-vendored third-party modules, builds a `RaftVisualFrontend` object WITHOUT running its constructor (no
weights, no gtsam), lets the reference's own `initialize_buffers` allocate the state, and then drives the
reference's OWN bound methods

    __initialize (:641-688)   __update (:577-638)   rm_keyframe (:530-574)
    add_neighborhood_factors (:690-708)   add_proximity_factors (:712-775)
    add_factors (:806-862)   rm_factors (:868-892)   __filter_repeated_edges (:896-907)

through the frame loop of `forward` (:282-365, whose three bookkeeping lines are the only thing restated
here: call __initialize at warm-up, call __update afterwards, rm_keyframe on rejection, kf_idx += 1 otherwise).

Only two things are replaced on the instance, because they need the GPU extension / the learned networks:
  * `distance(ii, jj, ...)`  -> seeded pseudo-distances (logged, so a test can replay the SAME numbers)
  * `update(...)`            -> `self.age += 1` (the one graph-visible effect of update(), :465)
  * `reproject(ii, jj)`      -> a target tensor whose every entry is a fresh per-edge serial number, so the
                                payload permutations (active -> inactive moves, removals) are pinned too.
Everything else that touches ii / jj / age / ii_inactive / jj_inactive is the reference's code, including the
real `CorrBlock` (corr.py:23-38) being concatenated / masked as the reference does (on tiny CPU feature maps).

Run in the build container only (`python tools/gen_golden_graph.py`); the JSON it writes is committed.
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tools"))


class _Anything(types.ModuleType):
    """module stand-in: any attribute is a dummy class (import-time `from gtsam import X` needs only)"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {"__init__": lambda self, *a, **k: None})


def install_stubs():
    import gen_golden
    gen_golden.install_stubs()                       # lietorch / icecream / droid_backends, sys.path += REF
    for name in ("gtsam", "gtsam.symbol_shorthand", "cv2", "torch_scatter"):
        sys.modules.setdefault(name, _Anything(name))
    sys.modules["gtsam"].symbol_shorthand = sys.modules["gtsam.symbol_shorthand"]


class _FakePose:
    """what gtsam_pose_to_torch (:46-49) reads"""

    def translation(self):
        return np.zeros(3)

    def rotation(self):
        return types.SimpleNamespace(quaternion=lambda: np.array([1.0, 0, 0, 0]))


def make_frontend(RaftVisualFrontend, buffer, stereo, max_factors, max_age):
    fe = RaftVisualFrontend.__new__(RaftVisualFrontend)
    torch.nn.Module.__init__(fe)
    # the constants of RaftVisualFrontend.__init__ (:67-131) that the bookkeeping reads
    fe.kf_idx, fe.buffer, fe.stereo, fe.device = 0, buffer, stereo, "cpu"
    fe.is_initialized = False
    fe.keyframe_warmup, fe.max_age, fe.max_factors = 8, max_age, max_factors
    fe.keyframe_thresh, fe.frontend_thresh, fe.frontend_window = 4.0, 16.0, 25
    fe.frontend_radius, fe.frontend_nms, fe.beta = 2, 1, 0.3
    fe.iters1, fe.iters2, fe.dsf, fe.corr_impl, fe.viz = 4, 2, 8, "volume", False
    fe.world_T_body_t0 = _FakePose()
    fe.cam0_t0_T_world = torch.tensor([0, 0, 0, 0, 0, 0, 1.0])
    fe.g_prior_cov = torch.eye(6) * 1e-4
    fe.idepth_prior_cov = torch.tensor(0.01)
    fe.initialize_buffers((128, 128))                # REFERENCE code (:162-237): 16x16 feature maps (4 poolings)
    return fe


def run_sequence(RaftVisualFrontend, seed, buffer, stereo, max_factors, max_age, reject_p):
    rng = np.random.default_rng(seed)
    fe = make_frontend(RaftVisualFrontend, buffer, stereo, max_factors, max_age)
    cls = "_RaftVisualFrontend"
    fe.features_imgs[:] = torch.from_numpy(rng.standard_normal(tuple(fe.features_imgs.shape)).astype(np.float16))
    log = {"distance_calls": [], "events": []}
    serial = [0]

    def distance(ii=None, jj=None, beta=0.3, bidirectional=True):
        ii = np.asarray(torch.as_tensor(ii).reshape(-1).tolist(), np.int64)
        jj = np.asarray(torch.as_tensor(jj).reshape(-1).tolist(), np.int64)
        if ii.shape[0] == 1:                         # the keyframe test of __update (:611-615)
            d = np.array([1.0 if rng.uniform() < reject_p else 9.0], np.float32)
        else:                                        # proximity candidates: grows with |i - j|, some > 100 (-> inf, :725)
            d = (np.abs(ii - jj) * rng.uniform(1.0, 7.0, ii.shape[0])).astype(np.float32)
            d[rng.uniform(size=d.shape) < 0.05] = 250.0
            # (no two finite distances are equal, as with real float distances: `torch.argsort(d)` (:750) is not a
            #  stable sort, so equal finite distances would pin torch's tie order instead of the reference's logic)
        log["distance_calls"].append({"ii": ii.tolist(), "jj": jj.tolist(), "d": d.tolist()})
        return torch.from_numpy(d.copy())

    def update(kf0=None, kf1=None, itrs=2, use_inactive=False, EP=1e-7, motion_only=False):
        fe.age += 1                                   # visual_frontend.py:465
        return None, None

    def reproject(ii, jj, cam_T_body=None, jacobian=False):
        n = ii.shape[0]
        ids = torch.arange(serial[0], serial[0] + n, dtype=torch.float32)
        serial[0] += n
        return ids.view(1, n, 1, 1, 1).expand(1, n, fe.ht, fe.wd, 2).clone(), None, None

    fe.distance, fe.update, fe.reproject = distance, update, reproject

    def snapshot(what, accepted=None):
        vol = fe.correlation_volumes
        ev = {"what": what, "kf_idx": int(fe.kf_idx), "accepted": accepted,
              "ii": fe.ii.tolist(), "jj": fe.jj.tolist(), "age": fe.age.tolist(),
              "ii_inactive": fe.ii_inactive.tolist(), "jj_inactive": fe.jj_inactive.tolist(),
              "ii_bad": fe.ii_bad.tolist(), "jj_bad": fe.jj_bad.tolist(),
              "payload": fe.gru_estimated_flow[0, :, 0, 0, 0].long().tolist(),
              "payload_inactive": fe.gru_estimated_flow_inactive[0, :, 0, 0, 0].long().tolist(),
              "n_volumes": 0 if vol is None else int(vol.corr_pyramid[0].shape[0])}
        # slot ids: every per-keyframe buffer is moved by rm_keyframe; timestamps carry the frame id
        ev["slot_frame_ids"] = fe.cam0_timestamps[:fe.kf_idx + 2].long().tolist()
        log["events"].append(ev)

    frame = 0
    fe.cam0_timestamps[0] = frame
    fe.kf_idx = 1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while fe.kf_idx < buffer - 1:
            frame += 1
            fe.cam0_timestamps[fe.kf_idx] = frame     # forward() :309-318 stores the candidate in slot kf_idx
            if not fe.is_initialized:
                if fe.kf_idx >= fe.keyframe_warmup:
                    getattr(fe, cls + "__initialize")()
                    snapshot("initialize", True)
            else:
                ok = getattr(fe, cls + "__update")()
                if not ok:
                    snapshot("update", False)
                    fe.rm_keyframe(fe.kf_idx - 1)    # :328-336
                    snapshot("rm_keyframe", False)
                    continue
                snapshot("update", True)
            fe.kf_idx += 1
    return {"seed": seed, "buffer": buffer, "stereo": stereo, "max_factors": max_factors, "max_age": max_age, **log}


def main():
    install_stubs()
    from slam.visual_frontends.visual_frontend import RaftVisualFrontend   # REFERENCE code
    cases = [dict(seed=1, buffer=28, stereo=False, max_factors=48, max_age=25, reject_p=0.25),
             dict(seed=2, buffer=40, stereo=False, max_factors=48, max_age=25, reject_p=0.35),
             dict(seed=3, buffer=34, stereo=False, max_factors=24, max_age=6, reject_p=0.2),
             dict(seed=4, buffer=30, stereo=False, max_factors=36, max_age=10, reject_p=0.5),
             dict(seed=5, buffer=24, stereo=True, max_factors=48, max_age=25, reject_p=0.25),
             dict(seed=6, buffer=48, stereo=False, max_factors=48, max_age=25, reject_p=0.0)]
    seqs = [run_sequence(RaftVisualFrontend, **c) for c in cases]
    import gzip
    out = os.path.join(ROOT, "tests", "golden", "factor_graph_sequences.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps({"generator": "tools/gen_golden_graph.py", "sequences": seqs}, separators=(",", ":")).encode())
    for s in seqs:
        ev = s["events"]
        print(f"seed {s['seed']}: {len(ev)} events, {len(s['distance_calls'])} distance calls, final E={len(ev[-1]['ii'])} "
              f"inactive={len(ev[-1]['ii_inactive'])} rejected={sum(1 for e in ev if e['what'] == 'rm_keyframe')}")
    print(out, os.path.getsize(out))


if __name__ == "__main__":
    main()
