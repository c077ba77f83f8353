#!/usr/bin/env python3
"""Where the tracker's time goes on the bench stream, by phase: host time to ISSUE a phase (no synchronisation) and, in a second
pass over the following frames, its time with a device synchronisation behind it.  usage: python tools/track_phases.py [K]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-slam_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    pipe = bench.Pipeline(dev, 100 + 2 * K + 32, 64, fusion=False)
    while not pipe.tracker.is_initialized:
        pipe.frame()
    for _ in range(5):
        pipe.frame()
    torch.cuda.synchronize()
    tr = pipe.tracker
    fe = tr.fe
    acc = collections.defaultdict(lambda: [0, 0.0])
    mode = {"sync": False}

    def wrap(obj, name, label=None):
        f = getattr(obj, name)
        label = label or name

        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            if mode["sync"]:
                torch.cuda.synchronize()
            e = acc[(label, mode["sync"])]
            e[0] += 1
            e[1] += time.perf_counter() - t0
            return r
        for attr in ("host_indices", "__self__"):        # (what TrackingFrontend.update() asks its update_op callable for)
            if hasattr(f, attr):
                try:
                    setattr(g, attr, getattr(f, attr))
                except AttributeError:
                    pass
        setattr(obj, name, g)

    wrap(tr.net, "features", "feature encoder")
    wrap(tr, "_enough_motion", "motion filter (incl. its read-back)")
    wrap(tr, "_store", "store keyframe (+ context encoder)")
    wrap(fe, "update", "update()")
    wrap(fe, "add_proximity_factors", "proximity factors (read-back + host graph + volumes)")
    wrap(fe, "rm_factors", "rm_factors")
    wrap(fe, "distance", "keyframe distance test (read-back)")
    wrap(fe, "get_viz_out", "viz packet")
    wrap(tr, "rm_keyframe", "rm_keyframe")
    # inside update()
    wrap(fe, "reproject", "  update: reproject")
    wrap(fe, "motion_features", "  update: motion features")
    wrap(fe, "ba", "  update: ba (2 iterations + covariances)")
    wrap(fe, "upsample", "  update: upsample")
    wrap(fe, "_edges", "  update: _edges (cached index tensors)")
    fe.update_op = tr.net.update          # (bound method looked up per call: wrap the object the frontend calls)
    wrap(fe, "update_op", "  update: update operator (droid_nets.update)")
    import nerfslam.corr as _corr
    wrap(_corr.CorrPool, "lookup_encoded", "  update: lookup + correlation encoder")
    from nerfslam import ba_plan as _bp
    wrap(_bp, "reduced_camera_matrix", "    ba: reduced_camera_matrix")
    wrap(_bp, "ba_solve", "    ba: ba_solve")
    wrap(_bp, "solve_depth", "    ba: solve_depth")
    wrap(_bp, "depth_cov", "    ba: depth_cov")
    wrap(_bp, "BaPlan", "    ba: BaPlan (host plan + upload)")
    res = {}
    for sync in (False, True):
        mode["sync"] = sync
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            pipe.frame()
        torch.cuda.synchronize()
        res[sync] = 1e3 * (time.perf_counter() - t0) / K
    print(f"ms per frame: issue-only pass {res[False]:.3f}, synchronised pass {res[True]:.3f}")
    for label in sorted({k[0] for k in acc}):
        a, b = acc[(label, False)], acc[(label, True)]
        print(f"  {label:58s} issue: {a[0]:4d} x {1e3 * a[1] / max(a[0], 1):7.3f} ms = {1e3 * a[1] / K:6.3f} ms/frame | "
              f"synchronised: {b[0]:4d} x {1e3 * b[1] / max(b[0], 1):7.3f} ms = {1e3 * b[1] / K:6.3f} ms/frame")
    pipe.close()


if __name__ == "__main__":
    main()
