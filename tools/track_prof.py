#!/usr/bin/env python3
"""Tracker alone (no mapper) on the bench's synthetic stream: K timed frames after initialisation; prints ms per frame and, with
--legs, the time of the per-frame path (feature net + motion filter) and of the keyframe path separately.
usage: python tools/track_prof.py [K]        (run under rocprofv3 --kernel-trace --stats for the per-kernel table)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-slam_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    n_frames = 100 + K + 16
    pipe = bench.Pipeline(dev, n_frames, 64, fusion=False)
    while not pipe.tracker.is_initialized:
        pipe.frame()
    for _ in range(5):
        pipe.frame()
    torch.cuda.synchronize()
    st0 = dict(pipe.tracker.stats); up0 = pipe.tracker.fe.n_updates
    per = []
    t0 = time.perf_counter()
    for _ in range(K):
        a = time.perf_counter()
        c0 = pipe.tracker.stats["candidates"]
        pipe.frame()
        torch.cuda.synchronize()
        per.append((1e3 * (time.perf_counter() - a), pipe.tracker.stats["candidates"] - c0))
    dt = time.perf_counter() - t0
    st1 = pipe.tracker.stats
    nk = [p for p, c in per if c == 0]
    kf = [p for p, c in per if c > 0]
    print(f"{K} frames: {1e3 * dt / K:.3f} ms/frame (sync per frame); candidates {st1['candidates'] - st0['candidates']}, "
          f"rejected {st1['rejected'] - st0['rejected']}, updates {pipe.tracker.fe.n_updates - up0}")
    print(f"  non-keyframe frames: {len(nk)} x {sum(nk) / max(len(nk), 1):.3f} ms;  keyframe-candidate frames: {len(kf)} x "
          f"{sum(kf) / max(len(kf), 1):.3f} ms")
    pipe.close()


if __name__ == "__main__":
    main()
