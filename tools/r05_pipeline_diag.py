"""Who bounds the --parallel_run pipeline on one GPU: the tracker or the mapper?  (round 5 diagnostic, NOT a benchmark.)

bench.py's parallel mode takes 7.5 ms per frame where the tracking leg alone takes 3.4 ms and the mapping leg 4.9 ms, and the queue
depth between them changes nothing (profiles/r05_ab_records.json).  This tool re-times the same pipeline with the mapper's
optimiser steps per frame varied (16 is the product's; the others make the numbers INVALID as results and are only there for the
slope): if the frame time follows the mapper's step count the mapper is the critical leg; if it barely moves, the tracker -- whose
~130 small dependent launches per frame each wait for a slot on a GPU the mapper's graph keeps full -- is.
usage: python tools/r05_pipeline_diag.py [frames per window]   -> one JSON line per setting"""
import json
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "nerf-slam_amd"), os.path.join(root, "tools")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch

import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
settings = [16, 8, 32, 2, 16]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
torch.set_grad_enabled(False)
n_frames = 100 + 5 + (3 * len(settings) + 2) * K + 8
pipe = bench.Pipeline(dev, n_frames, max(32, min(512, n_frames // 3 + 16)), fusion=True)
ngp = pipe.fusion.fusion.ngp
while not pipe.tracker.is_initialized:
    pipe.frame()
pipe.parallel = True
for _ in range(5):
    pipe.frame()
for spf in settings:
    pipe.drain()
    ngp.steps_per_frame = spf
    ws = []
    for _ in range(3):
        pipe.drain()
        s0, c0 = int(ngp.training_step), pipe.tracker.stats["candidates"]
        t0 = time.perf_counter()
        for _ in range(K):
            pipe.frame()
        t_track = time.perf_counter() - t0           # the tracker thread has ISSUED its K frames (it blocks on the bounded queue)
        pipe.drain()
        dt = time.perf_counter() - t0
        ws.append({"ms_per_frame": 1e3 * dt / K, "tracker_issue_ms_per_frame": 1e3 * t_track / K,
                   "steps": int(ngp.training_step) - s0, "candidates": pipe.tracker.stats["candidates"] - c0})
    ws.sort(key=lambda w: w["ms_per_frame"])
    print(json.dumps({"mapper_steps_per_frame": spf, "median": ws[1], "all_ms_per_frame": [round(w["ms_per_frame"], 3) for w in ws]}), flush=True)
# the tracker alone on its stream in the parallel mode's setting (mapper thread idle: every frame would be a packet-less spin of 0 steps)
ngp.shall_train = False
pipe.drain()
ws = []
for _ in range(3):
    pipe.drain()
    t0 = time.perf_counter()
    for _ in range(K):
        pipe.frame()
    pipe.drain()
    ws.append(1e3 * (time.perf_counter() - t0) / K)
print(json.dumps({"mapper_steps_per_frame": 0, "tracker_only_ms_per_frame": sorted(ws)}), flush=True)
pipe.close()
