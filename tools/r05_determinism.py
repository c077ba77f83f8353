"""Is the NeRF optimiser step bit-reproducible from run to run?  Two trainers, same seed, same scene, `n` steps each; compares the
hash-grid master, the MLP master and the poses bit for bit after 1, 2, 4, ... steps, and the sample arrays of the first step.
usage: python tools/r05_determinism.py [steps]"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd"), os.path.join(root, "tools")]
import torch

from nerfslam.ngp import NgpConfig, NgpNerf
from ngp_scene import sphere_scene

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene = sphere_scene()


def run(k):
    net = NgpNerf(NgpConfig(optimize_extrinsics=True), dev, seed=0)
    net.set_images(*scene)
    snaps = {}
    for s in range(1, k + 1):
        net.train_step(return_loss=False)
        if s & (s - 1) == 0:
            torch.cuda.synchronize()
            X = net.sets[1 - net.cur]
            snaps[s] = dict(grid=net.grid_master.clone(), mlp=net.mlp_master.clone(), c2w=net.c2w.clone(),
                            n=int(X["counter"][2].item()), pos=X["s_pos"].clone(), start=X["ray_start"].clone())
    return snaps


a, b = run(n), run(n)
out = {}
for s in sorted(a):
    out[s] = {k: bool(torch.equal(a[s][k], b[s][k])) for k in ("grid", "mlp", "c2w", "pos", "start")}
    out[s]["samples"] = (a[s]["n"], b[s]["n"])
    if not out[s]["mlp"]:
        out[s]["mlp_max_abs_diff"] = float((a[s]["mlp"] - b[s]["mlp"]).abs().max())
    if not out[s]["grid"]:
        out[s]["grid_entries_differing"] = int((a[s]["grid"] != b[s]["grid"]).sum())
print(json.dumps(out))
