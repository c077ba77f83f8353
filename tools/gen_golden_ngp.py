"""tests/golden/ngp_training.json: checksums of the NeRF trainer's state after a fixed number of optimiser steps on the
sphere scene -- a REGRESSION pin of this repository's own mapping arithmetic (round 5: the step is bit-reproducible), NOT a
parity pin against the instant-ngp fork, which is absent (mapping-path parity stays unpinned).  Must run on an MI355X:
    gpurun -- 'python tools/gen_golden_ngp.py gpurun_out/ngp_training.json'   then copy the file to tests/golden/
Regenerate (and say so in the commit) whenever a kernel of the step changes its arithmetic on purpose."""
import hashlib
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd"), os.path.join(root, "tools")]
import torch

from nerfslam.ngp import NgpConfig, NgpNerf
from ngp_scene import sphere_scene

STEPS = 48
SCENE = dict(n=4, H=60, W=80, f=75.0)


def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def state(dev, steps=STEPS):
    net = NgpNerf(NgpConfig(optimize_extrinsics=True), dev, seed=0)
    net.set_images(*sphere_scene(**SCENE))
    net.train_steps(steps, return_loss=False)
    torch.cuda.synchronize()
    return {"steps": steps, "scene": SCENE, "samples_of_last_step": int(net.last_samples), "loss": float(net.loss_tensor),
            "sha256": {"grid_half": sha(net.grid_half[:net.n_grid]), "mlp_master": sha(net.mlp_master), "c2w": sha(net.c2w),
                       "occupancy_bits": sha(net.bits)}}


if __name__ == "__main__":
    out = state(torch.device("cuda:0"))
    out["note"] = "regression pin of this repository's own arithmetic on gfx950; see tools/gen_golden_ngp.py"
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "tests", "golden", "ngp_training.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))
