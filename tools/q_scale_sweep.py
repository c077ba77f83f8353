#!/usr/bin/env python3
"""Fixed-point scale of the table gradient (NgpConfig.grad_fixed_scale = 2^q) against rendered quality and step time: the sphere
scene of tools/ngp_scene.py, three ray seeds per arm, PSNR of the 8 training views + depth L1 after 600 and 3000 steps, the step
time and the records / touched entries of the last step.  q = 0 is the f32-atomic gradient (the published algorithm's form)."""
import importlib.util, json, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd")]
import numpy as np
import torch
from nerfslam import eval as ev
from nerfslam.ngp import NgpConfig, NgpNerf
spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(root, "tools", "ngp_scene.py"))
sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
dev = torch.device("cuda:0")
imgs, deps, covs, poses, intr = sc.sphere_scene(n=8, H=60, W=80, f=75.0)
out = {}
for q in [int(a) for a in sys.argv[1:]] or [18, 20, 22, 0]:
    res = {}
    for steps in (600, 3000):
        ps, de, ms = [], [], []
        for seed in (0, 1, 2):
            net = NgpNerf(NgpConfig(grad_fixed_scale=float(2 ** q) if q else 0.0), dev, seed=seed)
            net.set_images(imgs, deps, covs, poses, intr)
            net.train_steps(steps - 64, return_loss=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            net.train_steps(64, return_loss=False)
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) / 64 * 1e3)
            for k in range(8):
                rgb, dep = net.render(poses[k], 60, 80)
                ps.append(ev.psnr(rgb.cpu(), imgs[k, ..., :3]))
                m = deps[k] > 0
                de.append(float((dep.cpu()[m] - deps[k][m]).abs().mean()))
            del net
        res[str(steps)] = {"psnr_db_mean": float(np.mean(ps)), "depth_l1_mm_mean": 1e3 * float(np.mean(de)), "ms_per_step": float(np.mean(ms)),
                           "psnr_db_by_seed": [float(np.mean(ps[8 * i:8 * i + 8])) for i in range(3)]}
    out["q%d" % q] = res
    print("q", q, json.dumps(res))
print("Q_SCALE_SWEEP " + json.dumps(out))
