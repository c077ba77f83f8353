"""NeRF trainer micro-benchmark: synthetic sphere scene, default NgpConfig; prints steps/s, samples/step.
usage: python tools/ngp_bench.py [steps] [warmup]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd")]
import numpy as np
import torch
from nerfslam.ngp import NgpConfig, NgpNerf

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
cfg = NgpConfig()
net = NgpNerf(cfg, dev, seed=0)
H, W, f = 120, 160, 150.0
imgs, deps, covs, poses = [], [], [], []
centre = np.array([0.5, 0.5, 0.5])
for k in range(8):
    a = 2 * np.pi * k / 8
    eye = centre + 1.2 * np.array([np.cos(a), 0.3, np.sin(a)])
    fwd = centre - eye; fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 1, 0]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = np.stack([right, -up, fwd, eye], 1)
    vv, uu = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d = np.stack([(uu + 0.5 - W / 2) / f, (vv + 0.5 - H / 2) / f, np.ones_like(uu, float)], -1) @ c2w[:, :3].T
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    oc = eye - centre
    b = (d * oc).sum(-1); disc = b * b - ((oc * oc).sum() - 0.25 ** 2)
    hit = disc > 0
    t = np.where(hit, -b - np.sqrt(np.maximum(disc, 0)), -1.0)
    pts = eye + t[..., None] * d
    col = np.where(hit[..., None], 0.5 + 0.5 * (pts - centre) / 0.25, 0.0)
    imgs.append(np.concatenate([col, hit[..., None].astype(float)], -1)); deps.append(t); covs.append(np.full((H, W), 0.05)); poses.append(c2w)
net.set_images(torch.tensor(np.array(imgs)), torch.tensor(np.array(deps)), torch.tensor(np.array(covs)), torch.tensor(np.array(poses)), (f, f, W / 2, H / 2))
for _ in range(warm):
    net.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter(); ns = 0; nr = 0
for _ in range(steps):
    net.train_step(); ns += net.last_samples; nr += net.counter[1].item()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"steps/s {steps / dt:.1f}  ms/step {1e3 * dt / steps:.3f}  samples/step {ns / steps:.0f}  rays-with-samples/step {nr / steps:.0f}  Msamples/s {ns / dt / 1e6:.1f}  loss {float(net.loss_tensor):.4f}")
