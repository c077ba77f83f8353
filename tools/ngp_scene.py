"""synthetic NeRF training scene shared by tools/ngp_bench.py and the tests: a coloured sphere (radius 0.25, centre
(0.5,0.5,0.5)) seen from a ring of cameras; exact depth, constant depth covariance."""
import numpy as np
import torch


def sphere_scene(n=8, H=120, W=160, f=150.0, radius=1.2):
    imgs, deps, covs, poses = [], [], [], []
    centre = np.array([0.5, 0.5, 0.5])
    for k in range(n):
        a = 2 * np.pi * k / n
        eye = centre + radius * np.array([np.cos(a), 0.3, np.sin(a)])
        fwd = centre - eye; fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, [0, 1, 0]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        c2w = np.stack([right, -up, fwd, eye], 1)  # camera looks along +z, y down
        vv, uu = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        d = np.stack([(uu + 0.5 - W / 2) / f, (vv + 0.5 - H / 2) / f, np.ones_like(uu, float)], -1) @ c2w[:, :3].T
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        oc = eye - centre
        b = (d * oc).sum(-1); disc = b * b - ((oc * oc).sum() - 0.25 ** 2)
        hit = disc > 0
        t = np.where(hit, -b - np.sqrt(np.maximum(disc, 0)), -1.0)
        pts = eye + t[..., None] * d
        col = np.where(hit[..., None], 0.5 + 0.5 * (pts - centre) / 0.25, 0.0)
        imgs.append(np.concatenate([col, hit[..., None].astype(float)], -1)); deps.append(t); covs.append(np.full((H, W), 0.05))
        poses.append(c2w)
    t32 = lambda x: torch.tensor(np.array(x), dtype=torch.float32)
    return t32(imgs), t32(deps), t32(covs), t32(poses), (f, f, W / 2, H / 2)
