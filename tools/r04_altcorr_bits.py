"""sha256 of the on-the-fly correlation's outputs on fixed inputs (run once per kernel variant: NS_ALTCORR_DIRECT=1 selects
round 3's kernels with fragments straight from global memory; the default is round 4's LDS-staged ones)."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-slam_amd"))
from nerfslam.corr import AltCorrBlock
from nerfslam.update_op import CorrEncoderWeights

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
NB, ht, wd, E = 8, 90, 160, 12
fm = torch.randn((1, NB, 128, ht, wd), device=dev, generator=g).half()
alt = AltCorrBlock(fm)
ii = torch.arange(0, E, device=dev) % NB
jj = (ii + 1) % NB
gy, gx = torch.meshgrid(torch.arange(ht, device=dev), torch.arange(wd, device=dev), indexing="ij")
grid = torch.stack([gx, gy], -1).float()[None]
coords = grid + 6.0 * torch.randn((E, 1, 1, 2), device=dev, generator=g) + 2.0 * torch.randn((E, ht, wd, 2), device=dev, generator=g)
coords[3] += 40.0 * torch.randn((ht, wd, 2), device=dev, generator=g)          # wild: wave-per-pixel fallback
coords[5] += torch.tensor([wd * 0.8, -ht * 0.6], device=dev)                   # leaves the image
coords = coords[None].contiguous()
out = alt(coords, ii, jj)
enc = CorrEncoderWeights(torch.randn((128, 196, 1, 1), device=dev, generator=g) / 14.0, 0.1 * torch.randn(128, device=dev, generator=g))
fused = alt.encoded(coords, ii, jj, enc).c1
torch.cuda.synchronize()
h = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
print("variant", "direct" if os.environ.get("NS_ALTCORR_DIRECT") else "lds", "plain", h(out), "fused", h(fused), "finite", bool(torch.isfinite(out).all()))
