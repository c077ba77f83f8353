#!/usr/bin/env python3
"""altcorr at config #5's grid (90x160, E = 48, smooth flow): f16 pyramid on the matrix cores vs the f32 tile kernel.
usage: python tools/altcorr_bench2.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-slam_amd")]
import torch
from nerfslam.corr import AltCorrBlock

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
NB, H, W, E = 64, 90, 160, 48
fm = (torch.randn((1, NB, 128, H, W), device=dev, generator=g) * 0.5).half()
ii = torch.arange(E, device=dev) % NB
jj = (ii + 1) % NB
gy, gx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
base = torch.stack([gx, gy], -1).float()
coords = (base[None] + torch.stack([3.0 * torch.sin(gy / 17.0) + 0.01 * gx, 2.0 * torch.cos(gx / 23.0)], -1)[None]).repeat(E, 1, 1, 1)[None].contiguous()
for name, feats in (("f16 pyramid, matrix cores", fm), ("f32 tile kernel", fm.float())):
    blk = AltCorrBlock(feats)
    out = blk(coords, ii, jj); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        blk(coords, ii, jj)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 5
    outb = E * 196 * H * W * 4
    print(f"{name:28s} {us:8.1f} us per call   output {outb / 1e6:.0f} MB -> {outb / us / 1e6:.2f} TB/s of writes")
