#!/usr/bin/env python3
"""The reduced camera system of config #5's global BA (256 poses, 6P = 1536, f64): csrc/ba_solve_large.hip (blocked Cholesky through
HBM, bordered right-hand side, back substitution, retraction) against torch.linalg (rocSOLVER potrf + potrs), what round 2 used."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-slam_amd")]
import torch
from nerfslam import ba_plan

dev = torch.device("cuda:0")
torch.manual_seed(0)
for P in (40, 256):
    n = 6 * P
    A = torch.randn((n, n), device=dev, dtype=torch.float64)
    H = (A @ A.t() / n + torch.eye(n, device=dev, dtype=torch.float64)).float().contiguous()
    v = torch.randn((n, 1), device=dev)
    wTb = torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]], device=dev).repeat(P, 1)
    cTw, cTb = wTb.clone(), wTb[0].clone()

    def hip():
        return ba_plan.ba_solve(H, v, 0, P, wTb, cTw, cTb)

    def rocsolver():
        Hd = torch.triu(H.double()); Hd = Hd + torch.triu(Hd, 1).t()
        L, info = torch.linalg.cholesky_ex(Hd)
        return torch.cholesky_solve(v.double(), L)

    for name, fn in (("csrc/ba_solve_large.hip", hip), ("torch.linalg (rocSOLVER)", rocsolver)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"6P = {n:5d}  {name:28s} {e0.elapsed_time(e1) / 10:8.3f} ms per solve")
    x1, x2 = hip()["dx"].reshape(-1).double(), rocsolver().reshape(-1)
    print(f"            max |dx_hip - dx_rocsolver| / max |dx| = {float((x1 - x2).abs().max() / x2.abs().max()):.2e}")
