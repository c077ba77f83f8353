#!/usr/bin/env python3
"""The dense BA of configs[4] alone, at its stated scale (160x90 grid, P = K' = 256, M = 3912: the graph of tests/synth.make_graph
-- radius-3 chain + random proximity pairs, 9..26 rows per depth map): `reduced_camera_matrix` + `solve_depth`, N iterations back
to back on device-resident inputs.  Prints ms per call by HIP events and checksums of H / v / E / disps (to compare kernel
variants bit for bit); run under `rocprofv3 --kernel-trace --stats` (or --pmc) for the per-kernel split -- what
profiles/r06_ba_c1280_* hold.  usage: ba_c1280_bench.py [iterations] [c640]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-slam_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

import synth
from nerfslam import ba_plan
from nerfslam._lib import check, lib, ptr, stream_ptr


def device_problem(dev, ht, wd, P, M, seed=61):
    """SURVEY 8(d) inputs, generated on the device (tests/synth.make_problem's numpy loop takes 15 s at this size)"""
    rng = np.random.default_rng(seed)
    ii, jj = synth.make_graph(P, M, rng)
    M = ii.shape[0]
    g = torch.Generator(device=dev).manual_seed(seed)
    n = P + 1
    poses = torch.zeros((n, 7), device=dev)
    poses[:, :3] = 0.05 * torch.randn((n, 3), device=dev, generator=g)
    w = 0.02 * torch.randn((n, 3), device=dev, generator=g)
    th = w.norm(dim=1, keepdim=True)
    poses[:, 3:6] = torch.sin(th / 2) * w / th
    poses[:, 6:] = torch.cos(th / 2)
    disps = 0.2 + 1.8 * torch.rand((n, ht, wd), device=dev, generator=g)
    sens = torch.zeros_like(disps)
    m = torch.rand(disps.shape, device=dev, generator=g) < 0.1
    sens[m] = disps[m] * 1.03
    W = wd * 8.0
    intr = torch.tensor([0.5 * W, 0.5 * W, (W - 1) / 2, (ht * 8.0 - 1) / 2], device=dev) / 8.0
    extr = torch.tensor([0, 0, 0, 0, 0, 0, 1.0], device=dev)
    tii, tjj = torch.from_numpy(ii).to(dev), torch.from_numpy(jj).to(dev)
    coords = torch.empty((M, ht, wd, 2), device=dev)
    valid = torch.empty((M, ht, wd), device=dev)
    check(lib().ns_reproject(ptr(poses), ptr(disps), ptr(intr), ptr(tii), ptr(tjj), ptr(coords), ptr(valid), M, ht, wd, stream_ptr()), "reproject")
    targets = (coords.permute(0, 3, 1, 2) + 0.5 * torch.randn((M, 2, ht, wd), device=dev, generator=g)).contiguous()
    del coords, valid
    weights = torch.rand((M, 2, ht, wd), device=dev, generator=g)
    eta = 0.2 * (1e-4 + (2e-2 - 1e-4) * torch.rand((P, ht * wd), device=dev, generator=g)) + 1e-7
    return dict(ii=ii, jj=jj, tii=tii, tjj=tjj, poses=poses, disps=disps, sens=sens, intr=intr, extr=extr, targets=targets,
                weights=weights, eta=eta, P=P, M=M, ht=ht, wd=wd)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
    small = "c640" in sys.argv
    dev = torch.device("cuda:0")
    ht, wd, P, M = (60, 80, 10, 96) if small else (90, 160, 256, 3912)
    pr = device_problem(dev, ht, wd, P, M)
    plan = ba_plan.BaPlan(pr["ii"], pr["jj"], 0, P, dev)
    dx = 1e-2 * torch.randn((P, 6), device=dev, generator=torch.Generator(device=dev).manual_seed(1))

    def rcm():
        return ba_plan.reduced_camera_matrix(plan, pr["poses"], pr["disps"], pr["intr"], pr["extr"], pr["sens"], pr["targets"],
                                             pr["weights"], pr["eta"], pr["tii"], pr["tjj"])

    H, v, Q, E, w = rcm()
    disps0 = pr["disps"].clone()
    d = pr["disps"].clone()
    ba_plan.solve_depth(plan, dx, d, Q, E, w, clamp_min=0.001)
    torch.cuda.synchronize()
    sums = {"H": float(H.double().sum()), "H_abs": float(H.double().abs().sum()), "v": float(v.double().sum()),
            "E_abs": float(E.double().abs().sum()), "Q": float(Q.double().sum()), "w": float(w.double().sum()),
            "disps_after": float(d.double().sum())}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_rcm = t_sd = 0.0
    for _ in range(n):
        ev[0].record()
        H, v, Q, E, w = rcm()
        ev[1].record()
        d.copy_(disps0)
        ba_plan.solve_depth(plan, dx, d, Q, E, w, clamp_min=0.001)
        ev[2].record()
        torch.cuda.synchronize()
        t_rcm += ev[0].elapsed_time(ev[1])
        t_sd += ev[1].elapsed_time(ev[2])
    HW = ht * wd
    Kp = plan.K
    alg = {"linearise_accumulate_contract_bytes": pr["M"] * HW * 20 + Kp * HW * 8 + (P + pr["M"]) * 6 * HW * 4 + 2 * Kp * HW * 4,
           "schur_min_bytes": (P + pr["M"]) * 6 * HW * 4 + 2 * Kp * HW * 4,
           "depth_update_bytes": (P + pr["M"]) * 6 * HW * 4 + 4 * Kp * HW * 4}
    print(json.dumps({"config": {"ht": ht, "wd": wd, "P": P, "M": pr["M"], "K": Kp, "n_pairs": plan.n_pairs, "iterations": n},
                      "reduced_camera_matrix_ms": t_rcm / n, "solve_depth_ms_incl_copy": t_sd / n, "checksums": sums,
                      "algorithmic_bytes": alg}))


if __name__ == "__main__":
    main()
