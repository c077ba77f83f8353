#!/usr/bin/env python3
"""A/B of the hash-grid table gradient + optimiser step at a full sample budget (2^18 samples along rays, every sample live):
   round 2: zero + count + scatter + accumulate + dense (gradient buffer)  THEN  the streaming Adam pass over the table
   round 3: scatter + accumulate-with-Adam + dense-with-Adam (no gradient buffer)
usage: python tools/ngp_bwd_bench.py [live_fraction]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-slam_amd")]
from nerfslam._lib import check, lib, ptr, stream_ptr  # noqa: E402
from nerfslam.ngp import NgpConfig  # noqa: E402


def us(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    live = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    dev = torch.device("cuda:0")
    c = NgpConfig()
    args = (c.n_levels, 2, c.log2_hashmap, c.base_res, C.c_float(c.per_level_scale))
    off = (C.c_uint32 * (c.n_levels + 1))()
    check(lib().ns_ngp_grid_layout(*args, None, None, off), "layout")
    n_par = int(off[c.n_levels]) * 2
    N, R = c.max_samples, 2048
    g = torch.Generator(device=dev).manual_seed(1)
    o = torch.rand((R, 1, 3), device=dev, generator=g) * 0.4 + 0.3
    d = torch.nn.functional.normalize(torch.randn((R, 1, 3), device=dev, generator=g), dim=-1)
    t = (0.02 + 0.0017 * torch.arange(N // R, device=dev))[None, :, None]
    pos = (o + t * d).clamp(0.0, 1.0).reshape(N, 3).contiguous()
    dfeat = (torch.randn((32, N), device=dev, generator=g) * 1e-3).half()
    if live < 1.0:
        dfeat[:, torch.rand(N, device=dev, generator=g) > live] = 0
    dfeat = dfeat.contiguous()
    S = c.grad_fixed_scale
    st = {k: torch.zeros(n_par, device=dev) for k in ("master", "m1", "m2")}
    hp = torch.zeros(n_par, dtype=torch.float16, device=dev)
    grad = torch.zeros(n_par, device=dev)
    ws_old = torch.zeros(lib().ns_ngp_encode_backward_workspace_bytes(*args, C.c_long(N)) // 4 + 1, device=dev)
    wsb = int(lib().ns_ngp_encode_backward_fused_workspace_bytes(*args, C.c_long(N)))
    ws_new = torch.zeros(wsb // 8 + 1, dtype=torch.int64, device=dev)

    def old_grad():
        check(lib().ns_ngp_encode_backward(*args, ptr(pos), ptr(dfeat), 1, ptr(grad), ptr(ws_old), C.c_size_t(ws_old.numel() * ws_old.element_size()), C.c_float(S), C.c_long(N), stream_ptr()), "old")

    def adam():
        check(lib().ns_ngp_adam(ptr(st["master"]), ptr(hp), ptr(grad), ptr(st["m1"]), ptr(st["m2"]), C.c_long(n_par), 7, C.c_float(c.lr),
                                C.c_float(c.beta1), C.c_float(c.beta2), C.c_float(c.eps), C.c_float(0.0), C.c_float(c.loss_scale),
                                C.c_float(S), stream_ptr()), "adam")

    def old_both():
        old_grad(); adam()

    def new_grad_only():
        check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(pos), ptr(dfeat), ptr(grad), ptr(ws_new), C.c_size_t(wsb), C.c_float(S),
                                                   C.c_long(N), None, None, None, None, None, 7, C.c_float(0), C.c_float(0), C.c_float(0),
                                                   C.c_float(0), C.c_float(1), None, 15, stream_ptr()), "new")

    def new_fused():
        check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(pos), ptr(dfeat), None, ptr(ws_new), C.c_size_t(wsb), C.c_float(S), C.c_long(N),
                                                   None, ptr(st["master"]), ptr(hp), ptr(st["m1"]), ptr(st["m2"]), 7, C.c_float(c.lr),
                                                   C.c_float(c.beta1), C.c_float(c.beta2), C.c_float(c.eps), C.c_float(c.loss_scale), None,
                                                   15, stream_ptr()), "new")
    old_grad(); torch.cuda.synchronize()
    touched = int((grad.view(torch.int64) != 0).sum())
    grad.zero_()
    print(f"N = {N}, live fraction {live}: {touched} of {n_par // 2} table entries touched ({touched / (n_par / 2):.2%})")
    for name, fn in (("r02 table gradient (zero+count+scatter+accum+dense)", old_grad), ("r02 adam (whole table)", adam),
                     ("r02 gradient + adam", old_both), ("r03 table gradient only (scatter+accum+dense)", new_grad_only),
                     ("r03 gradient with fused adam", new_fused)):
        grad.zero_()
        print(f"  {name:58s} {us(fn):8.1f} us")


if __name__ == "__main__":
    main()
