#!/usr/bin/env python3
"""Table gradient A/B under the current environment (NS_ENC_DENSE_BINNED, NS_FB_MERGE_RES, ...): bit-equality of the packed sums
with the round-2 path, and the time of each part of ns_ngp_encode_backward_fused_n (1 scatter, 2 accumulate + Adam, 4|8 dense
owner-computes + reduce) on 2^18 samples along rays.  usage: python tools/r04_bwd_ab.py [live_fraction] [uniform]"""
import ctypes as C
import os
import sys

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
    uniform = len(sys.argv) > 2 and sys.argv[2] == "uniform"
    gscale = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3      # sigma of the upstream gradient (1e-3: dense records; 2e-6: a converged scene)
    dev = torch.device("cuda:0")
    c = NgpConfig(aabb_scale=int(os.environ.get("NS_AABB", "4")))
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
    if uniform:
        pos = torch.rand((N, 3), device=dev, generator=g).contiguous()
    dfeat = (torch.randn((32, N), device=dev, generator=g) * gscale).half()
    if live < 1.0:      # the tail of every ray carries no gradient (transmittance gone), as in training
        steps = N // R
        dead = (torch.arange(steps, device=dev)[None, :] >= (steps * (2 * live * torch.rand((R, 1), device=dev, generator=g))).clamp(max=steps)).reshape(N)
        dfeat[:, dead] = 0
    dfeat = dfeat.contiguous()
    S = c.grad_fixed_scale
    st = {k: torch.zeros(n_par, device=dev) for k in ("master", "m1", "m2")}
    hp = torch.zeros(n_par, dtype=torch.float16, device=dev)
    grad, ref = torch.zeros(n_par, device=dev), torch.zeros(n_par, device=dev)
    ws_old = torch.zeros(lib().ns_ngp_encode_backward_workspace_bytes(*args, C.c_long(N)) // 4 + 1, device=dev)
    wsb = int(lib().ns_ngp_encode_backward_fused_workspace_bytes(*args, C.c_long(N)))
    ws_new = torch.zeros(wsb // 8 + 1, dtype=torch.int64, device=dev)
    check(lib().ns_ngp_encode_backward(*args, ptr(pos), ptr(dfeat), 1, ptr(ref), ptr(ws_old), C.c_size_t(ws_old.numel() * 4), C.c_float(S),
                                       C.c_long(N), stream_ptr()), "old")

    def grad_only(parts):
        check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(pos), ptr(dfeat), ptr(grad), ptr(ws_new), C.c_size_t(wsb), C.c_float(S),
                                                   C.c_long(N), None, None, None, None, None, 7, C.c_float(0), C.c_float(0), C.c_float(0),
                                                   C.c_float(0), C.c_float(1), None, parts, stream_ptr()), "new")

    def fused(parts):
        check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(pos), ptr(dfeat), None, ptr(ws_new), C.c_size_t(wsb), C.c_float(S), C.c_long(N),
                                                   None, ptr(st["master"]), ptr(hp), ptr(st["m1"]), ptr(st["m2"]), 7, C.c_float(c.lr),
                                                   C.c_float(c.beta1), C.c_float(c.beta2), C.c_float(c.eps), C.c_float(c.loss_scale), None,
                                                   parts, stream_ptr()), "new")
    grad_only(15); torch.cuda.synchronize()
    same = bool(torch.equal(grad.view(torch.int64), ref.view(torch.int64)))
    touched = int((ref.view(torch.int64) != 0).sum())
    nd = int(lib().ns_ngp_encode_backward_fused_dense_levels(*args))
    ctr = ws_new[:1].view(torch.int32)[:3].tolist()
    print(f"env DENSE_BINNED={os.environ.get('NS_ENC_DENSE_BINNED')} MERGE_RES={os.environ.get('NS_FB_MERGE_RES')} aabb={c.aabb_scale} live={live} "
          f"{'uniform' if uniform else 'rays'}: owner-computes levels {nd}, touched {touched}, bit-identical to the round-2 path: {same}, ctr {ctr}")
    fused(1); torch.cuda.synchronize()
    # records per (level, bin) of the scatter just run: workspace = [256 B counters][cnt: nh x 64 x ntiles run lengths] ...
    nh, ntiles = c.n_levels - nd if int(os.environ.get("NS_ENC_DENSE_BINNED", "2")) else 16 - nd - 0, (N + 1023) // 1024
    try:
        import numpy as np
        nlev_hashed = sum(1 for l in range(c.n_levels) if (off[l + 1] - off[l]) == (1 << c.log2_hashmap))
        mode = int(os.environ.get("NS_ENC_DENSE_BINNED", "2"))
        nh = nlev_hashed + (c.n_levels - nlev_hashed - nd if mode else 0)
        cnt = ws_new.view(torch.int32)[64:64 + nh * 64 * ntiles].reshape(nh, 64, ntiles).sum(dim=2).cpu().numpy()
        print("   records per binned level (total | heaviest bin | bins):", " ".join(f"{int(r.sum())}|{int(r.max())}|{int((r > 0).sum())}" for r in cnt))
    except Exception as e:      # (diagnostic only)
        print("   (record statistics unavailable:", e, ")")
    t1 = us(lambda: fused(1))
    t2 = us(lambda: fused(2))
    t48 = us(lambda: fused(12)) if nd else 0.0
    tall = us(lambda: fused(15))
    print(f"   gradient sigma {gscale:g}: scatter {t1:7.1f}  accumulate+adam {t2:7.1f}  dense {t48:7.1f}  all four {tall:7.1f} us")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
