#!/usr/bin/env python3
"""MLP weight-gradient kernel alone at 2^18 samples: time and agreement with the staged (round-3) kernel.
usage: NS_NGP_WGRAD=staged|tr python tools/r04_wgrad_bench.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-slam_amd")]
from nerfslam._lib import check, lib, ptr, stream_ptr  # noqa: E402


def us(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


dev = torch.device("cuda:0")
N = 1 << 18
g = torch.Generator(device=dev).manual_seed(3)
W = (torch.randn(10240, device=dev, generator=g) * 0.2).half()
featT = (torch.randn((32, N), device=dev, generator=g) * 0.5).half().contiguous()
dirs = torch.nn.functional.normalize(torch.randn((N, 3), device=dev, generator=g), dim=-1).contiguous()
dout = (torch.randn((N, 4), device=dev, generator=g) * 1e-2).half().contiguous()
frags = torch.zeros(int(lib().ns_ngp_mlp_fragment_table_bytes()) // 2, dtype=torch.float16, device=dev)
check(lib().ns_ngp_mlp_pack_fragments(ptr(W), ptr(frags), stream_ptr()), "pack")
wgs = 512
part = torch.zeros((wgs, 10240), device=dev)
n_dev = torch.tensor([218000], dtype=torch.int32, device=dev)
gw = torch.zeros(10240, device=dev)


def run():
    check(lib().ns_ngp_mlp_wgrad_recompute_n(ptr(frags), ptr(featT), ptr(dirs), ptr(dout), ptr(part), wgs, ptr(gw), C.c_long(N), ptr(n_dev),
                                             stream_ptr()), "wgrad")


gw.zero_(); run(); torch.cuda.synchronize()
ref = gw.clone()
print(f"NS_NGP_WGRAD={os.environ.get('NS_NGP_WGRAD')}: wgrad + reduce {us(run):7.1f} us   |sum| {float(ref.abs().sum()):.6e}  max {float(ref.abs().max()):.4e}")
torch.save(ref.cpu(), f"/tmp/wgrad_{os.environ.get('NS_NGP_WGRAD', 'tr')}.pt")
other = "/tmp/wgrad_staged.pt" if os.environ.get("NS_NGP_WGRAD", "tr") != "staged" else None
if other and os.path.exists(other):
    o = torch.load(other)
    print(f"   vs staged: max |diff| {float((ref.cpu() - o).abs().max()):.3e} of max {float(o.abs().max()):.3e}")
