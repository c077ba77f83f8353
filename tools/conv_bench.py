"""ns_conv_nhwc_f16 at the ConvGRU's shapes (E=48 edges, 80x60, [h,inp,corr,flow] = 448 channels) against torch/MIOpen.
usage: python tools/conv_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-slam_amd"))
import torch
import torch.nn.functional as F
from nerfslam.conv import PackedConv, conv_nhwc

dev = torch.device("cuda")
torch.backends.cudnn.benchmark = True
N, H, W = 48, 60, 80


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


CASES = (((128, 128, 128, 64), 256, 3), ((128, 128, 128, 64), 128, 3), ((128,), 128, 3), ((128,), 384, 3),
         ((128,), 64, 3), ((208,), 128, 1), ((128,), 576, 1))
WITH_TORCH = os.environ.get("NS_CONV_BENCH_TORCH", "1") == "1"
for chans, cout, k in CASES:
    cin = sum(chans)
    srcs = [torch.randn((N, H, W, c), device=dev).half() for c in chans]
    w = (torch.randn((cout, cin, k, k), device=dev) / (cin * k * k) ** 0.5)
    b = torch.randn((cout,), device=dev)
    layer = PackedConv(w, b)
    out = torch.empty((N, H, W, cout), dtype=torch.float16, device=dev)
    flops = 2.0 * N * H * W * cin * cout * k * k
    line = f"{k}x{k} {cin:4d}->{cout:4d}: "
    ms = timeit(lambda: conv_nhwc(srcs, layer, act="relu", out=out))
    line += f"hip {ms*1e3:7.1f} us {flops/ms/1e9:6.0f} TF/s | "
    if not WITH_TORCH:
        print(line, flush=True)
        continue
    x = torch.cat(srcs, -1).permute(0, 3, 1, 2)             # channels-last strides
    xc = x.contiguous()                                      # NCHW
    wh, bh = w.half(), b.half()
    wcl = wh.contiguous(memory_format=torch.channels_last)
    ms1 = timeit(lambda: torch.relu(F.conv2d(xc, wh, bh, padding=k // 2)))
    ms2 = timeit(lambda: torch.relu(F.conv2d(x, wcl, bh, padding=k // 2)))
    line += f"torch nchw {ms1*1e3:7.1f} us {flops/ms1/1e9:6.0f} TF/s | torch nhwc {ms2*1e3:7.1f} us {flops/ms2/1e9:6.0f} TF/s"
    print(line, flush=True)
