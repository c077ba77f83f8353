"""ns_conv_nhwc_f16 at the shapes the tracker runs with ONE image / ONE edge (encoders at 640x480, the motion filter's update):
few workgroups, each walking its K loop alone.  HIP events around replays of a graph of 40 launches; prints us per launch and the launch's
share of the MFMA peak.  usage: python tools/small_conv_bench.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-slam_amd"))
import torch
from nerfslam.conv import PackedConv, conv_nhwc

dev = torch.device("cuda")
# (name, H, W, source channels, cout, ksize)
CASES = (("stem 7x7 as 1x1 160->32 @240x320", 240, 320, (160,), 32, 1),
         ("layer1 3x3 32->32 @240x320", 240, 320, (32,), 32, 3),
         ("layer2 s2 as 1x1 288->64 @120x160", 120, 160, (288,), 64, 1),
         ("layer2 3x3 64->64 @120x160", 120, 160, (64,), 64, 3),
         ("layer2 shortcut 1x1 32->64 @120x160", 120, 160, (32,), 64, 1),
         ("layer3 s2 as 1x1 576->128 @60x80", 60, 80, (576,), 128, 1),
         ("layer3 3x3 128->128 @60x80", 60, 80, (128,), 128, 3),
         ("head 1x1 128->128 @60x80", 60, 80, (128,), 128, 1),
         ("head 1x1 128->256 @60x80", 60, 80, (128,), 256, 1),
         ("gate 3x3 448->256 @60x80", 60, 80, (128, 128, 128, 64), 256, 3),
         ("gate 3x3 448->128 @60x80", 60, 80, (128, 128, 128, 64), 128, 3),
         ("corr enc 3x3 128->128 @60x80 (E=1)", 60, 80, (128,), 128, 3),
         ("flow enc 1x1 208->128 @60x80", 60, 80, (208,), 128, 1),
         ("heads 3x3 128->384 @60x80", 60, 80, (128,), 384, 3))


def timeit(fn, iters=40):
    """GPU time per launch: `iters` launches captured in ONE HIP graph (no host launch path between them), replayed 5 times"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


res = {}
for name, H, W, chans, cout, k in CASES:
    cin = sum(chans)
    srcs = [torch.randn((1, H, W, c), device=dev).half() for c in chans]
    layer = PackedConv(torch.randn((cout, cin, k, k), device=dev) / (cin * k * k) ** 0.5, torch.randn((cout,), device=dev))
    out = torch.empty((1, H, W, cout), dtype=torch.float16, device=dev)
    us = timeit(lambda: conv_nhwc(srcs, layer, act="relu", out=out))
    gflop = 2.0 * H * W * cin * cout * k * k / 1e9
    res[name] = {"us": round(us, 2), "gflop": round(gflop, 3), "frac_of_2500_tflops": round(gflop / us * 1e-3 / 2500.0 * 1e3, 4)}
    print(f"{name:42s} {us:7.2f} us  {gflop:6.3f} GFLOP  {gflop / us * 1e3 / 1e3:7.1f} TFLOP/s")
print(json.dumps(res))
