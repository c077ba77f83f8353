"""one ConvGRU-gate convolution (E=48, 80x60, 448 -> 256, 3x3) a few times: target of rocprofv3 --pmc passes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf-slam_amd"))
import torch
from nerfslam.conv import PackedConv, conv_nhwc
dev = torch.device("cuda")
N, H, W = 48, 60, 80
chans, cout = (128, 128, 128, 64), int(sys.argv[1]) if len(sys.argv) > 1 else 256
srcs = [torch.randn((N, H, W, c), device=dev).half() for c in chans]
layer = PackedConv(torch.randn((cout, sum(chans), 3, 3), device=dev) / 63.0, torch.randn((cout,), device=dev))
out = torch.empty((N, H, W, cout), dtype=torch.float16, device=dev)
for _ in range(5):
    conv_nhwc(srcs, layer, act="sigmoid", out=out)
torch.cuda.synchronize()
