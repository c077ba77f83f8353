"""run only the HIP update operator (E=48, 80x60) a few times: target of rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf-slam_amd"))
import torch
from nerfslam.droid_nets import UpdateModule
from nerfslam.update_op import HipUpdateOperator
dev = torch.device("cuda")
torch.manual_seed(0)
E, ht, wd = 48, 60, 80
op = HipUpdateOperator(UpdateModule().to(dev).eval())
net = torch.randn((E, ht, wd, 128), device=dev).half(); inp = torch.randn((E, ht, wd, 128), device=dev).half()
corr = torch.randn((E, 196, ht, wd), device=dev).half(); flow = torch.randn((E, 4, ht, wd), device=dev)
ii = [k % 10 for k in range(E)]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    op(net, inp, corr, flow, ii)
torch.cuda.synchronize()
