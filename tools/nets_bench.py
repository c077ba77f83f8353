"""Time the conv nets of the tracker (encoders + UpdateModule/ConvGRU, torch + MIOpen/hipBLASLt, f16
autocast, random-init weights) at the bench's shapes: one 640x480 frame through feature_net / context_net, and one
update_net call over E=48 edges at 80x60.  They are NOT part of bench.py's step; this script says what they would add.

usage: python tools/nets_bench.py [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-slam_amd"))
import torch
from nerfslam.droid_nets import DroidNet

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.backends.cudnn.benchmark = os.environ.get("NS_NETS_FIND", "1") == "1"   # MIOpen Find (with workspace) instead of the immediate-mode fallback
dev = torch.device("cuda")
torch.manual_seed(0)
net = DroidNet().to(dev).eval()
if os.environ.get("NS_NETS_CL"):
    net = net.to(memory_format=torch.channels_last)
H, W, E = 480, 640, 48
ht, wd = H // 8, W // 8
img = torch.randn((1, 1, 3, H, W), device=dev)
hid = torch.randn((1, E, 128, ht, wd), device=dev).half()
inp = torch.randn((1, E, 128, ht, wd), device=dev).half()
corr = torch.randn((1, E, 196, ht, wd), device=dev).half()
flow = torch.randn((1, E, 4, ht, wd), device=dev)
ii = torch.arange(E, device=dev) % 10
jj = (ii + 1) % 10


def timed(name, fn):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    print(f"{name:28s} {ms:8.3f} ms")
    return ms


f = timed("feature_net (1 frame)", lambda: net.feature_net(img))
c = timed("context_net (1 frame)", lambda: net.context_net(img))
u = timed("update_net (E=48)", lambda: net.update_net(hid, inp, corr, flow, ii, jj))
from nerfslam.update_op import HipUpdateOperator
op = HipUpdateOperator(net.update_net)
cl = lambda t: t[0].permute(0, 2, 3, 1).contiguous().half()
hid_cl, inp_cl, ih = cl(hid), cl(inp), ii.tolist()
uh = timed("update operator, HIP (E=48)", lambda: op(hid_cl, inp_cl, corr[0], flow[0], ih))
u1 = timed("update_net (E=1, motion)", lambda: net.update_net(hid[:, :1], inp[:, :1], corr[:, :1], flow[:, :1], ii[:1], jj[:1]))
step = f + c + u1 + 6 * u
print(f"conv nets per keyframe step: {step:.2f} ms  (feature + context + motion filter + 6 updates), "
      f"{f + c + u1 + 6 * uh:.2f} ms with the HIP update operator")
