"""Both encoders at 640x480: the MFMA path (nerfslam/encoder_op.py, HIP-graph replay and eager) against the f16 torch / MIOpen
modules it replaces.  HIP events around back-to-back calls; prints ms per call."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "nerf-slam_amd"))
import torch
from nerfslam.droid_nets import DroidNetworks

dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
nets = DroidNetworks(dev, buffer=4)
img = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, device=dev)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


x = nets._normalize(img).half()
res = {}
res["torch_f16_fnet_ms"] = timed(lambda: nets.fnet_h(x))
res["torch_f16_cnet_ms"] = timed(lambda: nets.cnet_h(x))
i4 = img[None]
res["hip_graph_fnet_ms"] = timed(lambda: nets.fnet_hip(i4))
res["hip_graph_cnet_ms"] = timed(lambda: nets.cnet_hip(i4))
nets.fnet_hip.use_graph = nets.cnet_hip.use_graph = False
res["hip_eager_fnet_ms"] = timed(lambda: nets.fnet_hip(i4))
res["hip_eager_cnet_ms"] = timed(lambda: nets.cnet_hip(i4))
import json
print(json.dumps({k: round(v, 4) for k, v in res.items()}))
