"""NeRF trainer micro-benchmark: synthetic sphere scene, default NgpConfig; prints steps/s, samples/step.
usage: python tools/ngp_bench.py [steps] [warmup]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd")]
import numpy as np
import torch
from nerfslam.ngp import NgpConfig, NgpNerf

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
cfg = NgpConfig(optimize_extrinsics=bool(os.environ.get("NS_NGP_EXTRINSICS")),      # NerfFusion switches this on
                use_graph=not os.environ.get("NS_NGP_NO_GRAPH"))
for kv in filter(None, os.environ.get("NS_NGP_CFG", "").split(",")):      # A/B hook: "field=value,..." (ints / floats / 0-1 booleans)
    k_, v_ = kv.split("=")
    cur = getattr(cfg, k_)
    setattr(cfg, k_, bool(int(v_)) if isinstance(cur, bool) else type(cur)(v_))
net = NgpNerf(cfg, dev, seed=0)
import importlib.util
spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(root, "tools", "ngp_scene.py"))
sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
net.set_images(*sc.sphere_scene())
# (a stream of its own, like the mapper thread of the pipeline: NS_NGP_NULL_STREAM=1 for the legacy default stream)
work = torch.cuda.current_stream() if os.environ.get("NS_NGP_NULL_STREAM") else torch.cuda.Stream(device=dev)
with torch.cuda.stream(work):
    for _ in range(warm // 16):                 # 16 steps per call, as pyngp's frame() asks for them
        net.train_steps(16, return_loss=False)
    torch.cuda.synchronize()
    steps = steps // 16 * 16
    t0 = time.perf_counter(); ns = 0; nr = 0
    for _ in range(steps // 16):
        net.train_steps(16, return_loss=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
ns, nr = steps * net.last_samples, steps * int(net.last[1].item())   # (read once, after the timed loop: no per-step sync)
print(f"steps/s {steps / dt:.1f}  ms/step {1e3 * dt / steps:.3f}  samples/step {ns / steps:.0f}  rays-with-samples/step {nr / steps:.0f}  Msamples/s {ns / dt / 1e6:.1f}  loss {float(net.loss_tensor):.4f}")

# quality: render the training views at their poses and compare with the ground truth (linear rgb), PSNR in dB
from nerfslam import eval as ev
imgs, deps, covs, poses, intr = sc.sphere_scene()
ps, l1 = [], []
for k in (0, 3):
    rgb, dep = net.render(poses[k], imgs.shape[1], imgs.shape[2])
    ps.append(ev.psnr(rgb.cpu(), imgs[k, ..., :3]))
    m = deps[k] > 0
    l1.append(float((dep.cpu()[m] - deps[k][m]).abs().mean()))
print(f"after {net.step} steps: PSNR {sum(ps) / len(ps):.1f} dB over 2 training views, mean |depth error| on the object {100 * sum(l1) / len(l1):.2f} cm")
