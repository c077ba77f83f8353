"""Distribution of the table gradient's inputs on the sphere scene (units of 2^-18 after the fixed-point scale): how many
samples carry gradient at all, how large the contributions are (what a narrower record format would have to hold).
usage: python tools/dfeat_hist.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd")]
import torch, importlib.util
from nerfslam.ngp import NgpConfig, NgpNerf
spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(root, "tools", "ngp_scene.py"))
sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
net = NgpNerf(NgpConfig(optimize_extrinsics=True), torch.device("cuda:0"), seed=0)
net.set_images(*sc.sphere_scene())
print("fixed scale", net.cfg.grad_fixed_scale, "loss scale", net.cfg.loss_scale)
for steps in (3, 30, 300):
    while net.step < steps:
        net.train_step(return_loss=False)
    torch.cuda.synchronize()
    d = net.s_dfeat.float().abs() * net.cfg.grad_fixed_scale
    n = int(net.last[2].item())
    d = d[:, :n]
    nz = d[d > 0]
    print("step", net.step, "samples", n, "nonzero frac %.3f" % (nz.numel() / d.numel()), "max %.0f" % float(d.max()),
          "quantiles 50/90/99/99.9: %s" % [float(torch.quantile(nz[:: max(1, nz.numel() // 1000000)], q)) for q in (0.5, 0.9, 0.99, 0.999)],
          "frac >= 2^16: %.5f  >= 2^15: %.5f  >= 2^14: %.5f" % tuple(float((d >= 2.0 ** k).float().mean()) for k in (16, 15, 14)))
