import os, sys, faulthandler
faulthandler.enable()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd")]
import torch, importlib.util
from nerfslam.ngp import NgpConfig, NgpNerf
spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(root, "tools", "ngp_scene.py"))
sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
dev = torch.device("cuda:0")
imgs, deps, covs, poses, intr = sc.sphere_scene(n=8, H=60, W=80, f=75.0)
for use_graph in (False, True):
    net = NgpNerf(NgpConfig(use_graph=use_graph), dev, seed=0)
    net.set_images(imgs, deps, covs, poses.clone(), intr)
    for _ in range(100):
        net.train_step(return_loss=False)
    net.c2w[0, :, 3] += torch.tensor([0.03, -0.02, 0.01], device=dev)
    net.cfg.lr, net.cfg.optimize_extrinsics, net.cfg.extrinsic_lr_pos, net.cfg.extrinsic_lr_rot = 0.0, True, 3e-4, 0.0
    for k in range(4):
        net.train_step(return_loss=False)
        torch.cuda.synchronize()
        print("graph" if use_graph else "eager", k, "last", net.last.tolist(), "ctl", net.ctl.tolist()[:4], "dfeat", float(net.s_dfeat.float().abs().sum()),
              "dpos", float(net.dpos.abs().sum()), "ray_g", float(net.ray_g.abs().sum()), "cam_grad", float(net.cam_grad.abs().sum()),
              "cam_m1", float(net.cam_m1.abs().sum()), "c2w0", net.c2w[0, :, 3].tolist(), flush=True)
