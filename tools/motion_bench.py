import sys, time, torch
sys.path.insert(0, "nerf-slam_amd")
from nerfslam.droid_nets import DroidNetworks
dev = torch.device("cuda"); ht, wd = 60, 80
n = DroidNetworks(dev, seed=1, hip_update=True)
img = torch.randint(0, 255, (3, 480, 640), dtype=torch.uint8)
n.features(img); n.begin_keyframe(0, img)
corr = torch.randn((1, 1, 196, ht, wd), device=dev).half()
z = torch.zeros((1, 4, ht, wd), device=dev)
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / it
print("motion (graph replay) %.3f ms" % t(lambda: n.motion(corr, 0)))
print("operator direct, E=1  %.3f ms" % t(lambda: n.update_op(n.ctx_cl[0][None], n.inp_cl[0][None], corr[0], z, [0])))
print("delta_only direct     %.3f ms" % t(lambda: n.update_op.delta_only(n.ctx_cl[0][None], n.inp_cl[0][None], corr[0])))
