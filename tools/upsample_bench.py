
import os, sys, ctypes as C, torch
sys.path.insert(0, "nerf-slam_amd")
from nerfslam._lib import check, lib, ptr, stream_ptr
dev = torch.device("cuda")
n, ht, wd, nb = 10, 60, 80, 16
a = torch.rand((nb, ht, wd), device=dev); b = torch.rand((nb, ht, wd), device=dev)
kx = torch.arange(3, 13, device=dev)
oa = torch.zeros((nb, 8*ht, 8*wd), device=dev); ob = torch.zeros_like(oa)
big = torch.zeros((512 << 20) // 4, device=dev)
for rep in range(3):
  for dt, code in ((torch.float16, 1), (torch.float32, 2)):
    m = torch.randn((n, 576, ht, wd), device=dev).to(dt)
    for one in (False, True):
        os.environ["NS_VARIANTS"] = "1"
        if one: os.environ["NS_CVX_ONE_PIXEL"] = "1"
        else: os.environ.pop("NS_CVX_ONE_PIXEL", None)
        def run():
            check(lib().ns_cvx_upsample_keyframes(ptr(a), ptr(b), ptr(kx), ptr(m), code, ptr(oa), ptr(ob), n, ht, wd, C.c_float(1.0), stream_ptr()), "x")
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): run()
        e1.record(); torch.cuda.synchronize()
        hot = e0.elapsed_time(e1) * 5
        # cold: flush caches with a 512 MB fill between launches
        tot = 0.0
        for _ in range(10):
            big.add_(1.0); 
            e0.record(); run(); e1.record(); torch.cuda.synchronize(); tot += e0.elapsed_time(e1) * 1e3
        print(dt, "one-pixel" if one else "pair-px", "hot %.1f us  cold %.1f us" % (hot, tot / 10))
