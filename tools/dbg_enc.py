import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-slam_amd"))
import torch, oracle
from nerfslam._lib import check, lib, ptr, stream_ptr
cfg = oracle.ngp_cfg(n_levels=2, log2_hashmap=14, base_res=4, per_level_scale=2.0)
_, res, off = oracle.ngp_grid_layout(cfg)
n_par = int(off[-1]) * 2
pos = np.array([[0.3, 0.6, 0.2]], np.float32)
dL = np.array([[1.0, 2.0, 3.0, 4.0]], np.float16)
args = (2, 2, 14, 4, C.c_float(2.0))
grad = torch.zeros(n_par, dtype=torch.float32, device="cuda")
check(lib().ns_ngp_encode_backward(*args, ptr(torch.from_numpy(pos).cuda()), ptr(torch.from_numpy(dL).cuda()), ptr(grad), None, C.c_long(1), stream_ptr()), "bwd")
g = grad.cpu().numpy(); r = oracle.ngp_encode_bwd(cfg, pos, dL, n_par)
print("res", res, "off", off)
print("dev nz", [(i, float(g[i])) for i in np.nonzero(g)[0]])
print("ref nz", [(i, float(r[i])) for i in np.nonzero(r)[0]])
