"""bitwise comparison of ns_ngp_encode_forward_j_n between two builds of the library (tools/_bin/lib_prev.so vs the current one)"""
import ctypes as C, os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "nerf-slam_amd"))
from nerfslam.ngp import NgpConfig
dev = torch.device("cuda:0")
libs = {"new": C.CDLL(os.path.join(root, "nerf-slam_amd", "lib", "libnerfslam_hip.so")), "prev": C.CDLL(os.path.join(root, "tools", "_bin", "lib_prev.so"))}
c = NgpConfig()
L = c.n_levels
args = (L, 2, c.log2_hashmap, c.base_res, C.c_float(c.per_level_scale))
off = (C.c_uint32 * (L + 1))()
libs["new"].ns_ngp_grid_layout(L, 2, c.log2_hashmap, c.base_res, C.c_float(c.per_level_scale), None, None, off)
n_par = int(off[L]) * 2
g = torch.Generator().manual_seed(1)
N = 1 << 16
pos = torch.rand((N, 3), generator=g)
pos[:100] = torch.randn((100, 3), generator=g) * 3          # outside the unit cube (tail slots hold anything)
pos[100:110] = 1.0
pos[110:120] = 0.0
pos = pos.to(dev).contiguous()
par = (torch.rand(n_par, generator=g) - 0.5).half().to(dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = {}
for k, Lb in libs.items():
    feat = torch.zeros((2 * L, N), dtype=torch.float16, device=dev)
    jac = torch.zeros((6 * L, N), dtype=torch.float16, device=dev)
    rc = Lb.ns_ngp_encode_forward_j_n(*args, C.c_void_p(pos.data_ptr()), C.c_void_p(par.data_ptr()), C.c_void_p(feat.data_ptr()), 1,
                                      C.c_void_p(jac.data_ptr()), C.c_long(N), None, st)
    torch.cuda.synchronize()
    assert rc == 0, rc
    out[k] = (feat, jac)
for i, name in enumerate(("feat", "jac")):
    a, b = out["new"][i].view(torch.int16), out["prev"][i].view(torch.int16)
    d = (a != b)
    print(name, "mismatches", int(d.sum()), "of", d.numel())
    if d.any():
        idx = d.nonzero()[:10].tolist()
        for r, s in idx:
            print("  row", r, "sample", s, "new", float(out["new"][i][r, s]), "prev", float(out["prev"][i][r, s]), "pos", pos[s].tolist())
        rows = d.any(1).nonzero().flatten().tolist()
        print("  rows with mismatches:", rows[:40])
