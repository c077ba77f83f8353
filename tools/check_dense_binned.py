"""Table gradient with the multi-slice dense levels routed through the bins (NS_ENC_DENSE_BINNED=1, read once per process --
hence this script, run by tests/test_ngp_gpu.py in a subprocess): the packed sums of ns_ngp_encode_backward_fused_n equal the
owner-computes path's bit for bit, for ray-ordered samples and for samples clustered in two cells (slots overflow)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd")]
os.environ["NS_VARIANTS"] = os.environ["NS_ENC_DENSE_BINNED"] = "1"
from nerfslam._lib import check, lib, ptr, stream_ptr  # noqa: E402
from nerfslam.ngp import NgpConfig  # noqa: E402

dev = torch.device("cuda:0")
c = NgpConfig()
args = (c.n_levels, 2, c.log2_hashmap, c.base_res, C.c_float(c.per_level_scale))
off = (C.c_uint32 * 17)()
check(lib().ns_ngp_grid_layout(c.n_levels, 2, c.log2_hashmap, c.base_res, C.c_float(c.per_level_scale), None, None, off), "layout")
n_par = int(off[c.n_levels]) * 2
N, S = 1 << 17, 262144.0
for clustered in (False, True):
    rng = np.random.default_rng(5)
    if clustered:
        pos = (0.5 + rng.uniform(0, 1e-6, (N, 3))).astype(np.float32)
        pos[1::2, 0] += np.float32(1.3e-4)
    else:
        R = 1024
        o = rng.uniform(0.3, 0.7, (R, 1, 3))
        d = rng.standard_normal((R, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
        t = (0.02 + 0.0017 * np.arange(N // R))[None, :, None]
        pos = np.clip(o + t * d, 0.0, 1.0).reshape(N, 3).astype(np.float32)
    dLT = (rng.standard_normal((2 * c.n_levels, N)) * 1e-2).astype(np.float16)
    dLT[:, rng.uniform(size=N) < 0.2] = 0
    d_pos, d_dLT = torch.from_numpy(pos).to(dev), torch.from_numpy(dLT).to(dev)
    ref = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
    check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dLT), 1, ptr(ref), None, C.c_size_t(0), C.c_float(S), C.c_long(N),
                                       stream_ptr()), "owner-computes")
    wsb = int(lib().ns_ngp_encode_backward_fused_workspace_bytes(*args, C.c_long(N)))
    ws = torch.zeros(wsb // 8 + 1, dtype=torch.int64, device=dev)
    nul = C.c_void_p(0)
    for _ in range(2):
        gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
        check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(d_pos), ptr(d_dLT), ptr(gq), ptr(ws), C.c_size_t(wsb), C.c_float(S),
                                                   C.c_long(N), None, nul, nul, nul, nul, 1, C.c_float(0), C.c_float(0), C.c_float(0),
                                                   C.c_float(0), C.c_float(1), nul, 15, stream_ptr()), "fused")
        torch.cuda.synchronize()
        if not torch.equal(gq, ref):
            bad = (gq != ref).nonzero()
            raise SystemExit("binned dense levels: %d entries differ (first %d), clustered=%s" % (bad.shape[0], int(bad[0]), clustered))
    w32 = ws.view(torch.int32)
    assert int(w32[0]) == 0 and int(w32[1]) == 0, "overflow counter not cleared / error flag set"
print("dense levels through the bins: bit-identical")
