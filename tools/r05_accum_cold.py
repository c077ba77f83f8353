"""Is the table gradient's in-step time (87 + 105 us) against its stand-alone time (50 + 57 us) a matter of CACHE STATE?
The stand-alone figure comes from back-to-back launches on the same records: the optimiser state the accumulate pass touches
(f32 master + two moments + f16 copy: 176 MB for the default grid) and the records are still in the 256-MiB Infinity Cache from
the previous launch.  Inside the step they were last touched a whole step (~600 MB of other traffic) ago.  This tool times the
two passes (a) back to back, (b) each after a 1-GiB fill that evicts L2 and the Infinity Cache -- on a trained step's own
sample set (the bench's micro-bench), HIP events around every single launch.
usage: python tools/r05_accum_cold.py   -> one JSON line"""
import ctypes as C
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "nerf-slam_amd"), os.path.join(root, "tools")]
import torch

import bench
from nerfslam._lib import check, lib, ptr, stream_ptr

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net = bench._sphere_trainer(dev)
cf, S = net.cfg, net.cfg.max_samples
X = net.sets[1 - net.cur]
n_dev = C.c_void_p(X["counter"].data_ptr() + 8)
args = net._grid_args()
L = lib()
if len(sys.argv) > 1 and sys.argv[1] == "separate":      # three dense arrays (rounds 2-4) instead of the 32-byte records
    bw = {k: torch.zeros(net.n_grid, dtype=torch.float32, device=dev) for k in ("master", "m1", "m2")}
else:
    bw = dict(zip(("rec", "master", "m1", "m2"), type(net).new_grid_state(net.n_grid // 2, dev)))
bw["hp"] = torch.zeros_like(net.grid_half)
wsb = int(L.ns_ngp_encode_backward_fused_workspace_bytes(*args, C.c_long(S)))
bws = torch.zeros(wsb // 8 + 1, dtype=torch.int64, device=dev)
junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB


def run(parts):
    check(L.ns_ngp_encode_backward_fused_n(*args, ptr(X["s_pos"]), ptr(net.s_dfeat), None, ptr(bws), C.c_size_t(wsb),
                                           C.c_float(cf.grad_fixed_scale), C.c_long(S), n_dev, ptr(bw["master"]), ptr(bw["hp"]),
                                           ptr(bw["m1"]), ptr(bw["m2"]), 7, C.c_float(cf.lr), C.c_float(cf.beta1),
                                           C.c_float(cf.beta2), C.c_float(cf.eps), C.c_float(cf.loss_scale), None, parts, stream_ptr()),
          "ngp_encode_backward_fused")


def timed(parts, cold, n=12):
    ts = []
    for _ in range(n):
        if parts == 2:
            run(1)                       # the accumulate pass consumes what a scatter pass left
        if cold:
            junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(parts); e1.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1))
        if parts == 1:
            run(2)                       # drain the records (keeps the workspace counters consistent)
    ts.sort()
    return round(ts[len(ts) // 2], 1)


run(1); run(2); torch.cuda.synchronize()
out = {"samples": int(X["counter"][2].item()),
       "scatter_us": {"warm": timed(1, False), "after_1GiB_fill": timed(1, True)},
       "accumulate_us": {"warm": timed(2, False), "after_1GiB_fill": timed(2, True)},
       "note": "median of 12 single launches, HIP events; `after_1GiB_fill`: L2 and Infinity Cache evicted before the launch"}
print(json.dumps(out))
